#!/bin/bash
mkdir -p gpurun_out/fuzz gpurun_out/r3_full
timeout 2400 python3 -m pytest tests -q -m gpu > gpurun_out/r3_full/tests.log 2>&1; tail -4 gpurun_out/r3_full/tests.log
timeout 3000 python3 tools/fuzz_configs.py 1300 1000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/fuzz/fuzz2.log | tail -20
