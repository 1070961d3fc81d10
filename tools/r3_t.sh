#!/bin/bash
mkdir -p gpurun_out/fuzz
timeout 2500 python3 tools/fuzz_configs.py 20000 1500 2>&1 | grep -v amdgpu.ids | tee gpurun_out/fuzz/fuzz4.log | tail -12
