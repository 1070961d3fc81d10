#!/bin/bash
O=gpurun_out/r3_run31; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_cholesky_scale.py -q -m gpu --durations=5 2>&1 | tail -9
timeout 300 python3 tools/keygen_time.py c3 c5 2>&1 | tee $O/keygen.log
