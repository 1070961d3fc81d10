#!/bin/bash
O=gpurun_out/r3_run36; mkdir -p $O
for i in 1 2; do timeout 100 python3 tools/keygen_time.py c3 c2 c4 2>&1 | grep rep | tee -a $O/keygen.log; done
PSF_CHOL=stream timeout 100 python3 tools/keygen_time.py c3 2>&1 | grep rep | sed 's/^/stream /' | tee -a $O/keygen.log
timeout 600 python3 -m pytest tests/test_gpu_cholesky_scale.py tests/test_gpu_gso.py -q -m gpu 2>&1 | tail -3
