#!/bin/bash
O=gpurun_out/r3_run34; mkdir -p $O
timeout 300 tools/bin/probe_coissue 60 32 i8 2>&1 | tee $O/probe_i8.log
timeout 100 python3 tools/keygen_time.py c3 c2 2>&1 | grep rep | tee $O/keygen.log
timeout 600 python3 -m pytest tests/test_gpu_cholesky_scale.py tests/test_gpu_gso.py tests/test_gpu_gpv_scale.py -q -m gpu 2>&1 | tail -4
