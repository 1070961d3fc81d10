#!/bin/bash
O=gpurun_out/r3_run29; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_cholesky_scale.py tests/test_gpu_psfp_parity.py tests/test_gpu_structured.py tests/test_gpu_boundary_completion.py -q -m gpu --durations=5 2>&1 | tail -12
timeout 300 python3 tools/keygen_time.py c3 c5 2>&1 | tee $O/keygen.log
PSF_CHOL=gemm timeout 100 python3 tools/keygen_time.py c3 2>&1 | tee -a $O/keygen.log
