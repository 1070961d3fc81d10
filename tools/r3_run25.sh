#!/bin/bash
O=gpurun_out/r3_run25; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_cholesky_scale.py tests/test_gpu_psfp_parity.py tests/test_gpu_structured.py tests/test_gpu_boundary_completion.py tests/test_gpu_general_base.py -q -m gpu -x 2>&1 | tail -25 > $O/tests.log
cat $O/tests.log
for v in stream gemm; do PSF_CHOL=$v python3 tools/keygen_time.py c3 bench64; done
