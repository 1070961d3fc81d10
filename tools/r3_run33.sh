#!/bin/bash
O=gpurun_out/r3_run33; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_psfp_parity.py tests/test_gpu_pipeline_mode.py tests/test_gpu_structured.py tests/test_gpu_general_base.py tests/test_gpu_boundary_functions.py -q -m gpu 2>&1 | tail -5
PSF_HALVES=1 timeout 900 python3 -m pytest tests/test_gpu_psfp_parity.py tests/test_gpu_pipeline_mode.py tests/test_gpu_structured.py tests/test_gpu_distribution.py -q -m gpu 2>&1 | tail -5
for r in 1 2 3; do for v in 0 1; do PSF_HALVES=$v timeout 300 python3 bench.py --config c3 --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('halves=$v', d['ms_per_step'], d['valid'], d['kernels_ms'])"; done; done
