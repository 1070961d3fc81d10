#!/bin/bash
O=gpurun_out/r3_run10; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_cholesky_scale.py tests/test_gpu_psfp_parity.py tests/test_gpu_structured.py tests/test_gpu_boundary_completion.py -q -m gpu 2>&1 | tail -25 > $O/tests.log
cat $O/tests.log
for v in left right; do PSF_CHOL=$v timeout 600 python3 bench.py --config c3 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'trap_gen_s', d['trap_gen_s'], 'valid', d['valid'], d['ms_per_step'])"; done
