#!/bin/bash
# headline check: bench c3 twice + parity of the samplers
timeout 600 python3 -m pytest tests/test_gpu_psfp_parity.py tests/test_gpu_single_call.py tests/test_gpu_distribution.py -x -q -m gpu 2>&1 | tail -2
for r in 1 2 3; do
  timeout 300 python3 bench.py --no-cpu-baseline --no-latency --steps 5 --warmup 1 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['valid'], d['kernels_ms'])"
done
