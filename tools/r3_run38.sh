#!/bin/bash
O=gpurun_out/r3_run38; mkdir -p $O
for c in c4 c2; do timeout 300 python3 bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'][:20], d['ms_per_step'], d['kernels_ms'])" | tee -a $O/log.txt; done
timeout 900 python3 -m pytest tests/test_gpu_gpv_parity.py tests/test_gpu_gpv_scale.py tests/test_gpu_ring_parity.py -q -m gpu -x 2>&1 | tail -4
