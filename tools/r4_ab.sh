#!/bin/bash
# headline A/B: the Z_q fold at batch 4096 (same box, alternating)
for r in 1 2; do for f in 0 1; do
  PSF_ZQ_FOLD128=$f timeout 300 python3 bench.py --no-cpu-baseline --no-latency --steps 5 --warmup 1 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fold128=$f', d['ms_per_step'], d['kernels_ms'])"
done; done
