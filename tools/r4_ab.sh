#!/bin/bash
for cfg in c2 c4; do for r in 1 2; do
  timeout 300 python3 bench.py --config $cfg --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['valid'], d['kernels_ms'])"
done; done
