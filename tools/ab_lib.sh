#!/bin/bash
# same-box A/B of two builds of the library (PSF_LIB) on one bench configuration, alternating: tools/ab_lib.sh <config> <libA> <libB> [rounds]
cfg=$1; A=$2; B=$3; n=${4:-3}
for r in $(seq $n); do
  for L in $A $B; do
    PSF_LIB=$PWD/$L timeout 300 python3 bench.py --config $cfg --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', '$L', d['ms_per_step'], d['kernels_ms'])"
  done
done
