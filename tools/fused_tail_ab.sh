#!/bin/bash
# single call at C3: the separate rounding + syndrome kernels (PSF_FUSED_TAIL=0) against k_round_syndrome_small with one / two 16-row tiles per wave (experiments build)
export PSF_LIB="$(cd "$(dirname "$0")/.." && pwd)/tools_amd/lib/libpsf_mi355x_exp.so"
for rep in 1 2 3; do for k in off 1 2; do
  if [ $k = off ]; then export PSF_FUSED_TAIL=0; unset PSF_FUSED_RT; else export PSF_FUSED_TAIL=1 PSF_FUSED_RT=$k; fi
  echo -n "tiles per wave=$k: "
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['latency']; print(l['c3_b1_ms'], l['c3_b1_min_ms'], l.get('call_frac_b1'), l['kernels_ms_b1'])"
done; done
