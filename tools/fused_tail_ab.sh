#!/bin/bash
# single call at C3 with and without the fused tail of k_trmm_stream_fused (experiments build for the switch)
export PSF_LIB="$(cd "$(dirname "$0")/.." && pwd)/tools_amd/lib/libpsf_mi355x_exp.so"
for rep in 1 2; do for f in 0 1; do
  echo -n "PSF_FUSED_TAIL=$f: "
  PSF_FUSED_TAIL=$f python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['latency']; print(l['c3_b1_ms'], l['c3_b1_min_ms'], l.get('call_frac_b1'), l['kernels_ms_b1'])"
done; done
