#!/bin/bash
# same-box A/B of two builds of the library (PSF_LIB selects the one to load): tools/bin/libpsf_base.so (built from HEAD) against the working tree's
cd "$(dirname "$0")/.."
for r in 1 2 3; do for lib in tools/bin/libpsf_base.so tools_amd/lib/libpsf_mi355x.so; do for cfg in "$@"; do
  PSF_LIB=$PWD/$lib timeout 300 python3 bench.py --config $cfg --no-cpu-baseline --no-latency --steps 10 --warmup 2 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', '$cfg', d['ms_per_step'], d['valid'], d['kernels_ms'])"
done; done; done
