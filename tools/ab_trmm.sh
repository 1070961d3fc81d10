#!/bin/bash
# library-level A/B of the FP64 product variants (PSF_TRMM_VARIANT), alternating to see the run-to-run spread
# the PSF_* switches below are alive in the experiments build only (make -C tools_amd/csrc exp); the release library reads none of them
export PSF_LIB="${PSF_LIB:-$(cd "$(dirname "$0")/.." && pwd)/tools_amd/lib/libpsf_mi355x_exp.so}"
O=gpurun_out/ab; mkdir -p $O
line() { tail -1 "$1" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'], d.get('kernels_ms')['k_trmm_f64'], d['roofline']['frac'])"; }
PSF_TRMM_VARIANT=1 timeout 300 python3 -m pytest tests/test_gpu_psfp_parity.py tests/test_gpu_structured.py -q -m gpu -x 2>&1 | tail -2
for r in 1 2; do
for v in 0 1; do
  PSF_TRMM_VARIANT=$v timeout 300 python3 bench.py --config c3 --no-cpu-baseline --steps 5 --warmup 1 > $O/c3_v$v.log 2>&1; line $O/c3_v$v.log c3_v$v
done; done
PSF_TRMM_VARIANT=0 timeout 300 python3 bench.py --config c3 --structured --no-cpu-baseline --steps 5 --warmup 1 > $O/c3s_v0.log 2>&1; line $O/c3s_v0.log c3s_v0
PSF_TRMM_VARIANT=1 timeout 300 python3 bench.py --config c3 --structured --no-cpu-baseline --steps 5 --warmup 1 > $O/c3s_v1.log 2>&1; line $O/c3s_v1.log c3s_v1
