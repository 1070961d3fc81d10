#!/bin/bash
# A/B pass for experimental switches (environment variables read by the library): parity tests under the switch, then the bench line.
# usage: tools/ab_variants.sh  -> gpurun_out/ab/*.json
# the PSF_* switches below are alive in the experiments build only (make -C tools_amd/csrc exp); the release library reads none of them
export PSF_LIB="${PSF_LIB:-$(cd "$(dirname "$0")/.." && pwd)/tools_amd/lib/libpsf_mi355x_exp.so}"
O=gpurun_out/ab; mkdir -p $O
line() { tail -1 "$1" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['ms_per_step'], d.get('kernels_ms'), d.get('roofline'))"; }
# nearest plane: parity first
timeout 300 python3 -m pytest tests/test_gpu_gpv_parity.py tests/test_gpu_ring_parity.py -q -m gpu -x 2>&1 | tail -2
PSF_NP_IMMEDIATE=1 timeout 300 python3 -m pytest tests/test_gpu_gpv_parity.py tests/test_gpu_ring_parity.py -q -m gpu -x 2>&1 | tail -2
for cfg in c2 c4; do
  timeout 300 python3 bench.py --config $cfg --no-cpu-baseline --steps 10 --warmup 2 > $O/${cfg}_base.log 2>&1; line $O/${cfg}_base.log ${cfg}_base
  PSF_NP_IMMEDIATE=1 timeout 300 python3 bench.py --config $cfg --no-cpu-baseline --steps 10 --warmup 2 > $O/${cfg}_imm.log 2>&1; line $O/${cfg}_imm.log ${cfg}_imm
done
# FP64 product
PSF_TRMM_VARIANT=1 timeout 300 python3 -m pytest tests/test_gpu_psfp_parity.py tests/test_gpu_structured.py -q -m gpu -x 2>&1 | tail -2
timeout 300 python3 bench.py --config c3 --no-cpu-baseline --steps 5 --warmup 1 > $O/c3_v0.log 2>&1; line $O/c3_v0.log c3_v0
PSF_TRMM_VARIANT=1 timeout 300 python3 bench.py --config c3 --no-cpu-baseline --steps 5 --warmup 1 > $O/c3_v1.log 2>&1; line $O/c3_v1.log c3_v1
