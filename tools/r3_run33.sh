#!/bin/bash
O=gpurun_out/r3_run33; mkdir -p $O
for g in "8 4" "4 8" "2 16" "16 2"; do
  set -- $g
  export PSF_TRMM_GR=$1 PSF_TRMM_GC=$2
  echo "== GR=$1 GC=$2" | tee -a $O/log.txt
  timeout 300 python3 bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernels_ms'])" | tee -a $O/log.txt
  timeout 600 tools/pmc_traffic.sh c3 k_trmm_f64_big | tee -a $O/log.txt
done
