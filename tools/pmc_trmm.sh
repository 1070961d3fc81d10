#!/bin/bash
# L2 hit rate and fetch traffic of k_trmm_f64 for several super-tile shapes (PSF_TRMM_GROUP = rows per group)
# the PSF_* switches below are alive in the experiments build only (make -C tools_amd/csrc exp); the release library reads none of them
export PSF_LIB="${PSF_LIB:-$(cd "$(dirname "$0")/.." && pwd)/tools_amd/lib/libpsf_mi355x_exp.so}"
export TMPDIR=/tmp; R=$PWD
for G in 8; do
  export PSF_TRMM_GROUP=$G
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/pmc_g$G -o t --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_g$G.log 2>&1
  python3 - <<PY
import csv, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open('gpurun_out/pmc_g$G/t_counter_collection.csv')):
    if 'trmm' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
h = sum(agg['TCC_HIT_sum'])/len(agg['TCC_HIT_sum']); m = sum(agg['TCC_MISS_sum'])/len(agg['TCC_MISS_sum'])
print("GROUP rows=$G cols=%d: hit-rate %.3f  miss*128B = %.1f GB" % (64//$G, h/(h+m), m*128/1e9))
PY
  grep -o '"k_trmm_f64": [0-9.]*' gpurun_out/pmc_g$G.log
done
