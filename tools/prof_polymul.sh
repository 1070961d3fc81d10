#!/bin/bash
# rocprofv3 over tools/time_polymul.py: kernel-trace stats (-> gpurun_out/<tag>_kernel_stats_polymul.csv) and one SQ counter pass for the NTT kernels.
# usage: tools/prof_polymul.sh <tag>
export TMPDIR=/tmp
tag=${1:-r05}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof_pm -o t --output-format csv -- python3 $R/tools/time_polymul.py > $O/${tag}_polymul_under_rocprof.log 2>&1
f=$(ls $O/prof_pm/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" $O/${tag}_kernel_stats_polymul.csv
rm -rf $O/prof_pm
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES \
  -d $O/pmc_pm -o t --output-format csv -- python3 $R/tools/time_polymul.py 53248 3 > $O/pmc_pm.log 2>&1
cd $R
python3 - "$tag" <<'PY' > $O/${tag}_pmc_polymul.txt
import csv, collections, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_pm/*counter_collection.csv') + glob.glob('gpurun_out/pmc_pm/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('psf::', '')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg):
    if 'ntt' not in k: continue
    c = {n: sum(v) / len(v) for n, v in agg[k].items()}
    print(f"== {k}  (average per launch over {len(next(iter(agg[k].values())))} launches)")
    for n in sorted(c): print(f"   {n:24s} {c[n]:.4g}")
    wc = c.get('SQ_WAVE_CYCLES')
    if wc:
        print("   -> of the wave cycles: parked %.1f %%, issue-stalled %.1f %%, issuing %.1f %%" % (100 * c.get('SQ_WAIT_ANY', 0) / wc, 100 * c.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc))
    if c.get('SQ_WAVES'): print("   -> VALU instructions per wave: %.0f" % (c.get('SQ_INSTS_VALU', 0) / c['SQ_WAVES']))
PY
rm -rf $O/pmc_pm
column -s, -t < $O/${tag}_kernel_stats_polymul.csv | cut -c1-200 | head -12
cat $O/${tag}_pmc_polymul.txt
tail -1 $O/${tag}_polymul_under_rocprof.log | cut -c1-600
