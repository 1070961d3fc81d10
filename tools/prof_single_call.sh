#!/bin/bash
# rocprofv3 kernel trace of C3 single calls: tools/prof_single_call.sh <tag> <batches>   -> gpurun_out/<tag>_kernel_stats_single.csv + the per-dispatch trace of the last call
export TMPDIR=/tmp
tag=$1; batches=${2:-1}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_single -o t --output-format csv -- python3 $R/tools/single_call.py --batches $batches --reps 3 --skip-sets --out $O/${tag}_single_under_rocprof.json > $O/${tag}_single_under_rocprof.log 2>&1
f=$(ls $O/prof_${tag}_single/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" $O/${tag}_kernel_stats_single.csv
t=$(ls $O/prof_${tag}_single/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$t" ] && python3 - "$t" > $O/${tag}_trace_single.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last samp_p call: from the last k_normals_wave on
idx = max(i for i, r in enumerate(rows) if "k_normals_wave" in r["Kernel_Name"])
# walk back to the stream variant's call (second to last k_normals belongs to the batch-kernel arm): print the last two calls
starts = [i for i, r in enumerate(rows) if "k_normals_wave" in r["Kernel_Name"]]
for s in starts[-8:-7] + starts[-1:]:
    t0 = int(rows[s]["Start_Timestamp"])
    prev_end = t0
    print("call starting at dispatch", s)
    for r in rows[s:s + 12]:
        if "k_normals_wave" in r["Kernel_Name"] and r is not rows[s]:
            break
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"  {r['Kernel_Name'][:70]:70s} start +{(a - t0) / 1e3:9.1f} us  dur {(b - a) / 1e3:8.1f} us  gap {(a - prev_end) / 1e3:6.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))}")
        prev_end = b
PY
rm -rf $O/prof_${tag}_single
cd $R
column -s, -t < $O/${tag}_kernel_stats_single.csv | cut -c1-150 | head -20
cat $O/${tag}_trace_single.txt
