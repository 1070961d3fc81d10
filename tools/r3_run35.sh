#!/bin/bash
O=gpurun_out/r3_run35; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in c3 c2 c4; do
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$c -o kg --output-format csv -- python3 $R/tools/keygen_time.py $c > $R/$O/prof_$c.log 2>&1
f=$(find $R/$O/prof_$c -name "*kernel_stats.csv" | head -1); echo "== $c"; head -12 $f | cut -c1-150
rm -f $R/$O/prof_$c/*kernel_trace.csv
done
