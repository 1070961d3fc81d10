#!/bin/bash
O=gpurun_out/r3_run40; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o t --output-format csv -- python3 $R/bench.py --config c3 --steps 5 --warmup 1 --no-cpu-baseline > $R/$O/rocprof.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); grep -E "recombine|zq_mfma|k_split|k_zq_combine|k_gadget|k_perturb|k_normals" $f | cut -c1-200
rm -f $O/prof/*kernel_trace.csv
