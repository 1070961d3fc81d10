#!/bin/bash
O=gpurun_out/r3_run45; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in SCHED; do
export PSF_LIB=$R/tools_amd/lib/libpsf_$v.so
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$v -o t --output-format csv -- python3 $R/bench.py --config c3 --steps 5 --warmup 1 --no-cpu-baseline > $R/$O/rocprof_$v.log 2>&1
f=$(find $R/$O/prof_$v -name "*kernel_stats.csv" | head -1); echo "== $v"; grep -E "recombine_mfma_big" $f | cut -c1-40,150-300
rm -f $R/$O/prof_$v/*kernel_trace.csv
done
