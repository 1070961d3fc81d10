#!/bin/bash
O=gpurun_out/r3_run32; mkdir -p $O
R=$GRAFT_REPO_ROOT
for i in 1 2; do timeout 100 python3 tools/keygen_time.py c3 2>&1 | grep rep | tee -a $O/keygen.log; done
PSF_CHOL=stream timeout 100 python3 tools/keygen_time.py c3 2>&1 | grep rep | sed 's/^/stream /' | tee -a $O/keygen.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o kg --output-format csv -- python3 $R/tools/keygen_time.py c5 > $R/$O/prof.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-200 | tee $O/c5_kernel_stats_head.csv
rm -f $O/prof/*kernel_trace.csv
