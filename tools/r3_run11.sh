#!/bin/bash
O=gpurun_out/r3_run11; mkdir -p $O
R=$GRAFT_REPO_ROOT
python3 tools/keygen_time.py c3 c2 c4 > $O/keygen.log 2>&1
PSF_CHOL=right python3 tools/keygen_time.py c3 > $O/keygen_right.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o kg --output-format csv -- python3 $R/tools/keygen_time.py c3 > $R/$O/prof.log 2>&1
cd $R
cat $O/keygen.log $O/keygen_right.log; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-160
