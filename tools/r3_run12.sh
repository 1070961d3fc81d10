#!/bin/bash
O=gpurun_out/r3_run12; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_cholesky_scale.py tests/test_gpu_gso.py tests/test_gpu_gpv_scale.py tests/test_gpu_boundary_completion.py -q -m gpu 2>&1 | tail -25 > $O/tests.log
cat $O/tests.log
python3 tools/keygen_time.py c3 c2 c4 > $O/keygen.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o kg --output-format csv -- python3 $R/tools/keygen_time.py c3 c2 > $R/$O/prof.log 2>&1
cd $R
cat $O/keygen.log; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-150
