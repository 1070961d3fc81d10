#!/bin/bash
O=gpurun_out/r3_run28; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o kg --output-format csv -- python3 $R/tools/keygen_time.py c5 > $R/$O/prof.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-200
rm -f $O/prof/*kernel_trace.csv
timeout 900 python3 -m pytest tests/test_gpu_cholesky_scale.py tests/test_gpu_psfp_parity.py tests/test_gpu_structured.py tests/test_gpu_boundary_completion.py tests/test_gpu_general_base.py -q -m gpu --durations=8 2>&1 | tail -14
