#!/bin/bash
O=gpurun_out/r3_run46; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests/test_gpu_psfp_parity.py tests/test_gpu_full_size.py -q -m gpu -x 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o t --output-format csv -- python3 $R/bench.py --config c3 --steps 5 --warmup 1 --no-cpu-baseline > $R/$O/rocprof.log 2>&1
f=$(find $R/$O/prof -name "*kernel_stats.csv" | head -1); grep -E "recombine_mfma_big" $f | cut -c1-40,150-300
rm -f $R/$O/prof/*kernel_trace.csv
