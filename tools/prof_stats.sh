#!/bin/bash
# rocprofv3 kernel-trace stats of one bench configuration: tools/prof_stats.sh <config> <tag> [bench args...]
# writes gpurun_out/<tag>_kernel_stats_<config>.csv and the bench line next to it
export TMPDIR=/tmp
cfg=$1; tag=$2; shift 2
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_$cfg -o t --output-format csv -- python3 $R/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline "$@" > $O/${tag}_bench_under_rocprof_$cfg.log 2>&1
f=$(ls $O/prof_${tag}_$cfg/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" $O/${tag}_kernel_stats_$cfg.csv
rm -rf $O/prof_${tag}_$cfg
cd $R
column -s, -t < $O/${tag}_kernel_stats_$cfg.csv | cut -c1-160 | head -25
tail -1 $O/${tag}_bench_under_rocprof_$cfg.log | cut -c1-400
