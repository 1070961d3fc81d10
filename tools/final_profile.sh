#!/bin/bash
# Round-end measurement pass on the GPU box: kernel-trace stats for C3, C2 and C4 (rocprofv3), then the plain bench lines with the CPU baseline.
# Outputs under gpurun_out/final/; copy what should be judged into profiles/.
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/final; mkdir -p $O
cd /tmp
for cfg in c3 c2 c4; do
  rocprofv3 --kernel-trace --stats -d $O/prof_$cfg -o t --output-format csv -- python3 $R/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof_$cfg.log 2>&1
  f=$(ls $O/prof_$cfg/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" $O/kernel_stats_$cfg.csv
  tail -1 $O/bench_under_rocprof_$cfg.log > $O/bench_under_rocprof_$cfg.json
done
cd $R
for cfg in c3 c2 c4; do
  python3 bench.py --config $cfg > $O/bench_$cfg.log 2>&1
  tail -1 $O/bench_$cfg.log > $O/bench_$cfg.json
done
rm -rf $O/prof_c3 $O/prof_c2 $O/prof_c4
ls -la $O
