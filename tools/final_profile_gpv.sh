#!/bin/bash
# the nearest-plane part of tools/final_profile.sh alone (bench lines with CPU legs, rocprofv3 kernel stats, timeline of one C2 call)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/final; mkdir -p $O
for cfg in c2 c4; do
  timeout 600 python3 bench.py --config $cfg > $O/bench_$cfg.log 2>&1; tail -1 $O/bench_$cfg.log > $O/bench_$cfg.json
done
cd /tmp
for cfg in c2 c4; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$cfg -o t --output-format csv -- python3 $R/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline > $O/rocprof_$cfg.log 2>&1
  f=$(ls $O/prof_$cfg/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_$cfg.csv
  rm -rf $O/prof_$cfg
done
cd $R
timeout 300 tools/trace_timeline.sh c2 final > /dev/null 2>&1; cp gpurun_out/final_trace_c2.csv $O/trace_c2.csv 2>/dev/null
tail -c 600 $O/bench_c2.json; echo; tail -c 600 $O/bench_c4.json; echo; grep np_step $O/kernel_stats_c2.csv $O/kernel_stats_c4.csv | cut -c1-220
