#!/bin/bash
# Round-end measurement pass on the GPU box: bench lines (with the CPU legs) for the headline and the other single-GPU configurations, kernel-trace
# stats under rocprofv3, PMC traffic of the dominant kernels.  Outputs under gpurun_out/final/; copy what should be judged into profiles/.
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/final; mkdir -p $O
for cfg in c3 c2 c4; do
  python3 bench.py --config $cfg > $O/bench_$cfg.log 2>&1; tail -1 $O/bench_$cfg.log > $O/bench_$cfg.json
done
python3 bench.py --config c3 --structured > $O/bench_c3s.log 2>&1; tail -1 $O/bench_c3s.log > $O/bench_c3_structured.json
python3 bench.py --config c5 --steps 2 --warmup 1 > $O/bench_c5.log 2>&1; tail -1 $O/bench_c5.log > $O/bench_c5_one_gpu.json
cd /tmp
for cfg in c3 c2 c4; do
  rocprofv3 --kernel-trace --stats -d $O/prof_$cfg -o t --output-format csv -- python3 $R/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline > $O/rocprof_$cfg.log 2>&1
  f=$(ls $O/prof_$cfg/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_$cfg.csv
  rm -rf $O/prof_$cfg
done
rocprofv3 --kernel-trace --stats -d $O/prof_c3s -o t --output-format csv -- python3 $R/bench.py --config c3 --structured --steps 5 --warmup 1 --no-cpu-baseline > $O/rocprof_c3s.log 2>&1
f=$(ls $O/prof_c3s/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_c3_structured.csv
rm -rf $O/prof_c3s
cd $R
tools/pmc_traffic.sh c3 k_trmm_f64 > $O/traffic_c3.json 2>$O/traffic_c3.err
tools/pmc_traffic.sh c3 k_trmm_f64 --structured > $O/traffic_c3_structured.json 2>>$O/traffic_c3.err
tools/pmc_traffic.sh c2 k_np_step > $O/traffic_c2.json 2>>$O/traffic_c3.err
tools/pmc_traffic.sh c4 k_np_step > $O/traffic_c4.json 2>>$O/traffic_c3.err
python3 tools/cpu_faithful.py > $O/cpu_faithful.json 2>/dev/null
ls -la $O; cat $O/traffic_*.json
