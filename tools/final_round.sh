#!/bin/bash
# Round-end measurement pass on the GPU box with the RELEASE library: the default bench line (headline + other_configs + latency + CPU legs), the c2 / c4 lines,
# rocprofv3 kernel-trace stats of the same commands, the PMC traffic of the dominant kernel (FETCH_SIZE / WRITE_SIZE in separate passes, no trace domains) and of
# one nearest-plane call through tools/bin/np_harness.  Every step bounded by `timeout`.  usage: tools/final_round.sh <tag>; outputs under gpurun_out/<tag>_final/
unset PSF_LIB
export TMPDIR=/tmp
tag=${1:-r06}
R=$PWD; O=$R/gpurun_out/${tag}_final; rm -rf $O; mkdir -p $O
timeout 600 python3 bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_c3.json
for cfg in c2 c4 c3prime; do
  timeout 600 python3 bench.py --config $cfg > $O/bench_$cfg.log 2>&1; tail -1 $O/bench_$cfg.log > $O/bench_$cfg.json
done
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o t --output-format csv -- python3 $R/bench.py > $O/rocprof_c3.log 2>&1
f=$(ls $O/prof_c3/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_c3.csv
tail -1 $O/rocprof_c3.log > $O/bench_under_rocprof_c3.json; rm -rf $O/prof_c3
for cfg in c2 c4; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$cfg -o t --output-format csv -- python3 $R/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline > $O/rocprof_$cfg.log 2>&1
  f=$(ls $O/prof_$cfg/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_$cfg.csv
  tail -1 $O/rocprof_$cfg.log > $O/bench_under_rocprof_$cfg.json; rm -rf $O/prof_$cfg
done
cd $R
timeout 600 tools/pmc_traffic.sh c3 k_trmm_f64_big --no-other-configs --no-latency > $O/traffic_c3.json 2>$O/traffic_c3.err
timeout 900 tools/pmc_np.sh c2 > $O/traffic_c2.json 2>$O/traffic_c2.err
timeout 900 tools/pmc_np.sh c4 > $O/traffic_c4.json 2>$O/traffic_c4.err
timeout 300 python3 tools/keygen_time.py c3 c2 c4 > $O/keygen.log 2>&1
ls -la $O; head -c 600 $O/traffic_c3.json; echo; head -c 300 $O/traffic_c2.json; echo; head -c 300 $O/traffic_c4.json; echo; cat $O/keygen.log
