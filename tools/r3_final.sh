#!/bin/bash
# Round-3 measurement pass (every step under `timeout`): bench lines with the CPU legs, rocprofv3 kernel stats, PMC traffic of the FP64 product and of the
# nearest plane (through the C++ harness), SQ counters of the sampler kernels, key-generation times.  Outputs under gpurun_out/r3_final/; what is judged is copied into profiles/.
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3_final; rm -rf $O; mkdir -p $O
for cfg in c3 c2 c4 c3prime c2s240; do
  timeout 600 python3 bench.py --config $cfg --steps 20 --warmup 2 > $O/bench_$cfg.log 2>&1; tail -1 $O/bench_$cfg.log > $O/bench_$cfg.json
done
timeout 300 python3 bench.py --config c3 --structured --steps 20 --warmup 2 > $O/bench_c3s.log 2>&1; tail -1 $O/bench_c3s.log > $O/bench_c3_structured.json
timeout 900 python3 bench.py --config c5 --steps 2 --warmup 1 > $O/bench_c5.log 2>&1; tail -1 $O/bench_c5.log > $O/bench_c5_one_gpu.json
python3 tools/keygen_time.py c3 c3prime c2 c4 c5 > $O/keygen.log 2>&1
cd /tmp
for cfg in c3 c2 c4; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$cfg -o t --output-format csv -- python3 $R/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline > $O/rocprof_$cfg.log 2>&1
  f=$(ls $O/prof_$cfg/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_$cfg.csv
  grep "^{\"metric\"" $O/rocprof_$cfg.log | tail -1 > $O/bench_under_rocprof_$cfg.json
  rm -rf $O/prof_$cfg
done
cd $R
timeout 600 tools/pmc_traffic.sh c3 k_trmm_f64 > $O/traffic_c3.json 2>$O/traffic_c3.err
timeout 600 tools/pmc_traffic.sh c3 k_trmm_f64 --structured > $O/traffic_c3_structured.json 2>>$O/traffic_c3.err
timeout 900 bash tools/pmc_np.sh c2 > $O/traffic_c2.json 2>$O/traffic_c2.err
timeout 900 bash tools/pmc_np.sh c4 > $O/traffic_c4.json 2>$O/traffic_c4.err
timeout 900 bash tools/pmc_sq.sh > $O/pmc_sq.log 2>&1; cp gpurun_out/pmc_sq/summary.txt $O/pmc_sq_summary.txt 2>/dev/null
ls -la $O | head -40; cat $O/traffic_c3.json | cut -c1-400; cat $O/keygen.log
