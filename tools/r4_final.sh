#!/bin/bash
# Round-4 measurement pass on the GPU box (every step bounded by `timeout`): bench lines with the CPU legs and the new `latency` object, kernel-trace
# stats under rocprofv3 (headline + one-call regime), PMC traffic (FP64 product at batch 4096, the streaming product of a 16-preimage call, the
# nearest plane through np_harness), the single-call record with the reference's three bench sets, the host-pointer path, the probes.
# Outputs under gpurun_out/r4_final/; tools/r4_collect.py copies what should be judged into profiles/ and refreshes the hash-tied traffic files.
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r4_final; rm -rf $O; mkdir -p $O
for cfg in c3 c3prime c2 c2s240 c4; do
  timeout 600 python3 bench.py --config $cfg > $O/bench_$cfg.log 2>&1; tail -1 $O/bench_$cfg.log > $O/bench_$cfg.json
done
timeout 300 python3 bench.py --config c3 --structured > $O/bench_c3s.log 2>&1; tail -1 $O/bench_c3s.log > $O/bench_c3_structured.json
timeout 900 python3 bench.py --config c5 --steps 2 --warmup 1 > $O/bench_c5.log 2>&1; tail -1 $O/bench_c5.log > $O/bench_c5_one_gpu.json
cd /tmp
for cfg in c3 c2 c4; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$cfg -o t --output-format csv -- python3 $R/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline > $O/rocprof_$cfg.log 2>&1
  f=$(ls $O/prof_$cfg/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_$cfg.csv
  grep '^{"metric' $O/rocprof_$cfg.log | tail -1 > $O/bench_under_rocprof_$cfg.json
  rm -rf $O/prof_$cfg
done
cd $R
timeout 900 python3 tools/single_call.py --out $O/single_call.json > $O/single_call.log 2>&1
bash tools/prof_single_call.sh r4final 1,16,64 > $O/prof_single.log 2>&1
cp gpurun_out/r4final_kernel_stats_single.csv $O/kernel_stats_single_call.csv 2>/dev/null; cp gpurun_out/r4final_trace_single.txt $O/trace_single_call.txt 2>/dev/null
timeout 600 tools/pmc_traffic.sh c3 k_trmm_f64 > $O/traffic_c3.json 2>$O/traffic_c3.err
timeout 600 tools/pmc_traffic.sh c3 k_trmm_f64 --structured > $O/traffic_c3_structured.json 2>>$O/traffic_c3.err
timeout 600 bash tools/pmc_single_call.sh 16 > $O/traffic_single_b16.json 2>&1
timeout 600 bash tools/pmc_single_call.sh 1 > $O/traffic_single_b1.json 2>&1
for cfg in c2 c4; do timeout 1200 bash tools/pmc_np.sh $cfg > $O/traffic_$cfg.json 2>$O/traffic_$cfg.err; done
timeout 300 tools/bin/probe_stream 30801 5 > $O/probe_stream.log 2>&1
timeout 300 python3 tools/host_path_timing.py 32 > $O/host_path.log 2>&1
timeout 300 bash tools/prof_host_async.sh > $O/host_async_timeline.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_poly -o t --output-format csv -- python3 $R/tools/time_polymul.py > $O/polymul.log 2>&1
f=$(ls $O/prof_poly/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_polymul.csv
rm -rf $O/prof_poly
cd $R
timeout 600 python3 tools/keygen_time.py c3 c2 c4 > $O/keygen.log 2>&1
ls -la $O; tail -1 $O/bench_c3.json | cut -c1-600
