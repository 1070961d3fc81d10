#!/bin/bash
# Round-end measurement pass on the GPU box: bench lines (with the CPU legs) for the headline and the other single-GPU configurations, kernel-trace
# stats under rocprofv3, PMC traffic of the FP64 product, the probe harnesses.  Every step is bounded by `timeout` (the PMC passes of the nearest-plane
# configurations are left out: FETCH_SIZE collection segfaults at C2 and did not finish in 45 minutes at C4).  Outputs under gpurun_out/final/; copy what
# should be judged into profiles/.
# the PSF_* switches below are alive in the experiments build only (make -C tools_amd/csrc exp); the release library reads none of them
export PSF_LIB="${PSF_LIB:-$(cd "$(dirname "$0")/.." && pwd)/tools_amd/lib/libpsf_mi355x_exp.so}"
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
for cfg in c3 c2 c4; do
  timeout 600 python3 bench.py --config $cfg > $O/bench_$cfg.log 2>&1; tail -1 $O/bench_$cfg.log > $O/bench_$cfg.json
done
timeout 300 python3 bench.py --config c3 --structured > $O/bench_c3s.log 2>&1; tail -1 $O/bench_c3s.log > $O/bench_c3_structured.json
timeout 600 python3 bench.py --config c5 --steps 2 --warmup 1 > $O/bench_c5.log 2>&1; tail -1 $O/bench_c5.log > $O/bench_c5_one_gpu.json
PSF_TRMM_VARIANT=0 timeout 300 python3 bench.py --config c3 --no-cpu-baseline > $O/bench_c3_lds.log 2>&1; tail -1 $O/bench_c3_lds.log > $O/bench_c3_lds_kernel.json
PSF_TRMM_VARIANT=1 timeout 300 python3 bench.py --config c3 --no-cpu-baseline > $O/bench_c3_reg.log 2>&1; tail -1 $O/bench_c3_reg.log > $O/bench_c3_reg_kernel.json
cd /tmp
for cfg in c3 c2 c4; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$cfg -o t --output-format csv -- python3 $R/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline > $O/rocprof_$cfg.log 2>&1
  f=$(ls $O/prof_$cfg/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_$cfg.csv
  tail -1 $O/rocprof_$cfg.log > $O/bench_under_rocprof_$cfg.json
  rm -rf $O/prof_$cfg
done
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_c3s -o t --output-format csv -- python3 $R/bench.py --config c3 --structured --steps 5 --warmup 1 --no-cpu-baseline > $O/rocprof_c3s.log 2>&1
f=$(ls $O/prof_c3s/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_c3_structured.csv
rm -rf $O/prof_c3s
cd $R
timeout 600 tools/pmc_traffic.sh c3 k_trmm_f64 > $O/traffic_c3.json 2>$O/traffic_c3.err
timeout 600 tools/pmc_traffic.sh c3 k_trmm_f64 --structured > $O/traffic_c3_structured.json 2>>$O/traffic_c3.err
timeout 300 tools/bin/probe_trmm 240 32 3 > $O/probe_trmm.log 2>/dev/null
timeout 300 bash tools/pmc_probe_trmm.sh 0x10001000 > $O/probe_trmm_traffic.log 2>&1
timeout 100 tools/bin/probe_xcc > $O/probe_xcc.log 2>&1
timeout 300 bash tools/pmc_l2share.sh > $O/probe_l2share.log 2>&1
PSF_LIB=$R/tools_amd/lib/libpsf_clock_probe.so timeout 300 python3 tools/trmm_clock_probe.py > $O/trmm_clock.log 2>&1
timeout 300 tools/trace_timeline.sh c2 final > /dev/null 2>&1; cp gpurun_out/final_trace_c2.csv $O/trace_c2.csv 2>/dev/null
timeout 600 python3 tools/cpu_faithful.py > $O/cpu_faithful.json 2>/dev/null
ls -la $O; cat $O/traffic_*.json; tail -3 $O/trmm_clock.log
