#!/bin/bash
# Round-5 measurement pass on the GPU box (every step bounded by `timeout`): bench lines with the CPU legs and the `latency` objects (now also for the nearest-plane
# types), kernel-trace stats under rocprofv3 (headline, C2, C4, one-call regime, R_q products), PMC traffic (FP64 product, nearest plane through np_harness), the
# single-call record, the host-pointer paths of all three types, the probes, key generation.  Outputs under gpurun_out/r5_final/; tools/r5_collect.py copies what
# should be judged into profiles/ and refreshes the hash-tied traffic files.     usage: tools/r5_final.sh [part ...]   (parts: bench prof pmc single host misc; default all)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r5_final; mkdir -p $O
parts=${@:-bench prof pmc single host misc}
has() { [[ " $parts " == *" $1 "* ]]; }
if has bench; then
  for cfg in c3 c3prime c2 c2s240 c4; do
    timeout 600 python3 bench.py --config $cfg > $O/bench_$cfg.log 2>&1; tail -1 $O/bench_$cfg.log > $O/bench_$cfg.json
  done
  PSF_NP_WALK=0 timeout 600 python3 bench.py --config c2 --no-cpu-baseline --no-latency > $O/bench_c2_launch_per_block.log 2>&1; tail -1 $O/bench_c2_launch_per_block.log > $O/bench_c2_launch_per_block.json
  PSF_NP_WALK=3 timeout 600 python3 bench.py --config c4 --no-cpu-baseline --no-latency > $O/bench_c4_walk2.log 2>&1; tail -1 $O/bench_c4_walk2.log > $O/bench_c4_walk2.json
  # comparison arms in the same box: the samplers / syndrome product of round 4 (fp32 screen, every digit pair), one recombination launch per digit pair
  PSF_ROUND=lean PSF_ZQ_POW2=0 timeout 600 python3 bench.py --config c3 --no-cpu-baseline --no-latency > $O/bench_c3_round4_stages.log 2>&1; tail -1 $O/bench_c3_round4_stages.log > $O/bench_c3_round4_stages.json
  PSF_NP_COMBINE=0 timeout 600 python3 bench.py --config c2 --no-cpu-baseline --no-latency > $O/bench_c2_combine_per_pair.log 2>&1; tail -1 $O/bench_c2_combine_per_pair.log > $O/bench_c2_combine_per_pair.json
  PSF_NP_COMBINE=0 timeout 600 python3 bench.py --config c4 --no-cpu-baseline --no-latency > $O/bench_c4_combine_per_pair.log 2>&1; tail -1 $O/bench_c4_combine_per_pair.log > $O/bench_c4_combine_per_pair.json
  timeout 300 python3 bench.py --config c3 --structured > $O/bench_c3s.log 2>&1; tail -1 $O/bench_c3s.log > $O/bench_c3_structured.json
  timeout 900 python3 bench.py --config c5 --steps 2 --warmup 1 > $O/bench_c5.log 2>&1; tail -1 $O/bench_c5.log > $O/bench_c5_one_gpu.json
fi
if has prof; then
  cd /tmp
  for cfg in c3 c2 c4; do
    timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$cfg -o t --output-format csv -- python3 $R/bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --no-latency > $O/rocprof_$cfg.log 2>&1
    f=$(ls $O/prof_$cfg/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_$cfg.csv
    rm -rf $O/prof_$cfg
  done
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_poly -o t --output-format csv -- python3 $R/tools/time_polymul.py > $O/polymul.log 2>&1
  f=$(ls $O/prof_poly/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_polymul.csv
  rm -rf $O/prof_poly
  cd $R
  bash tools/prof_single_call.sh r5final 1,16,64 > $O/prof_single.log 2>&1
  cp gpurun_out/r5final_kernel_stats_single.csv $O/kernel_stats_single_call.csv 2>/dev/null; cp gpurun_out/r5final_trace_single.txt $O/trace_single_call.txt 2>/dev/null
fi
if has pmc; then
  timeout 600 tools/pmc_traffic.sh c3 k_trmm_f64 > $O/traffic_c3.json 2>$O/traffic_c3.err
  for cfg in c2 c4; do timeout 1200 bash tools/pmc_np.sh $cfg > $O/traffic_$cfg.json 2>$O/traffic_$cfg.err; done
  timeout 600 bash tools/pmc_single_call.sh 16 > $O/traffic_single_b16.json 2>&1
fi
if has single; then
  timeout 900 python3 tools/single_call.py --out $O/single_call.json > $O/single_call.log 2>&1
  timeout 600 python3 tools/time_midsize.py > $O/midsize.log 2>&1
fi
if has host; then
  timeout 300 python3 tools/host_path_timing.py 32 > $O/host_path.log 2>&1
  timeout 300 python3 tools/host_path_timing_gpv.py > $O/host_path_gpv.log 2>&1
  timeout 600 python3 tools/host_async_stress_gpv.py 300 > $O/host_async_stress_gpv.log 2>&1
fi
if has misc; then
  timeout 300 python3 tools/time_ring_fa.py > $O/ring_fa.log 2>&1; PSF_RING_FA=matmul timeout 300 python3 tools/time_ring_fa.py >> $O/ring_fa.log 2>&1
  timeout 300 tools/bin/probe_ldsdma_l2 > $O/probe_ldsdma_l2.log 2>&1
  timeout 600 python3 tools/keygen_time.py c3 c2 c4 > $O/keygen.log 2>&1
  PSF_KEYGEN_TIMING=1 timeout 600 python3 tools/keygen_time.py c3 c2 c4 > $O/keygen_phases.log 2>&1
  bash tools/keygen_timeline.sh c3 r5final > /dev/null 2>&1; cp gpurun_out/r5final_keygen_timeline_c3.txt $O/keygen_timeline_c3.txt 2>/dev/null
  bash tools/keygen_timeline.sh c2 r5final > /dev/null 2>&1; cp gpurun_out/r5final_keygen_timeline_c2.txt $O/keygen_timeline_c2.txt 2>/dev/null
fi
ls -la $O | tail -40; tail -1 $O/bench_c3.json 2>/dev/null | cut -c1-600
