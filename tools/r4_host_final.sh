#!/bin/bash
# Host-pointer path, final records of the round: transport parity test, timing, full-size row check, kernel + copy timeline, and the three bench lines whose
# `roofline.traffic` needs the traffic records written by the pass before.   bash tools/r4_host_final.sh  (writes gpurun_out/r4_host/)
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r4_host; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pipeline_mode.py tests/test_gpu_boundary_completion.py tests/test_cpp_mirror.py -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 300 python3 tools/host_path_timing.py 8 > $O/host_path.log 2>&1; tail -4 $O/host_path.log
PSF_HOST_COPY=runtime timeout 300 python3 tools/host_path_timing.py 8 > $O/host_path_runtime_copies.log 2>&1; tail -4 $O/host_path_runtime_copies.log
timeout 300 python3 tools/host_async_check.py 4096 > $O/host_async_check.log 2>&1; tail -5 $O/host_async_check.log
bash tools/prof_host_async.sh > $O/host_async_timeline.txt 2>&1; tail -30 $O/host_async_timeline.txt
for cfg in c3 c2 c4; do
  timeout 600 python3 bench.py --config $cfg > $O/bench_$cfg.log 2>&1; tail -1 $O/bench_$cfg.log > $O/bench_$cfg.json
done
timeout 300 python3 tools/single_call.py --skip-c3 --out $O/single_sets.json > $O/single_sets.log 2>&1; tail -4 $O/single_sets.log
