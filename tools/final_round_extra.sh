#!/bin/bash
# second part of the round-end pass: key-generation timeline at C3, the C5 shape on one GPU (bench line with its trap_gen time), the fused-tail A/B, the C4 split A/B,
# the whole GPU suite.  usage: tools/final_round_extra.sh <tag>; outputs under gpurun_out/<tag>_final/
tag=${1:-r06}
R=$PWD; O=$R/gpurun_out/${tag}_final; mkdir -p $O
bash tools/keygen_timeline.sh c3 $tag > /dev/null 2>&1; cp gpurun_out/${tag}_keygen_timeline_c3.txt $O/keygen_timeline_c3.txt
timeout 900 python3 bench.py --config c5 --steps 2 --warmup 1 --no-latency > $O/bench_c5.log 2>&1; tail -1 $O/bench_c5.log > $O/bench_c5_one_gpu.json
bash tools/fused_tail_ab.sh > $O/fused_tail_ab.log 2>&1
bash tools/c4_split_ab.sh > $O/c4_split_ab.log 2>&1
timeout 1500 python3 -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1
tail -3 $O/gpu_tests.log; head -12 $O/keygen_timeline_c3.txt; tail -c 700 $O/bench_c5_one_gpu.json; echo; cat $O/fused_tail_ab.log $O/c4_split_ab.log | grep -v amdgpu
