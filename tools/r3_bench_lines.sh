#!/bin/bash
# the bench lines of every configuration (with CPU legs), after profiles/*_traffic.json have been refreshed for the current kernel sources
O=gpurun_out/r3_lines; rm -rf $O; mkdir -p $O
for cfg in c3 c2 c4 c3prime c2s240; do
  timeout 600 python3 bench.py --config $cfg --steps 20 --warmup 2 > $O/bench_$cfg.log 2>&1; tail -1 $O/bench_$cfg.log > $O/bench_$cfg.json
done
timeout 300 python3 bench.py --config c3 --structured --steps 20 --warmup 2 > $O/bench_c3s.log 2>&1; tail -1 $O/bench_c3s.log > $O/bench_c3_structured.json
timeout 900 python3 bench.py --config c5 --steps 2 --warmup 1 > $O/bench_c5.log 2>&1; tail -1 $O/bench_c5.log > $O/bench_c5_one_gpu.json
timeout 300 python3 bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_default.json
ls -la $O
