#!/bin/bash
# Wrap-up of round 4 on the final tree: the whole GPU suite, the host-pointer timings (large and small calls), the bench lines of the three headline configurations.
cd "$(dirname "$0")/.."
O=gpurun_out/r4_wrap; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu --durations=5 > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 300 python3 tools/host_path_timing.py 32 > $O/host_path.log 2>&1; tail -4 $O/host_path.log
timeout 600 python3 tools/single_call.py --batches 1,4,16,64 --out $O/single_call.json > $O/single_call.log 2>&1; grep -E "^\[c3\]|^\[set\]" $O/single_call.log
for cfg in c3 c2 c4; do timeout 600 python3 bench.py --config $cfg > $O/bench_$cfg.log 2>&1; tail -1 $O/bench_$cfg.log > $O/bench_$cfg.json; done
python3 -c "
import json
for c in ('c3','c2','c4'):
    d=json.load(open('$O/bench_%s.json'%c)); print(c, d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'])
"
