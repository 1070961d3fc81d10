#!/bin/bash
# Last checks of round 4 on the final tree: smoke(), the default bench line (the driver's command), the bench lines whose `roofline.traffic` needs the traffic
# records of the pass before, soak on the final kernels (incl. 2-4 preimage calls: the streaming stage kernels), random configurations.
cd "$(dirname "$0")/.."
O=gpurun_out/r4_last; mkdir -p $O
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 900 python3 bench.py ) > $O/bench_default.log 2>&1; tail -4 $O/bench_default.log | cut -c1-400
for cfg in c3 c3prime c2s240; do timeout 600 python3 bench.py --config $cfg > $O/bench_$cfg.log 2>&1; tail -1 $O/bench_$cfg.log > $O/bench_$cfg.json; done
timeout 300 python3 bench.py --config c3 --structured > $O/bench_c3s.log 2>&1; tail -1 $O/bench_c3s.log > $O/bench_c3_structured.json
timeout 900 python3 bench.py --config c5 --steps 2 --warmup 1 > $O/bench_c5.log 2>&1; tail -1 $O/bench_c5.log > $O/bench_c5_one_gpu.json
{ timeout 900 python3 tools/soak.py c3 20 32; timeout 900 python3 tools/soak.py psfp:512:1073741824:9:512:1 150 1; timeout 900 python3 tools/soak.py psfp:512:1073741824:9:512:2 100 2;
  timeout 900 python3 tools/soak.py psfp:512:1073741824:9:512:3 80 3; timeout 900 python3 tools/soak.py psfp:512:1073741824:9:512:4 80 4; timeout 900 python3 tools/soak.py psfp:512:1073741824:9:512:8 60 8;
  timeout 600 python3 tools/soak.py psfp:64:128:6:100:1 1000 1; timeout 600 python3 tools/soak.py psfp:64:128:6:100:4 400 4; timeout 600 python3 tools/soak.py psfp:40:65536:4:120:3 400 3;
  timeout 600 python3 tools/soak.py c2 10 16; timeout 600 python3 tools/soak.py c4 10 16; } > $O/soak.log 2>&1
grep "SOAK" $O/soak.log
timeout 1500 python3 tools/fuzz_configs.py 60000 300 > $O/fuzz_a.log 2>&1; tail -2 $O/fuzz_a.log
timeout 1500 python3 tools/fuzz_configs.py 61000 200 --wide > $O/fuzz_b.log 2>&1; tail -2 $O/fuzz_b.log
