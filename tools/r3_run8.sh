#!/bin/bash
O=gpurun_out/r3_run8; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_distribution.py tests/test_gpu_full_size.py -q -m gpu -k "ring_preimages or full_size_batch" 2>&1 | tail -15 > $O/tests.log
for c in c3prime c2s240 c2 c4; do timeout 600 python3 bench.py --config $c --steps 10 --warmup 2 2>$O/bench_$c.err | tail -1 > $O/bench_$c.json; done
cat $O/tests.log; for c in c3prime c2s240; do cut -c1-700 $O/bench_$c.json; done
