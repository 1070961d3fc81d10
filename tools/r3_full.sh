#!/bin/bash
# full GPU suite, then the measurement pass
mkdir -p gpurun_out/r3_full
timeout 2400 python3 -m pytest tests -q -m gpu --durations=12 > gpurun_out/r3_full/tests.log 2>&1; tail -18 gpurun_out/r3_full/tests.log
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/r3_final.sh > gpurun_out/r3_full/final.log 2>&1; tail -5 gpurun_out/r3_full/final.log
