#!/bin/bash
O=gpurun_out/r3_run39; mkdir -p $O
timeout 300 python3 bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernels_ms'])" | tee -a $O/log.txt
timeout 900 python3 -m pytest tests/test_gpu_psfp_parity.py tests/test_gpu_full_size.py tests/test_gpu_general_base.py tests/test_gpu_pipeline_mode.py -q -m gpu -x 2>&1 | tail -5
