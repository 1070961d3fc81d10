#!/bin/bash
# round 3, GPU call 1: co-issue probe, bit-equality of the re-ordered FP64 product, A/B against the round-2 build
O=gpurun_out/r3_run1; mkdir -p $O
timeout 600 tools/bin/probe_coissue 60 32 > $O/probe_coissue.log 2>&1
timeout 900 python3 -m pytest tests/test_gpu_pipeline_mode.py tests/test_gpu_psfp_parity.py -q -m gpu -x 2>&1 | tail -3 > $O/tests.log
bash tools/ab_lib.sh c3 tools/bin/libpsf_r2.so tools_amd/lib/libpsf_mi355x.so 3 > $O/ab_c3.log 2>&1
PSF_PIPELINE=1 timeout 300 python3 bench.py --config c3 --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/c3_pipeline.json
cat $O/tests.log $O/ab_c3.log; tail -40 $O/probe_coissue.log
