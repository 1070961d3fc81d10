#!/bin/bash
# round 6, first GPU call: the new tests, then the whole GPU suite, the default bench line (headline + other_configs), the launcher route and the multi-handle mode
set -u
mkdir -p gpurun_out/r6a
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_np_forms.py tests/test_gpu_walk_contention.py tests/test_gpu_golden_product.py tests/test_gpu_ntt.py -x -q -m gpu -s > gpurun_out/r6a/new_tests.log 2>&1
echo "new tests rc=$?" | tee -a gpurun_out/r6a/summary.txt
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r6a/gpu_tests.log 2>&1
echo "gpu suite rc=$?" | tee -a gpurun_out/r6a/summary.txt
tail -3 gpurun_out/r6a/gpu_tests.log | tee -a gpurun_out/r6a/summary.txt
timeout 600 python bench.py > gpurun_out/r6a/bench_default.json 2> gpurun_out/r6a/bench_default.err
echo "bench default rc=$?" | tee -a gpurun_out/r6a/summary.txt
timeout 600 python bench.py --gpus 1 --force-dist --steps 5 --warmup 1 --no-cpu-baseline --no-latency > gpurun_out/r6a/bench_force_dist.json 2> gpurun_out/r6a/bench_force_dist.err
echo "bench force-dist (self-launched) rc=$?" | tee -a gpurun_out/r6a/summary.txt
timeout 600 python bench.py --multi-handle --gpus 1 --steps 3 --warmup 1 > gpurun_out/r6a/bench_multi_handle.json 2> gpurun_out/r6a/bench_multi_handle.err
echo "bench multi-handle rc=$?" | tee -a gpurun_out/r6a/summary.txt
timeout 60 python bench.py --gpus 2 > gpurun_out/r6a/bench_gpus2.out 2>&1
echo "bench --gpus 2 on a 1-GPU box rc=$? (2 expected)" | tee -a gpurun_out/r6a/summary.txt
tail -c 1500 gpurun_out/r6a/bench_default.json
