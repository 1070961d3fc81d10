cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_single_call.py tests/test_gpu_psfp_parity.py tests/test_gpu_general_base.py tests/test_gpu_random_configs.py tests/test_gpu_switch_matrix.py tests/test_gpu_pipeline_mode.py -q -m gpu -x 2>&1 | tail -5
python3 tools/single_call.py --batches 1,2,3,4,5,16 --skip-sets --reps 20 --out gpurun_out/single_small_on.json 2>&1 | grep -E "^\[c3\]|kernels"
PSF_RECOMBINE_SMALL=0 PSF_SYNDROME_SMALL=0 python3 tools/single_call.py --batches 1,2,4 --skip-sets --reps 20 --out gpurun_out/single_small_off.json 2>&1 | grep -E "^\[c3\]|kernels"
timeout 300 python3 tools/host_path_timing.py 32 2>&1 | tail -9
