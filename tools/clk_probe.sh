# the PSF_* switches below are alive in the experiments build only (make -C tools_amd/csrc exp); the release library reads none of them
export PSF_LIB="${PSF_LIB:-$(cd "$(dirname "$0")/.." && pwd)/tools_amd/lib/libpsf_mi355x_exp.so}"
export PSF_NP_ONE_STREAM=1 PSF_LIB=$PWD/tools_amd/lib/libpsf_np_profile.so TMPDIR=/tmp
R=$PWD
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_clk -o t --output-format csv -- python3 $R/tools/np_profile.py c2 > $R/gpurun_out/clk.log 2>&1
cd $R; grep -E "k_np_sample|k_np_gemm" gpurun_out/prof_clk/*kernel_stats.csv | cut -c1-200; tail -9 gpurun_out/clk.log; rm -rf gpurun_out/prof_clk
rocm-smi --showclocks 2>/dev/null | head -20
