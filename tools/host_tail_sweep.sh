# the PSF_* switches below are alive in the experiments build only (make -C tools_amd/csrc exp); the release library reads none of them
export PSF_LIB="${PSF_LIB:-$(cd "$(dirname "$0")/.." && pwd)/tools_amd/lib/libpsf_mi355x_exp.so}"
cd /root/repo
run() { echo "=== $*"; env "$@" timeout 300 python3 tools/host_path_timing.py 8 2>&1 | grep -E "synchronous|async|device-pointer|same rows|fresh" | tail -9; }
run PSF_HOST_TAIL=256
run PSF_HOST_TAIL=512
run PSF_HOST_TAIL=768
run PSF_HOST_TAIL=1024
run PSF_HOST_TAIL=2048
run PSF_HOST_TAIL=4096
run PSF_HOST_TAIL=512 PSF_HOST_WORKERS=8
run PSF_HOST_TAIL=512 PSF_HOST_CHUNK_MB=8
