#!/bin/bash
# A/B of the host-pointer path (C3, batch 4096): how a chunk crosses PCIe (PSF_HOST_COPY = sdma | runtime | kernel:N), asynchronous calls sliced / unsliced
# (PSF_HOST_ASYNC_SLICE), worker threads, host-side widening (streaming stores / PSF_HOST_PLAIN_WIDEN / none: PSF_HOST_DEBUG=1; no copies either: =2).
#   bash tools/host_copy_ab.sh > gpurun_out/host_copy_ab.log
# the PSF_* switches below are alive in the experiments build only (make -C tools_amd/csrc exp); the release library reads none of them
export PSF_LIB="${PSF_LIB:-$(cd "$(dirname "$0")/.." && pwd)/tools_amd/lib/libpsf_mi355x_exp.so}"
cd "$(dirname "$0")/.."
nproc; tools/bin/probe_widen 4; tools/bin/probe_widen 8
run() { echo "=== $*"; env "$@" timeout 300 python3 tools/host_path_timing.py 8 2>&1 | grep -E "synchronous|async|device-pointer|same rows" | tail -8; }
run PSF_HOST_COPY=sdma
run PSF_HOST_COPY=sdma PSF_HOST_ASYNC_SLICE=0
run PSF_HOST_COPY=sdma PSF_HOST_WORKERS=8
run PSF_HOST_COPY=sdma PSF_HOST_WORKERS=8 PSF_HOST_ASYNC_SLICE=0
run PSF_HOST_COPY=sdma PSF_HOST_WORKERS=2 PSF_HOST_ASYNC_SLICE=0
run PSF_HOST_COPY=sdma PSF_HOST_CHUNK_MB=64 PSF_HOST_ASYNC_SLICE=0
run PSF_HOST_COPY=sdma PSF_HOST_DEBUG=1
run PSF_HOST_COPY=runtime
run PSF_HOST_COPY=kernel:32
run PSF_HOST_COPY=runtime PSF_HOST_DEBUG=2
echo "=== full-size row check of overlapped calls (sdma)"
PSF_HOST_COPY=sdma timeout 300 python3 tools/host_async_check.py 4096 2>&1 | tail -5
PSF_HOST_COPY=sdma PSF_HOST_ASYNC_SLICE=0 timeout 300 python3 tools/host_async_check.py 4096 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_pipeline_mode.py tests/test_gpu_boundary_completion.py -q -m gpu 2>&1 | tail -3
