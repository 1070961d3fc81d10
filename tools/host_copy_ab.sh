#!/bin/bash
# A/B of the D2H mechanism of the host-pointer path (C3, batch 4096): the runtime's copies against a kernel storing into the pinned chunk buffers
# (PSF_HOST_COPY_KERNEL = workgroups of that kernel), under the three queue-priority settings.   bash tools/host_copy_ab.sh > gpurun_out/host_copy_ab.log
cd "$(dirname "$0")/.."
for prio in 1 0 2; do
  for grid in 0 8 32 128 512; do
    echo "=== PSF_HOST_PRIO=$prio PSF_HOST_COPY_KERNEL=$grid"
    PSF_HOST_PRIO=$prio PSF_HOST_COPY_KERNEL=$grid timeout 300 python3 tools/host_path_timing.py 8 2>&1 | grep -E "synchronous|async|device-pointer|same rows" | tail -8
  done
done
