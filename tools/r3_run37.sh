#!/bin/bash
O=gpurun_out/r3_run37; mkdir -p $O
for p in 3 2; do for c in c4 c2; do
echo "=== NP_PROFILE=$p $c" | tee -a $O/np_profile.log
PSF_LIB=$PWD/tools_amd/lib/libpsf_np_profile$p.so timeout 300 python3 tools/np_profile.py $c 2>&1 | grep -v amdgpu.ids | tee -a $O/np_profile.log
done; done
