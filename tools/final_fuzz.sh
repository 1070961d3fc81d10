#!/bin/bash
# Fuzz and soak of the round's final tree on the GPU box (release library): random configurations of the three types (wide menus), mid-size batches, single calls of random
# keys, host-pointer against device-pointer calls, soaks at C3 / C2 / C4.  usage: tools/final_fuzz.sh <tag> -> gpurun_out/<tag>_*.log; every step bounded by `timeout`.
tag=${1:-r06_fuzz3}
O=$PWD/gpurun_out; mkdir -p $O
timeout 900 python3 tools/fuzz_configs.py 120000 200 --wide > $O/${tag}_configs.log 2>&1
timeout 900 python3 tools/fuzz_midsize.py 6000 200 > $O/${tag}_midsize.log 2>&1
timeout 700 python3 tools/fuzz_fused_tail.py 3000 120 > $O/${tag}_fused_tail.log 2>&1
timeout 300 python3 tools/host_vs_device_fuzz.py 20000 1500 > $O/${tag}_host_vs_device.log 2>&1
timeout 300 python3 tools/host_vs_device_fuzz_gpv.py 20000 600 > $O/${tag}_host_vs_device_gpv.log 2>&1
timeout 400 python3 tools/soak.py c3 40 > $O/${tag}_soak_c3.log 2>&1
timeout 300 python3 tools/soak.py c2 100 > $O/${tag}_soak_c2.log 2>&1
timeout 300 python3 tools/soak.py c4 40 > $O/${tag}_soak_c4.log 2>&1
tail -n 2 $O/${tag}_*.log
