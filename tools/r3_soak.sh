#!/bin/bash
O=gpurun_out/r3_soak; mkdir -p $O
{ timeout 900 python3 tools/soak.py c3 60 32; timeout 600 python3 tools/soak.py psfp:64:128:6:100:300 200 64; timeout 600 python3 tools/soak.py psfp:20:257:4:120:1000 100 64; timeout 900 python3 tools/soak.py c2 40 16; timeout 900 python3 tools/soak.py c4 40 16; } > $O/soak.log 2>&1
tail -25 $O/soak.log
