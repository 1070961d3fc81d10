#!/bin/bash
# Round-5 soak: full-size batches (batch kernels) and the single-call regime of the same keys (streaming product, wave gadget kernels, split-K stages) with fresh
# seeds per iteration; invariants on every row, a slice bit for bit against the oracle.
O=gpurun_out/r5_soak; mkdir -p $O
{ timeout 900 python3 tools/soak.py c3 30 32; timeout 900 python3 tools/soak.py psfp:512:1073741824:9:512:1 300 1; timeout 900 python3 tools/soak.py psfp:512:1073741824:9:512:16 200 16;
  timeout 900 python3 tools/soak.py psfp:512:1073741824:9:512:64 100 32; timeout 600 python3 tools/soak.py psfp:512:1073741824:9:512:700 30 32;
  timeout 600 python3 tools/soak.py psfp:8:128:3:30:1 2000 1; timeout 600 python3 tools/soak.py psfp:15:157:3.9:40:64 300 64; timeout 600 python3 tools/soak.py psfp:64:128:6:100:300 200 64;
  timeout 900 python3 tools/soak.py c2 30 16; timeout 900 python3 tools/soak.py c4 30 16; } > $O/soak.log 2>&1
grep -i "SOAK\|iteration\|error" $O/soak.log | tail -30
