#!/bin/bash
# FETCH_SIZE (KiB per dispatch, L2 misses) of every kernel of tools/bin/probe_trmm, one dispatch per variant.  usage: tools/pmc_probe_trmm.sh [mask]
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/pmc_probe; rm -rf $O; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/f -o t --output-format csv -- $R/tools/bin/${PROBE:-probe_trmm} 240 32 1 ${1:-0x3800} ${2:-8} ${3:-8} ${4:-0} > $O/f.log 2>&1
cd $R
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
for f in glob.glob(O + '/f/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'trmm' not in r['Kernel_Name'] and 'variant' not in r['Kernel_Name']: continue
        v = float(r['Counter_Value'])
        print(f"{r['Kernel_Name'][:60]:60s} {r['Counter_Name']} {v:14.0f} KiB  x2 (gfx950 128-B requests) = {2*v*1024/1e9:8.2f} GB")
PY
grep -v "^running" $O/f.log | cut -c1-160 | tail -8
