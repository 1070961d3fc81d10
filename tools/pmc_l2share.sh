#!/bin/bash
# FETCH_SIZE per dispatch of tools/bin/probe_l2share, as multiples of the buffer size
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/pmc_l2share; rm -rf $O; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/f -o t --output-format csv -- $R/tools/bin/probe_l2share > $O/f.log 2>&1
cd $R
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
names = [l.strip() for l in open(O + '/f.log') if l.startswith('case:')]
rows = []
for f in glob.glob(O + '/f/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_stream' in r['Kernel_Name']: rows.append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
rows.sort()
for (d, v), nm in zip(rows, names):
    print(f"{nm[:100]:100s} FETCH_SIZE x2 = {2*v*1024/2**20:9.1f} MiB = {2*v*1024/(64*2**20):6.2f} x S")
PY
