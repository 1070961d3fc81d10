#!/bin/bash
# HBM-side traffic of the dominant kernel of a bench configuration from the PMC counters (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE in
# SEPARATE --pmc passes (no trace domains), KiB units, FETCH_SIZE doubled on gfx950 for 16 B/lane streams.  usage: tools/pmc_traffic.sh <config> <kernel substring> [bench args]
export TMPDIR=/tmp
cfg=$1; kern=$2; shift 2
R=$PWD; O=$R/gpurun_out/pmc_traffic_$cfg; mkdir -p $O
cd /tmp
rocprofv3 --pmc FETCH_SIZE -d $O/f -o t --output-format csv -- python3 $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline "$@" > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/w -o t --output-format csv -- python3 $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline "$@" > $O/w.log 2>&1
cd $R
python3 - $O "$kern" $cfg <<'PY'
import csv, glob, sys, json, collections
O, kern, cfg = sys.argv[1:4]
agg = collections.defaultdict(list)
for f in glob.glob(O + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if kern in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
res = {n: (sum(v) / len(v), len(v)) for n, v in agg.items()}
out = {"config": cfg, "kernel": kern, "launches": {n: c for n, (a, c) in res.items()}}
if 'FETCH_SIZE' in res and 'WRITE_SIZE' in res:
    fetch, write = res['FETCH_SIZE'][0], res['WRITE_SIZE'][0]
    out.update({"FETCH_SIZE_KiB_avg": fetch, "WRITE_SIZE_KiB_avg": write, "hbm_bytes_per_launch": int(2 * fetch * 1024 + write * 1024)})
print(json.dumps(out))
PY
rm -rf $O
