#!/bin/bash
# HBM-side traffic of the nearest-plane configurations through tools/bin/np_harness (C++ over the C ABI): FETCH_SIZE and WRITE_SIZE in separate --pmc passes,
# summed over every kernel of ONE samp_p call (k_np_solve, k_np_project, d/64 x k_np_step, k_np_combine8, k_np_add_sol) -- usage: tools/pmc_np.sh c2|c4
export TMPDIR=/tmp
cfg=$1
R=$PWD; O=$R/gpurun_out/pmc_np_$cfg; rm -rf $O; mkdir -p $O
cd /tmp
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $O/f -o t --output-format csv -- $R/tools/bin/np_harness $cfg 2 > $O/f.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $O/w -o t --output-format csv -- $R/tools/bin/np_harness $cfg 2 > $O/w.log 2>&1
cd $R
python3 - $O $cfg <<'PY'
import csv, glob, sys, json, collections
O, cfg = sys.argv[1:3]
tot = collections.defaultdict(float); per = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(int)
first_np = {}
for f in glob.glob(O + '/*/*counter_collection.csv'):
    rows = list(csv.DictReader(open(f)))
    # the harness runs 2 samp_p calls after key generation: keep the kernels of the LAST call = everything from the last k_np_solve dispatch on
    solve = [int(r['Dispatch_Id']) for r in rows if 'k_np_solve' in r['Kernel_Name']]
    if not solve: continue
    start = max(solve)
    for r in rows:
        if int(r['Dispatch_Id']) < start: continue
        name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('psf::', '')
        if not name.startswith('k_np') : continue
        tot[r['Counter_Name']] += float(r['Counter_Value'])
        per[name][r['Counter_Name']] += float(r['Counter_Value'])
        calls[name] += 1
out = {"config": cfg, "FETCH_SIZE_KiB": tot.get('FETCH_SIZE'), "WRITE_SIZE_KiB": tot.get('WRITE_SIZE'),
       "per_kernel_KiB": {k: dict(v) for k, v in per.items()}, "launches": dict(calls)}
if tot.get('FETCH_SIZE') is not None and tot.get('WRITE_SIZE') is not None:
    out["hbm_bytes_per_call"] = int(2 * tot['FETCH_SIZE'] * 1024 + tot['WRITE_SIZE'] * 1024)
print(json.dumps(out))
PY
tail -2 $O/f.log
