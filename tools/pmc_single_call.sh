#!/bin/bash
# HBM-side traffic of ONE C3 samp_p call with few preimages, per kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (no trace domains), KiB units,
# FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM section).  usage: tools/pmc_single_call.sh <batch>   -> one JSON line
export TMPDIR=/tmp
B=${1:-16}
R=$PWD; O=$R/gpurun_out/pmc_single_$B; rm -rf $O; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/f -o t --output-format csv -- python3 $R/tools/single_call.py --batches $B --reps 2 --skip-sets --out $O/f.json > $O/f.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/w -o t --output-format csv -- python3 $R/tools/single_call.py --batches $B --reps 2 --skip-sets --out $O/w.json > $O/w.log 2>&1
cd $R
python3 - $O $B <<'PY'
import csv, glob, sys, json, collections
O, B = sys.argv[1], int(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
keep = ("k_normals_wave", "k_trmm_stream", "k_perturb_round", "k_split_P", "k_zq_mfma", "k_zq_combine", "k_gadget", "k_recombine")
for f in glob.glob(O + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('psf::', '')
        if any(k in name for k in keep):
            agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
per = {}
tot = 0
for name, c in sorted(agg.items()):
    if 'FETCH_SIZE' not in c or 'WRITE_SIZE' not in c:
        continue
    # the streaming arm and the batch-kernel arm share every kernel but the product; averages over all launches of the run
    fetch = sum(c['FETCH_SIZE']) / len(c['FETCH_SIZE']); write = sum(c['WRITE_SIZE']) / len(c['WRITE_SIZE'])
    per[name] = {"FETCH_SIZE_KiB_avg": round(fetch, 1), "WRITE_SIZE_KiB_avg": round(write, 1), "hbm_bytes_per_launch": int(2 * fetch * 1024 + write * 1024), "launches": len(c['FETCH_SIZE'])}
    if 'k_trmm_f64_big' not in name:
        tot += per[name]["hbm_bytes_per_launch"]
print(json.dumps({"config": "c3", "batch": B, "per_kernel": per, "hbm_bytes_per_call_streaming_arm": tot,
                  "note": "FETCH_SIZE x 2 + WRITE_SIZE per launch (KiB -> bytes); k_trmm_f64_big appears only because the tool also times the batch-kernel arm"}))
PY
tail -2 $O/f.log | cut -c1-300
rm -rf $O
