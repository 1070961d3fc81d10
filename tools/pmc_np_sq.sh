#!/bin/bash
# SQ counters of the nearest-plane kernels through tools/bin/np_harness (C++ over the C ABI): vector / matrix instruction counts and busy cycles per kernel -- usage: tools/pmc_np_sq.sh c2|c4
export TMPDIR=/tmp
cfg=$1
R=$PWD; O=$R/gpurun_out/pmc_np_sq_$cfg; rm -rf $O; mkdir -p $O
cd /tmp
timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS -d $O/a -o t --output-format csv -- $R/tools/bin/np_harness $cfg 2 > $O/a.log 2>&1
timeout 900 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA -d $O/b -o t --output-format csv -- $R/tools/bin/np_harness $cfg 2 > $O/b.log 2>&1
cd $R
python3 - $O <<'PY'
import csv, collections, glob, sys
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('psf::', '')
        if not k.startswith('k_np'): continue
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(agg, key=lambda k: -sum(agg[k].get('SQ_WAVE_CYCLES', [0]))):
    c = {n: sum(v) / len(v) for n, v in agg[k].items()}
    nl = len(next(iter(agg[k].values())))
    print(f"== {k}  (average per launch over {nl} launches)")
    for n in sorted(c): print(f"   {n:28s} {c[n]:.4g}")
PY
