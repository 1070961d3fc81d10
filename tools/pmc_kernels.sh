#!/bin/bash
# SQ-level PMC passes (separate runs, --pmc only) for one bench configuration: instruction counts and wave-cycle breakdown per kernel.
# usage: tools/pmc_kernels.sh <config> <tag> [bench args]   ->  gpurun_out/<tag>_pmc_<config>.txt
export TMPDIR=/tmp
cfg=$1; tag=$2; shift 2
R=$PWD; O=$R/gpurun_out/pmc_$tag; mkdir -p $O
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SALU \
  -d $O/a -o t --output-format csv -- python3 $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline "$@" > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA \
  -d $O/b -o t --output-format csv -- python3 $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline "$@" > $O/b.log 2>&1
cd $R
python3 - $O $cfg $tag <<'PY'
import csv, collections, glob, sys
O, cfg, tag = sys.argv[1:4]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('psf::', '')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
out = open(f'gpurun_out/{tag}_pmc_{cfg}.txt', 'w')
for k in sorted(agg, key=lambda k: -sum(agg[k].get('SQ_WAVE_CYCLES', [0]))):
    c = {n: sum(v) / len(v) for n, v in agg[k].items()}
    nl = len(next(iter(agg[k].values())))
    if c.get('SQ_WAVE_CYCLES', 0) * nl < 1e7: continue
    print(f"== {k}  (average per launch over {nl} launches)", file=out)
    for n in sorted(c): print(f"   {n:28s} {c[n]:.4g}", file=out)
    wc, w = c.get('SQ_WAVE_CYCLES'), c.get('SQ_WAVES')
    if wc and w:
        print("   -> per wave: %.0f cycles (x4 = SQ clocks?), VALU %.0f  SALU %.0f  LDS %.0f  SMEM %.0f instructions" % (wc / w, c.get('SQ_INSTS_VALU', 0) / w, c.get('SQ_INSTS_SALU', 0) / w, c.get('SQ_INSTS_LDS', 0) / w, c.get('SQ_INSTS_SMEM', 0) / w), file=out)
        print("   -> of the wave cycles: parked %.1f %%, issue-stalled %.1f %%, issuing %.1f %%" % (100 * c.get('SQ_WAIT_ANY', 0) / wc, 100 * c.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc), file=out)
out.close()
print(open(f'gpurun_out/{tag}_pmc_{cfg}.txt').read())
PY
rm -rf $O
