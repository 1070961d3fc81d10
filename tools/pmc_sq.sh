#!/bin/bash
# SQ-level PMC passes (separate runs, --pmc only) for the C3 step: where the waves' cycles go (parked / issue-stalled / issuing),
# MFMA pipe busy cycles, LDS bank conflicts.  Summary per kernel -> gpurun_out/pmc_sq/summary.txt
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/pmc_sq; mkdir -p $O
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 \
  -d $O/a -o t --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA \
  -d $O/b -o t --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/b.log 2>&1
cd $R
python3 - <<'PY' > gpurun_out/pmc_sq/summary.txt
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_sq/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('psf::', '')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
want = ['k_trmm_f64_big', 'k_perturb_round_tab', 'k_perturb_round_lean', 'k_perturb_round_wave', 'k_gadget_queue', 'k_normals_wave', 'k_recombine_mfma', 'k_zq_mfma<4>', 'k_gadget_queue<true>', 'k_zq_mfma<4, false, true>']
for k in want:
    if k not in agg: continue
    c = {n: sum(v) / len(v) for n, v in agg[k].items()}
    print(f"== {k}  (average per launch over {len(next(iter(agg[k].values())))} launches)")
    for n in sorted(c): print(f"   {n:34s} {c[n]:.4g}")
    wc = c.get('SQ_WAVE_CYCLES')
    if wc:
        print("   -> of the wave cycles: parked %.1f %%, issue-stalled %.1f %%, issuing %.1f %%" % (100 * c.get('SQ_WAIT_ANY', 0) / wc, 100 * c.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc))
    if c.get('SQ_BUSY_CYCLES') and c.get('SQ_VALU_MFMA_BUSY_CYCLES'):
        print("   -> MFMA busy / SQ busy cycles: %.3f" % (c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES']))
    if c.get('SQ_LDS_IDX_ACTIVE'):
        print("   -> LDS bank-conflict cycles / LDS active cycles: %.4f" % (c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']))
PY
cat gpurun_out/pmc_sq/summary.txt
rm -rf $O/a $O/b
