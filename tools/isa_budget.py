"""Static instruction budget of one kernel from hipcc's device assembly: per basic block (label to label) the instruction count by issue class and the
issue cycles they cost one wave (gfx950: a vector instruction issues over 4 cycles, the 32-bit integer multiplies, v_mad_u64_u32 and the transcendental /
FP64-conversion group over 16; scalar, LDS and memory instructions 1 issue slot each).  The hot loop is picked by hand from the listing (block names below).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -S --cuda-device-only -o /tmp/psfp.s -x hip tools_amd/csrc/psfp.hip
    python tools/isa_budget.py /tmp/psfp.s _ZN3psf20k_perturb_round_lean
"""
import collections, re, sys

QUARTER = ("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mad_u64_u32", "v_mad_i64_i32", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_exp_f32", "v_log_f32", "v_rcp_f32",
           "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32", "v_div_fixup_f64", "v_div_fmas_f64", "v_div_scale_f64", "v_trig_preop_f64")
HALF = ("v_fma_f64", "v_mul_f64", "v_add_f64", "v_cvt_f64", "v_cvt_i32_f64", "v_cvt_u32_f64", "v_cvt_f32_f64", "v_floor_f64", "v_fract_f64", "v_rndne_f64", "v_ldexp_f64", "v_cmp_", "v_min_f64", "v_max_f64")


def klass(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(QUARTER): return "valu16"
    if op.startswith("v_") and ("_f64" in op and not op.startswith("v_cmp")): return "valu_f64"
    if op.startswith("v_"): return "valu4"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    return "other"


def main(path, prefix, detail=None):
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and l.rstrip().endswith(tuple(":")) or (l.startswith(prefix) and ":" in l.split(";")[0]))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
    blocks, cur = collections.OrderedDict(), "entry"
    blocks[cur] = []
    for l in lines[start + 1:end]:
        s = l.split(";")[0].strip()
        if not s: continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            cur = m.group(1); blocks[cur] = []; continue
        if s.startswith("."): continue
        blocks[cur].append(s)
    tot = collections.Counter()
    print(f"{'block':14s} {'n':>5s} {'valu4':>6s} {'f64':>5s} {'v16':>5s} {'salu':>5s} {'br':>4s} {'lds':>4s} {'vmem':>5s} {'wait':>5s}  issue cycles   back edge")
    for b, ins in blocks.items():
        c = collections.Counter(klass(i.split()[0]) for i in ins)
        cyc = 4 * c["valu4"] + 8 * c["valu_f64"] + 16 * c["valu16"] + c["salu"] + c["branch"] + c["lds"] + c["vmem"] + c["wait"]
        tgt = [i.split()[-1] for i in ins if i.startswith(("s_cbranch", "s_branch"))]
        if len(ins) >= 8 or detail:
            print(f"{b:14s} {len(ins):5d} {c['valu4']:6d} {c['valu_f64']:5d} {c['valu16']:5d} {c['salu']:5d} {c['branch']:4d} {c['lds']:4d} {c['vmem']:5d} {c['wait']:5d}  {cyc:8d}       {' '.join(tgt)}")
        tot.update(c)
        if detail and b in detail.split(","):
            ops = collections.Counter(i.split()[0] for i in ins)
            for op, k in ops.most_common(): print(f"      {k:4d} {op}")
    print("total", dict(tot))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
