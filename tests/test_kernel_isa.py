"""The default FP64 product k_trmm_f64_big -- and k_chol_update_big, the same K loop run on two row blocks of the key's chunk stream -- rely on orderings the compiler cannot see (asm-statement MFMAs, loads whose results are
in flight, a hand-written drain before the accumulators are read).  This test compiles the device code for gfx950 (no GPU needed) and
checks the EMITTED instruction stream, so that a toolchain bump which breaks one of the assumptions fails here and not as a silent
wrong bit on the GPU (psf_kernels.hpp).  If it fails: PSF_TRMM_VARIANT=1 (k_trmm_f64_reg, builtin MFMAs) is the fallback."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


KERNELS = {"k_trmm_f64_big": r"_ZN3psf14k_trmm_f64_big\w+", "k_chol_update_big": r"_ZN3psf17k_chol_update_big\w+"}


@pytest.fixture(scope="module", params=sorted(KERNELS))
def big_isa(request, device_asm):
    m = re.search(r"^(" + KERNELS[request.param] + r"):.*?^\s*\.end_amdhsa_kernel", device_asm, re.S | re.M)
    assert m, request.param + " not found in the device assembly"
    body = m.group(0)
    ins = [ln.strip() for ln in body.split("\n") if ln.strip() and not ln.strip().startswith(";") and not ln.strip().startswith(".")
           or ln.strip().startswith(".LBB")]
    return body, ins


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("no hipcc on this host")
    out = tmp_path_factory.mktemp("isa") / "psfp.s"
    src = os.path.join(ROOT, "tools_amd", "csrc", "psfp.hip")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                           "--cuda-device-only", "-S", "-o", str(out), "-x", "hip", src], stderr=subprocess.DEVNULL)
    return out.read_text()


def _inner_loop(ins):
    """the loop that holds the MFMAs: from its label to the backward branch"""
    labels = {ln[:-1].split(":")[0]: i for i, ln in enumerate(ins) if ln.startswith(".LBB")}
    for i, ln in enumerate(ins):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            seg = ins[labels[m.group(1)] + 1:i + 1]
            if any(s.startswith("v_mfma_f64") for s in seg):
                return seg
    raise AssertionError("no loop with MFMAs found")


def test_no_scratch_and_register_budget(big_isa):
    body, _ = big_isa
    assert "scratch_" not in body and re.search(r"\.amdhsa_private_segment_fixed_size 0\b", body)
    nv = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
    off = int(re.search(r"\.amdhsa_accum_offset (\d+)", body).group(1))
    assert nv - off == 256, "the 32 accumulator tiles must live in 256 AccVGPRs"
    assert nv <= 512 and off <= 128


def test_inner_loop_is_mfma_loads_waits_and_scalar_code_only(big_isa):
    _, ins = big_isa
    loop = _inner_loop(ins)
    ops = [ln.split()[0] for ln in loop]
    allowed = re.compile(r"^(v_mfma_f64_16x16x4_f64|global_load_dwordx2|s_\w+)$")
    bad = [o for o in ops if not allowed.match(o)]
    assert not bad, f"unexpected instructions in the K loop (copies / spills / accumulator moves?): {sorted(set(bad))}"
    n_mfma, n_load = ops.count("v_mfma_f64_16x16x4_f64"), ops.count("global_load_dwordx2")
    assert n_mfma % 32 == 0 and n_mfma >= 32 and n_load * 32 == n_mfma * 12      # 12 loads per 32 MFMAs (one k-step)
    assert ops.count("s_waitcnt") * 32 == n_mfma                                  # one vmcnt wait per k-step, in front of its MFMAs


def test_no_operand_is_read_while_its_load_is_in_flight(big_isa):
    """Walk the K loop twice (steady state): a VGPR pair that a global_load has been issued into must not be an MFMA source before the
    s_waitcnt that covers that load; with TR_BIG_PD = 4 a slot's loads are covered by the fourth wait after them (vmcnt(36) = three
    younger k-steps outstanding)."""
    _, ins = big_isa
    loop = _inner_loop(ins)
    seq = loop + loop
    pending = {}                                   # register pair -> number of waits seen since its load
    for ln in seq:
        op = ln.split()[0]
        if op == "s_waitcnt":
            assert "vmcnt(36)" in ln, ln
            for k in list(pending):
                pending[k] += 1
                if pending[k] >= 4:
                    del pending[k]
        elif op == "global_load_dwordx2":
            dst = re.match(r"global_load_dwordx2 (v\[\d+:\d+\])", ln).group(1)
            pending[dst] = 0
        elif op.startswith("v_mfma"):
            srcs = re.findall(r"v\[\d+:\d+\]", ln)
            for sreg in srcs:
                assert sreg not in pending, f"{ln}: reads {sreg} before the wait that covers its load"


def test_accumulators_are_read_only_after_the_drain(big_isa):
    """gfx950 stores straight from AccVGPRs (global_store ... a[..]); whatever reads an accumulator first -- a store or a v_accvgpr_read --
    must come after the last MFMA AND after the five s_nop 15 that let it retire (the hazard recogniser does not see asm MFMAs)."""
    _, ins = big_isa
    last_mfma = max(i for i, ln in enumerate(ins) if ln.startswith("v_mfma_f64"))
    acc_use = re.compile(r"\ba(\d+|\[\d+:\d+\])")
    first_mfma = min(i for i, ln in enumerate(ins) if ln.startswith("v_mfma_f64"))      # (before it: the zero-initialisation of the tiles)
    # v_accvgpr_mov / _write only appear where the tiles are zeroed (also in the block that skips an empty K loop, which sits after the loop in the text)
    readers = [i for i, ln in enumerate(ins) if i > first_mfma and not ln.startswith(("v_mfma", "v_accvgpr_mov", "v_accvgpr_write")) and acc_use.search(ln.split(";")[0])]
    assert readers, "no instruction reads an accumulator?"
    assert not [i for i in readers if i < last_mfma], "an accumulator is touched by a non-MFMA instruction inside the product"
    first_read = min(readers)
    between = ins[last_mfma + 1:first_read]
    nops = [ln for ln in between if ln.startswith("s_nop 15")]
    assert len(nops) >= 5, "the 80 wait states that let the last MFMA retire must sit between the last MFMA and the first accumulator read"
