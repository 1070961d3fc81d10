"""The randomized nearest plane (MatZ::sample_d_precomputed_gso, gpv.rs:160; gpv_ring.rs:204-211) has three launch forms for the SAME chains: one launch per
block (k_np_step, PSF_NP_WALK=0), the whole walk in one launch with updater workgroups that keep the running projections in registers (k_np_walk, the default where
the batch fits) and the whole walk with the helper waves updating the projections in memory (k_np_walk2, PSF_NP_WALK=3), each with one or two preimages per
sampler wave (PSF_NP_G); the recombination e = sum z b behind it runs as one launch over the occupied tiles of the digit planes (default) or as one launch per
digit pair (PSF_NP_COMBINE=0).  Every form must return the default's bytes -- which equal the oracle's -- on batches with several 64-preimage groups (one of them
partial), several blocks and a short top block; a wait of the default one-launch form that gives up (a GPU shared with other work) is walked again inside the
call by k_np_walk_solo and costs time only."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, hashlib, json, math
sys.path.insert(0, %r)
import numpy as np
import tools_amd as T
from oracle import oracle as O
kind, B = sys.argv[1], int(sys.argv[2])
if kind == "gpv":
    n, q, s = 14, 2**9, 70.0
    psf = T.PSFGPV(T.GadgetParameters.init_default(n, q), s)
    psf.trap_gen(21)
    d = psf.m
else:
    n, q = 16, 3329
    s = ((2 * 2 * 1.005 * math.sqrt(n) + 1) * 2) * 4
    psf = T.PSFGPVRing(T.GadgetParametersRing.init_default(n, q), s, 1.005)
    psf.trap_gen(22)
    d = psf.d
u = O.uniform_targets(9, B, n, q)
try:
    outs = [psf.samp_p(u, seed=50 + i, first_index=777 * i) for i in range(2)]
except T.PsfError as ex:
    print(json.dumps({"status": ex.status}))
    sys.exit(0)
h = hashlib.sha256()
for e in outs:
    h.update(np.ascontiguousarray(e).tobytes())
print(json.dumps({"hash": h.hexdigest(), "d": int(d), "status": 0}))
''' % ROOT

FORMS = [{}, {"PSF_NP_WALK": "0"}, {"PSF_NP_WALK": "3"}, {"PSF_NP_G": "2"}, {"PSF_NP_WALK": "0", "PSF_NP_G": "2"}, {"PSF_NP_WALK": "3", "PSF_NP_G": "2"},
         {"PSF_NP_WALK": "0", "PSF_NP_IMMEDIATE": "0"}, {"PSF_NP_COMBINE": "0"}]


def run(kind, B, **extra):
    from tests.conftest import exp_env
    env = exp_env(**extra)                       # a switch = the experiments build of the library; no switch = the release library
    r = subprocess.run([sys.executable, "-c", SCRIPT, kind, str(B)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("kind,B", [("gpv", 150), ("gpv", 64), ("gpv", 5), ("ring", 200), ("ring", 129)])
def test_every_launch_form_of_the_walk_returns_the_same_bytes(kind, B):
    base = run(kind, B)
    assert base["status"] == 0 and base["d"] >= 192          # at least three blocks of 64 rows, the top one short
    for form in FORMS[1:]:
        got = run(kind, B, **form)
        assert got.get("hash") == base["hash"], (form, kind, B)


def test_the_default_form_equals_the_oracle(oracle):
    """(the parity tests of the two types run the default form at many shapes; this pins the shape of the form comparison above to the oracle as well)"""
    import numpy as np
    import tools_amd as T
    n, q, s = 14, 2**9, 70.0
    psf = T.PSFGPV(T.GadgetParameters.init_default(n, q), s)
    A, (bt, gt) = psf.trap_gen(21)
    orc = oracle.PSFGPV(oracle.gadget_params_default(n, q), s)
    orc.load_key(A, bt, gt)
    u = oracle.uniform_targets(9, 150, n, q)
    assert (psf.samp_p(u, seed=50) == orc.samp_p(50, u)).all()
    psf.close()


@pytest.mark.parametrize("kind,B", [("gpv", 1100), ("gpv", 2048), ("ring", 1025), ("ring", 1800), ("ring", 2100)])
def test_batches_between_one_and_two_thousand_take_one_preimage_per_wave(oracle, kind, B):
    """round 6: beyond 4 x CUs preimages the release library walks per block, with ONE preimage per sampler wave up to 2048 preimages (k_np_step<1>) and two beyond
    (k_np_step<2>); the one-launch walk with two per wave (k_np_walk<2>) left it.  The form the handle reports, and sampled rows against the oracle."""
    import math
    import tools_amd as T
    if kind == "gpv":
        n, q, s = 14, 2**9, 70.0
        psf = T.PSFGPV(T.GadgetParameters.init_default(n, q), s)
        A, (bt, gt) = psf.trap_gen(21)
        orc = oracle.PSFGPV(oracle.gadget_params_default(n, q), s)
        orc.load_key(A, bt, gt)
    else:
        n, q = 16, 3329
        s = ((2 * 2 * 1.005 * math.sqrt(n) + 1) * 2) * 4
        psf = T.PSFGPVRing(T.GadgetParametersRing.init_default(n, q), s, 1.005)
        psf.trap_gen(22)
        a, r, e, bt, gt = psf.export_key()
        orc = oracle.PSFGPVRing(oracle.gadget_params_ring_default(n, q), s, 1.005)
        orc.load_key(a, r, e, gso_t=gt)
    u = oracle.uniform_targets(9, B, n, q)
    got = psf.samp_p(u, seed=50, first_index=3)
    form, G, blocks, reruns = psf.nearest_plane_form()
    assert form == 0 and G == (1 if B <= 2048 else 2) and reruns == 0, (kind, B, form, G)
    for b in (0, 1, 1023, 1024, B // 2, B - 1):
        assert (got[b] == orc.samp_p(50, u[b:b + 1], first_index=3 + b)[0]).all(), (kind, B, b)
    psf.close()


def test_a_wait_that_gives_up_is_walked_again_inside_the_call(oracle):
    """A poll limit of 1 makes the first wait of the one-launch walk that is not satisfied at once give up: the abort word is raised, every workgroup leaves at its next
    wait, and the launches enqueued behind the walk (a fresh projection + k_np_walk_solo, no waits between workgroups) walk the batch again.  The call returns
    status 0 and the oracle's rows -- contention costs time, never the call (gpv.rs:152-161 never fails on a valid key) -- and the handle counts the re-run."""
    import math
    import tools_amd as T
    for kind, B in (("gpv", 150), ("gpv", 5), ("ring", 200)):
        if kind == "gpv":
            n, q, s = 14, 2**9, 70.0
            psf = T.PSFGPV(T.GadgetParameters.init_default(n, q), s)
            A, (bt, gt) = psf.trap_gen(21)
            orc = oracle.PSFGPV(oracle.gadget_params_default(n, q), s)
            orc.load_key(A, bt, gt)
        else:
            n, q = 16, 3329
            s = ((2 * 2 * 1.005 * math.sqrt(n) + 1) * 2) * 4
            psf = T.PSFGPVRing(T.GadgetParametersRing.init_default(n, q), s, 1.005)
            psf.trap_gen(22)
            a, r, e, bt, gt = psf.export_key()
            orc = oracle.PSFGPVRing(oracle.gadget_params_ring_default(n, q), s, 1.005)
            orc.load_key(a, r, e, gso_t=gt)
        u = oracle.uniform_targets(9, B, n, q)
        want = psf.samp_p(u, seed=50, first_index=3)
        assert psf.nearest_plane_form()[0] == 1 and psf.nearest_plane_form()[3] == 0          # the default form here is the one-launch walk; nothing re-run so far
        assert (want == orc.samp_p(50, u, first_index=3)).all()
        psf._debug_set_walk(1, spins=1)
        got = psf.samp_p(u, seed=50, first_index=3)                                           # no PsfError: status 0
        form, G, blocks, reruns = psf.nearest_plane_form()
        assert form == 1 and reruns >= 1, (kind, B, form, reruns)
        assert (got == want).all(), (kind, B)
        psf._debug_set_walk(0)                                                                # and the launch-per-block form agrees, as ever
        assert (psf.samp_p(u, seed=50, first_index=3) == want).all() and psf.nearest_plane_form()[0] == 0
        psf.close()


@pytest.mark.parametrize("walk", ["3"])
def test_a_wait_that_gives_up_reports_a_sampler_failure(walk):
    """the opt-in second one-launch form (PSF_NP_WALK=3, k_np_walk2: measured slower, kept for comparison) has no re-run behind it: PSF_NP_WALK_SPINS=1 makes its
    first unsatisfied wait end the call with PSF_ERR_SAMPLER (status 9) -- no hang, no silent rows."""
    got = run("gpv", 150, PSF_NP_WALK=walk, PSF_NP_WALK_SPINS="1")
    assert got["status"] == 9
    ok = run("gpv", 150, PSF_NP_WALK=walk)                   # and the same process configuration without the limit is fine
    assert ok["status"] == 0


@pytest.mark.parametrize("kind,B", [("gpv", 300), ("gpv", 257), ("gpv", 640), ("ring", 513), ("ring", 1000)])
def test_two_halves_side_by_side_return_the_rows_of_the_undivided_call(oracle, exp_lib, kind, B):
    """EXPERIMENTS build (measured neutral inside one call at C4, 4.416 -> 4.396 ms, and therefore not in the release library): a batch that runs one launch per block
    cut into two column ranges that walk side by side on two streams.  Forced here at small sizes (ragged halves: the first is a multiple of 128, the second whatever is left): the rows
    are the rows of the undivided call and of the oracle, through the synchronous and the asynchronous host entry points."""
    import math
    import numpy as np
    import tools_amd as T
    if kind == "gpv":
        n, q, s = 14, 2**9, 70.0
        psf = T.PSFGPV(T.GadgetParameters.init_default(n, q), s)
        A, (bt, gt) = psf.trap_gen(21)
        orc = oracle.PSFGPV(oracle.gadget_params_default(n, q), s)
        orc.load_key(A, bt, gt)
    else:
        n, q = 16, 3329
        s = ((2 * 2 * 1.005 * math.sqrt(n) + 1) * 2) * 4
        psf = T.PSFGPVRing(T.GadgetParametersRing.init_default(n, q), s, 1.005)
        psf.trap_gen(22)
        a, r, e, bt, gt = psf.export_key()
        orc = oracle.PSFGPVRing(oracle.gadget_params_ring_default(n, q), s, 1.005)
        orc.load_key(a, r, e, gso_t=gt)
    u = oracle.uniform_targets(9, B, n, q)
    psf._debug_set_walk(0)                                   # one launch per block (at C4's size the batch does not fit the one-launch walk by itself)
    psf._debug_set_split(0)
    whole = psf.samp_p(u, seed=50, first_index=3)
    assert psf._debug_last_parts() == 1
    psf._debug_set_split(1)
    halves = psf.samp_p(u, seed=50, first_index=3)
    assert psf._debug_last_parts() == 2
    assert (halves == whole).all()
    assert (whole[:40] == orc.samp_p(50, u[:40], first_index=3)).all() and (whole[-40:] == orc.samp_p(50, u[-40:], first_index=3 + B - 40)).all()
    out = [np.full_like(whole, -7) for _ in range(3)]
    for i in range(3):                                       # three asynchronous calls: two in flight, the halves of consecutive calls share the two streams
        psf.samp_p_async(u, out[i], seed=50 + i, first_index=3)
    psf.wait()
    assert (out[0] == whole).all()
    psf._debug_set_split(0)
    assert (psf.samp_p(u, seed=52, first_index=3) == out[2]).all()
    psf.close()
