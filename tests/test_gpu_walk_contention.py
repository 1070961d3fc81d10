"""The one-launch nearest-plane walk (k_np_walk, gpv.rs:160) on a GPU that is NOT idle.  Its workgroups wait for each other through device memory, so all of them must
be resident at once; the library decides that from the device's occupancy figures, which cannot know about other tenants.  What must hold (the
reference never fails on a valid key): another kernel holding compute units -- all of them, or all but a few so that only part of the walk's grid is dispatched --
costs time, never the call; two handles walking on two streams at once do not starve each other.  tests/cpp/liboccupy.so is a test helper (a kernel that holds
workgroup slots for a given time on a stream of its own), built by __graft_entry__.build()."""
import ctypes as C
import os
import subprocess
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OCC = os.path.join(ROOT, "tests", "cpp", "liboccupy.so")


@pytest.fixture(scope="module")
def occupy():
    src = os.path.join(ROOT, "tests", "cpp", "occupy.hip")
    if not os.path.exists(OCC) or os.path.getmtime(OCC) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", OCC, src])
    from tools_amd import _ffi
    _ffi.lib()                                   # one HIP runtime in the process: the product's loader maps it first
    L = C.CDLL(OCC)
    L.occupy_start.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.occupy_wait.argtypes = [C.c_void_p]
    L.occupy_done.argtypes = [C.c_void_p]
    return L


def _cus():
    from tools_amd import _ffi
    name = C.create_string_buffer(64)
    cus = C.c_int(0)
    assert _ffi.lib().psf_device_info(0, name, 64, C.byref(cus)) == 0
    return cus.value


def _gpv(oracle, B):
    import tools_amd as T
    n, q, s = 14, 2**9, 70.0
    psf = T.PSFGPV(T.GadgetParameters.init_default(n, q), s)
    A, (bt, gt) = psf.trap_gen(21)
    orc = oracle.PSFGPV(oracle.gadget_params_default(n, q), s)
    orc.load_key(A, bt, gt)
    u = oracle.uniform_targets(9, B, n, q)
    return psf, orc, u


@pytest.mark.parametrize("left_free", [0, 12])
def test_walk_beside_a_kernel_that_holds_the_compute_units(oracle, occupy, left_free):
    """left_free = 0: every workgroup slot of the chip is held for 60 ms -- the walk's grid is dispatched when the slots come back.  left_free = 12: all but twelve
    slots are held for 400 ms while the walk's poll limit is ~10 ms: the first few sampler workgroups become resident, wait for updaters that cannot be dispatched,
    give up -- and the call is walked again by the form without waits.  Either way: status 0, the oracle's rows."""
    psf, orc, u = _gpv(oracle, 150)
    want = orc.samp_p(50, u, first_index=3)
    assert (psf.samp_p(u, seed=50, first_index=3) == want).all()           # idle GPU first (buffers allocated, form = one launch)
    assert psf.nearest_plane_form()[0] == 1
    slots = 2 * _cus()                                                       # 512 threads + 64 KiB of LDS per workgroup: two per compute unit
    if left_free:
        psf._debug_set_walk(-1, spins=20000)
    st = C.c_void_p()
    assert occupy.occupy_start(0, 400 if left_free else 60, slots - left_free, 512, 64 * 1024, C.byref(st)) == 0
    time.sleep(0.005)                                                        # the holder is on the chip
    assert occupy.occupy_done(st) == 0
    t0 = time.time()
    got = psf.samp_p(u, seed=50, first_index=3)                              # PsfError (status 9) before this round
    took = time.time() - t0
    assert occupy.occupy_wait(st) == 0
    form, G, blocks, reruns = psf.nearest_plane_form()
    print(f"[contention] left_free={left_free}: call took {took * 1e3:.1f} ms beside the holder, walks re-run so far: {reruns}")
    assert (got == want).all()
    assert (psf.samp_p(u, seed=51, first_index=3) == orc.samp_p(51, u, first_index=3)).all()      # and the handle is fine afterwards
    psf.close()


def test_two_handles_walking_on_two_streams_take_turns(oracle):
    """Two PSFGPV handles, device-pointer calls on two non-blocking streams back to back, many times: each walk alone fits the chip, two at once would each hold part
    of the slots and wait for the rest.  One-launch walks of a process take turns per device (an event chain), so both always complete; rows = oracle."""
    import torch
    import tools_amd as T
    n, q, s, B = 14, 2**9, 70.0, 150
    hs, orcs = [], []
    for seed in (21, 22):
        psf = T.PSFGPV(T.GadgetParameters.init_default(n, q), s)
        A, (bt, gt) = psf.trap_gen(seed)
        orc = oracle.PSFGPV(oracle.gadget_params_default(n, q), s)
        orc.load_key(A, bt, gt)
        hs.append(psf); orcs.append(orc)
    u = oracle.uniform_targets(9, B, n, q)
    du = torch.from_numpy(u.astype(np.int64)).cuda()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[torch.empty((B, hs[0].m), dtype=torch.int64, device="cuda") for _ in range(6)] for _ in range(2)]
    torch.cuda.synchronize()
    for it in range(6):
        for i in range(2):
            hs[i].samp_p_dev(du.data_ptr(), outs[i][it].data_ptr(), B, seed=60 + it, first_index=0, stream=streams[i].cuda_stream)
    torch.cuda.synchronize()
    for i in range(2):
        assert hs[i].last_status() == 0
        for it in (0, 5):
            assert (outs[i][it].cpu().numpy() == orcs[i].samp_p(60 + it, u)).all(), (i, it)
        assert hs[i].nearest_plane_form()[3] == 0                            # nobody had to give up
        hs[i].close()
