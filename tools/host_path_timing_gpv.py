"""Host-pointer against device-pointer samp_p for PSFGPV (C2: n=256, q=3329, s=1024, batch 1024) and PSFGPVRing (C4: batch 4096)."""
# the PSF_* switches this script sets are alive in the experiments build only (make -C tools_amd/csrc exp); the release library reads none of them
import os as _os
_os.environ.setdefault("PSF_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tools_amd", "lib", "libpsf_mi355x_exp.so"))
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools_amd as T

def run(name, psf, B, m, n, q):
    rng = np.random.default_rng(1)
    u = rng.integers(0, q, size=(B, n), dtype=np.uint64)
    dev = torch.device("cuda:0")
    ud = torch.from_numpy(u.astype(np.int64)).to(dev)
    ed = torch.empty((B, m), dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    psf.samp_p_dev(ud.data_ptr(), ed.data_ptr(), B, seed=1, stream=st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10):
        psf.samp_p_dev(ud.data_ptr(), ed.data_ptr(), B, seed=2 + i, stream=st)
    torch.cuda.synchronize(); dd = (time.perf_counter() - t0) / 10
    e = psf.samp_p(u, seed=1)
    ts = []
    for i in range(6):
        t0 = time.perf_counter(); e = psf.samp_p(u, seed=2 + i, out=e); ts.append(time.perf_counter() - t0)
    os.environ["PSF_HOST_STRAIGHT"] = "1"
    ts2 = []
    for i in range(4):
        t0 = time.perf_counter(); psf.samp_p(u, seed=2 + i, out=e); ts2.append(time.perf_counter() - t0)
    os.environ.pop("PSF_HOST_STRAIGHT")
    print(f"{name}: straight form (two allocations, pageable copies) {min(ts2)*1e3:.2f} ms into the same reused buffer")
    e = psf.samp_p(u, seed=7, out=e)
    psf.samp_p_dev(ud.data_ptr(), ed.data_ptr(), B, seed=7, stream=st); torch.cuda.synchronize()
    same = bool((ed.cpu().numpy().reshape(-1) == np.asarray(e).reshape(-1)).all())
    # a loop of overlapped asynchronous calls (samp_p_async / wait): two output buffers in turn
    outs = [np.empty_like(e), np.empty_like(e)]
    NC = 16
    psf.samp_p_async(u, outs[0], seed=50); psf.samp_p_async(u, outs[1], seed=51); psf.wait()
    t0 = time.perf_counter()
    for i in range(NC):
        psf.samp_p_async(u, outs[i & 1], seed=100 + i, first_index=3 * i)
    psf.wait()
    da = (time.perf_counter() - t0) / NC
    psf.samp_p_dev(ud.data_ptr(), ed.data_ptr(), B, seed=100 + NC - 1, first_index=3 * (NC - 1), stream=st); torch.cuda.synchronize()
    same_async = bool((ed.cpu().numpy().reshape(-1) == outs[(NC - 1) & 1].reshape(-1)).all())
    print(f"{name}: {NC} overlapped asynchronous calls {da*1e3:.2f} ms per call = {dd/da:.3f} of device-resident, last call's rows equal the device path's: {same_async}", flush=True)
    print(f"{name}: device pointers {dd*1e3:.2f} ms per call, host pointers {min(ts)*1e3:.2f} ms (median {sorted(ts)[3]*1e3:.2f}), same rows {same}", flush=True)

gpv = T.PSFGPV(T.GadgetParameters.init_default(256, 3329), 1024.0); gpv.trap_gen(3, export=False)
run("C2 PSFGPV batch 1024", gpv, 1024, gpv.m, 256, 3329)
gpv.close()
n = 256; s = ((2 * 2 * 1.005 * math.sqrt(n) + 1) * 2) * 4
ring = T.PSFGPVRing(T.GadgetParametersRing.init_default(n, 3329), s, 1.005); ring.trap_gen(4)
try:
    run("C4 PSFGPVRing batch 4096", ring, 4096, ring.d, ring.n if hasattr(ring, "n") else n, 3329)
except Exception as ex:
    print("ring:", type(ex).__name__, ex)
