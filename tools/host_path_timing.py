"""Wall time of the host-pointer entry points (what the Rust shim binds: shim/src/lib.rs PSF::samp_p / samp_p_batch) against the device-pointer one, C3 shape:
psfp_samp_p (one synchronous call: first-call latency), a loop of psfp_samp_p_async calls with one psfp_wait at the end (steady state: the rows of call i cross PCIe
and are widened while call i + 1 computes), and psfp_samp_p_dev (device-resident, the figure bench.py reports)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
import tools_amd as T
from tools_amd._ffi import lib, check

n, q, r, s, B = 512, 2**30, 9.0, 512.0, 4096
NCALL = int(sys.argv[1]) if len(sys.argv) > 1 else 8
psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s)
check(lib().psfp_trap_gen(psf._h, C.c_uint64(3)), "trap_gen")
rng = np.random.default_rng(1)
u = rng.integers(0, q, size=(B, n), dtype=np.uint64)
rec = {"config": "c3", "batch": B}
e2 = np.ones((B, psf.m), dtype=np.int64)           # caller-owned, already touched output buffers (what a caller reuses)
e3 = np.ones((B, psf.m), dtype=np.int64)
ts = []
for i in range(4):
    t0 = time.perf_counter()
    psf.samp_p(u, seed=20 + i, out=e2)
    ts.append(time.perf_counter() - t0)
    print(f"host-pointer psfp_samp_p (synchronous) into a reused buffer: {ts[-1]*1e3:.1f} ms  ({B/ts[-1]:.0f} preimages/s)", flush=True)
rec["sync_call_ms"] = [round(t * 1e3, 2) for t in ts]
t0 = time.perf_counter()
fresh = psf.samp_p(u, seed=30)
print(f"host-pointer psfp_samp_p into freshly allocated memory: {(time.perf_counter()-t0)*1e3:.1f} ms", flush=True)
# steady state: asynchronous calls back to back, two output buffers in turn
outs = [e2, e3]
psf.samp_p_async(u, outs[0], seed=40); psf.wait()
t0 = time.perf_counter()
for i in range(NCALL):
    psf.samp_p_async(u, outs[i & 1], seed=50 + i)
psf.wait()
dt = (time.perf_counter() - t0) / NCALL
rec["async_steady_ms_per_call"] = round(dt * 1e3, 2)
print(f"host-pointer psfp_samp_p_async x {NCALL} + psfp_wait: {dt*1e3:.2f} ms per call  ({B/dt:.0f} preimages/s)", flush=True)
dev = torch.device("cuda:0")
ud = torch.from_numpy(u.astype(np.int64)).to(dev)
ed = torch.empty((B, psf.m), dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
psf.samp_p_dev(ud.data_ptr(), ed.data_ptr(), B, seed=9, stream=st)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(NCALL):
    psf.samp_p_dev(ud.data_ptr(), ed.data_ptr(), B, seed=50 + i, stream=st)
torch.cuda.synchronize(); dd = (time.perf_counter() - t0) / NCALL
rec["device_resident_ms_per_call"] = round(dd * 1e3, 2)
rec["async_over_device_resident"] = round(dd / dt, 4)
print(f"device-pointer psfp_samp_p_dev x {NCALL}: {dd*1e3:.2f} ms per call; host-pointer steady state runs at {dd/dt:.3f} of it", flush=True)
last = outs[(NCALL - 1) & 1]
print("same rows as the device-resident call with the same seed:", bool((ed.cpu().numpy() == last).all()))
print(json.dumps(rec))
