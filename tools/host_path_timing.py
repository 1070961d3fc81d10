"""Wall time of the host-pointer entry point psfp_samp_p (what a Rust shim calls) against the device-pointer one, C3 shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
import tools_amd as T
from tools_amd._ffi import lib, check

n, q, r, s, B = 512, 2**30, 9.0, 512.0, 4096
psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s)
check(lib().psfp_trap_gen(psf._h, C.c_uint64(3)), "trap_gen")
rng = np.random.default_rng(1)
u = rng.integers(0, q, size=(B, n), dtype=np.uint64)
for i in range(3):
    t0 = time.perf_counter()
    e = psf.samp_p(u, seed=10 + i)
    dt = time.perf_counter() - t0
    print(f"host-pointer samp_p: {dt*1e3:.1f} ms  ({B/dt:.0f} preimages/s), output {e.nbytes/1e9:.2f} GB")
from tools_amd._ffi import lib as _lib
e2 = np.ones((B, psf.m), dtype=np.int64)           # caller-owned, already touched output buffer (what a Rust shim would reuse)
for i in range(3):
    t0 = time.perf_counter()
    check(_lib().psfp_samp_p(psf._h, C.c_uint64(20 + i), C.c_uint64(0), C.c_size_t(B), u.ctypes.data_as(C.POINTER(C.c_uint64)),
                             e2.ctypes.data_as(C.POINTER(C.c_int64))), "samp_p")
    dt = time.perf_counter() - t0
    print(f"host-pointer samp_p into a reused buffer: {dt*1e3:.1f} ms  ({B/dt:.0f} preimages/s)")
dev = torch.device("cuda:0")
ud = torch.from_numpy(u.astype(np.int64)).to(dev)
ed = torch.empty((B, psf.m), dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for i in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    psf.samp_p_dev(ud.data_ptr(), ed.data_ptr(), B, seed=10 + i, stream=st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"device-pointer samp_p_dev: {dt*1e3:.1f} ms")
print("same rows:", bool((ed.cpu().numpy() == psf.samp_p(u, seed=11, out=e2)).all()))
