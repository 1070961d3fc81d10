"""Stress of the asynchronous host-pointer path at full size (C3, batch 4096): N overlapped psfp_samp_p_async calls with fresh seeds, two output buffers in turn; every call's
rows are checked against the device-pointer call of the same seed through a 64-bit checksum of all entries and an exact comparison of 64 sampled rows (the DMA transport,
the per-slot rings and the worker threads are exercised ~N x 64 chunk transfers deep).   python3 tools/host_async_stress.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
import tools_amd as T
from tools_amd._ffi import lib, check

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n, q, r, s, B = 512, 2**30, 9.0, 512.0, 4096
psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s)
check(lib().psfp_trap_gen(psf._h, C.c_uint64(3)), "trap_gen")
rng = np.random.default_rng(5)
u = rng.integers(0, q, size=(B, n), dtype=np.uint64)
dev = torch.device("cuda:0")
ud = torch.from_numpy(u.astype(np.int64)).to(dev)
ed = torch.empty((B, psf.m), dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
outs = [np.zeros((B, psf.m), dtype=np.int64) for _ in range(2)]
rows = np.sort(rng.choice(B, size=64, replace=False))
bad = 0
t0 = time.time()
pending = []          # (call index, buffer) whose rows are complete only after the NEXT call has been issued and a wait has returned
def verify(i, buf):
    global bad
    psf.samp_p_dev(ud.data_ptr(), ed.data_ptr(), B, seed=1000 + i, first_index=7 * i, stream=st)
    torch.cuda.synchronize()
    want_sum = int(ed.sum().item())
    got_sum = int(buf.sum(dtype=np.int64))
    ok = want_sum == got_sum and bool((ed[torch.from_numpy(rows).to(dev)].cpu().numpy() == buf[rows]).all())
    if not ok:
        bad += 1
        print(f"call {i}: MISMATCH (checksum {got_sum} vs {want_sum})", flush=True)
for i in range(N):
    psf.samp_p_async(u, outs[i & 1], seed=1000 + i, first_index=7 * i)
    if i >= 1 and (i % 2 == 1):
        psf.wait()                      # both buffers complete: check the two calls, then go on overlapping
        verify(i - 1, outs[(i - 1) & 1]); verify(i, outs[i & 1])
psf.wait()
if N % 2 == 1:
    verify(N - 1, outs[(N - 1) & 1])
print(f"STRESS host async: {N} calls of {B} preimages, {bad} mismatches, {time.time() - t0:.1f} s")
