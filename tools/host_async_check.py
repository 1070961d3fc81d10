"""Debug/verification: asynchronous host-pointer calls at the C3 shape against synchronous ones, row by row."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import tools_amd as T
from tools_amd._ffi import lib, check
n, q, r, s, B = 512, 2**30, 9.0, 512.0, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s)
check(lib().psfp_trap_gen(psf._h, C.c_uint64(3)), "trap_gen")
u = np.random.default_rng(1).integers(0, q, size=(B, n), dtype=np.uint64)
ref = [psf.samp_p(u, seed=50 + i).copy() for i in range(3)]
outs = [np.full((B, psf.m), -7, dtype=np.int64) for _ in range(3)]
for i in range(3):
    psf.samp_p_async(u, outs[i], seed=50 + i)
psf.wait()
for i in range(3):
    bad = np.nonzero((outs[i] != ref[i]).any(axis=1))[0]
    print(f"call {i}: {len(bad)} rows differ", (bad[:5], bad[-5:]) if len(bad) else "", "untouched entries:", int((outs[i] == -7).sum()))
    if len(bad):
        b = bad[0]; cols = np.nonzero(outs[i][b] != ref[i][b])[0]
        print("   first bad row", b, "cols", cols[:8], "got", outs[i][b][cols[:4]], "want", ref[i][b][cols[:4]])
print("valid (A e = u):", bool((psf.f_a(outs[2][:64]) == u[:64]).all()))
