import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import tools_amd as T
from tools_amd._ffi import lib, check
n, q, r, s, B = 512, 2**30, 9.0, 512.0, 4096
psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s)
check(lib().psfp_trap_gen(psf._h, C.c_uint64(3)), "trap_gen")
u = np.random.default_rng(1).integers(0, q, size=(B, n), dtype=np.uint64)
outs = [np.ones((B, psf.m), dtype=np.int64) for _ in range(2)]
psf.samp_p_async(u, outs[0], seed=1); psf.wait()
psf.enable_timing(True)
for i in range(4):
    psf.samp_p_async(u, outs[i & 1], seed=2 + i)
psf.wait()
print("kernels of the last overlapped call:", {k: round(v, 2) for k, v in psf.get_timing()})
psf.samp_p_async(u, outs[0], seed=9); psf.wait()
print("kernels of a lone call:", {k: round(v, 2) for k, v in psf.get_timing()})
