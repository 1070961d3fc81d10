"""Cycle breakdown of k_gpv_nearest_plane (workgroup 0) from a -DNP_PROFILE build of the library:
   PSF_LIB=tools/bin/libpsf_prof.so python tools/np_profile.py [c2|c4]"""
import ctypes as C, math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools_amd as T
from tools_amd._ffi import lib, check

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
dev = torch.device("cuda", 0)
if cfg == "c2":
    n, q, B = 256, 3329, 1024
    psf = T.PSFGPV(T.GadgetParameters.init_default(n, q), 1024.0)
    psf.trap_gen(3, export=False)
    m = psf.m
else:
    n, q, B = 256, 3329, 4096
    s = ((2 * 2 * 1.005 * math.sqrt(n) + 1) * 2) * 4
    psf = T.PSFGPVRing(T.GadgetParametersRing.init_default(n, q), s, 1.005)
    check(lib().psfring_trap_gen(psf._h, C.c_uint64(3)), "trap_gen")
    m = psf.d
u = torch.empty((B, n), dtype=torch.int64, device=dev)
e = torch.empty((B, m), dtype=torch.int64, device=dev)
psf.uniform_targets_dev(u.data_ptr(), B, seed=7)
psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=1)
torch.cuda.synchronize()
out = (C.c_longlong * 8)()
lib().psf_debug_np_prof(out, 1)
psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=2)
torch.cuda.synchronize()
lib().psf_debug_np_prof(out, 0)
names = ["projection fma", "butterfly", "tables+prefetch issue+barrier", "sampling (wave p)", "barrier 2", "update"]
tot = sum(out[:6])
for nm, v in zip(names, out):
    print(f"{nm:32s} {v:12d} ticks  {100.0 * v / max(tot, 1):5.1f}%  per step {v / m:8.1f}")
print("total ticks", tot, "steps", m)

import struct
print("largest FP53 exactness bound reached by a workgroup: 2^%.1f" % math.log2(max(struct.unpack("d", struct.pack("q", out[6]))[0], 1.0)))
