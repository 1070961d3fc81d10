#!/usr/bin/env python3
"""trap_gen wall time per configuration (key generation only, no sampling): python3 tools/keygen_time.py c3 [c2 c4 ...]"""
import ctypes as C
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tools_amd as T
from tools_amd._ffi import lib, check

CFG = {"c3": ("p", 512, 2**30, 9.0, 512.0), "c3prime": ("p", 512, 1073741789, 9.0, 512.0), "c2": ("g", 256, 3329, None, 1024.0),
       "c4": ("r", 256, 3329, None, 0.0), "c5": ("p", 1024, 2**60, 10.0, 1024.0), "bench64": ("p", 64, 128, 6.0, 100.0)}
for name in sys.argv[1:] or ["c3"]:
    kind, n, q, r, s = CFG[name]
    for rep in range(2):
        t0 = time.perf_counter()
        if kind == "p":
            psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s)
            t1 = time.perf_counter()
            check(lib().psfp_trap_gen(psf._h, C.c_uint64(3)), "trap_gen")
        elif kind == "g":
            psf = T.PSFGPV(T.GadgetParameters.init_default(n, q), s)
            t1 = time.perf_counter()
            psf.trap_gen(3, export=False)
        else:
            s = ((2 * 2 * 1.005 * math.sqrt(n) + 1) * 2) * 4
            psf = T.PSFGPVRing(T.GadgetParametersRing.init_default(n, q), s, 1.005)
            t1 = time.perf_counter()
            check(lib().psfring_trap_gen(psf._h, C.c_uint64(3)), "trap_gen")
        t2 = time.perf_counter()
        print(f"{name} rep {rep}: create {t1 - t0:.3f} s, trap_gen {t2 - t1:.3f} s", flush=True)
        psf.close()
