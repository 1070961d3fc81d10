import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import tools_amd as T
from oracle import oracle as O
for (n, q, r, s) in [(64, 128, 6.0, 100.0), (512, 2**30, 9.0, 512.0)]:
    psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s)
    psf.trap_gen(3, export=False)
    u = O.uniform_targets(5, 1, n, q)
    for it in range(5):
        psf.enable_timing(True)
        st = psf.samp_p_stages(u, seed=44, first_index=9)
        tm = dict(psf.get_timing())
        psf.enable_timing(False)
        print(n, it, {k: round(v, 4) for k, v in tm.items()}, bool((psf.f_a(st["e"]) == u).all()))
        time.sleep(0.05)
