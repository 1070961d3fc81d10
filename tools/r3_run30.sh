#!/bin/bash
O=gpurun_out/r3_run30; mkdir -p $O
timeout 1200 python3 - <<'PY' 2>&1 | tee $O/slow.log
import time, math, sys
t0=time.perf_counter()
import torch, numpy as np
sys.path.insert(0,'.')
import tools_amd as T
def lap(s):
    global t0
    torch.cuda.synchronize(); t=time.perf_counter(); print(f"{s}: {t-t0:.2f} s", flush=True); t0=t
lap("import")
psf = T.PSFPerturbation(T.GadgetParameters.init_default(512, 2**30), 9.0, 512.0)
psf.trap_gen(5, export=False)
lap("trap_gen")
B, m = 4096, psf.m
dev = torch.device("cuda:0")
u = torch.empty((B, psf.n), dtype=torch.int64, device=dev)
e = torch.empty((B, m), dtype=torch.int64, device=dev)
lap("alloc")
psf.uniform_targets_dev(u.data_ptr(), 1, seed=9)
u[1:] = u[0]
lap("targets")
psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=2024)
lap("samp_p")
var = torch.zeros(m, dtype=torch.float64, device=dev)
mean = torch.zeros(m, dtype=torch.float64, device=dev)
for c0 in range(0, m, 4096):
    x = e[:, c0:c0 + 4096].to(torch.float64)
    lap(f"to {c0}")
    mean[c0:c0 + 4096] = x.mean(dim=0)
    lap(f"mean {c0}")
    var[c0:c0 + 4096] = x.var(dim=0, unbiased=True)
    lap(f"var {c0}")
idx = [0, 127, 128, 4095, 15440, 15441, 20000, m - 129, m - 128, m - 1]
sub = e[:, idx].to(torch.float64).cpu().numpy()
lap("sub")
PY
