#!/usr/bin/env python3
"""One samp_p call with 17 ... 1024 preimages at C3 (PSFPerturbation n=512 q=2^30; psf.rs:48-80 batched by the library): the streaming product (k_trmm_stream, the
library's shape per batch size) and, from 256 on, the batch kernel (PSF_TRMM_STREAM_MAX=0).  Median HIP-event time of `reps` synchronised calls, the product kernel's
own time, and a comparison of the rows of the two forms.  (Round 5 used the same harness for k_trmm_stream_lds -- the normals through LDS once per workgroup in
lock-step rounds of eight k-steps -- which was correct and SLOWER at every size, 1.31 vs 1.28 ms at 64 preimages, 2.42 vs 1.87 ms at 128: profiles/r05_notes.md.)"""
# the PSF_* switches this script sets are alive in the experiments build only (make -C tools_amd/csrc exp); the release library reads none of them
import os as _os
_os.environ.setdefault("PSF_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tools_amd", "lib", "libpsf_mi355x_exp.so"))
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tools_amd as T

batches = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "17,32,48,64,96,128,256,512,1024").split(",")]
reps = 12
n, q, r, s = 512, 2**30, 9.0, 512.0
psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s)
psf.trap_gen(3, export=False)
m = psf.m
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
Bmax = max(batches)
u = torch.empty((Bmax, n), dtype=torch.int64, device=dev)
psf.uniform_targets_dev(u.data_ptr(), Bmax, seed=7)
e = torch.empty((Bmax, m), dtype=torch.int64, device=dev)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out = {}


def arm(B, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        call = lambda: psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=11, stream=st)
        call(); call()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            ev0.record(); call(); ev1.record()
            torch.cuda.synchronize()
            ts.append(ev0.elapsed_time(ev1))
        ts.sort()
        psf.enable_timing(True)
        call()
        tm = dict(psf.get_timing())
        psf.enable_timing(False)
        rows = e[:B].clone()
        return round(ts[len(ts) // 2], 4), round(tm.get("k_trmm_f64", 0.0), 4), rows
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


for B in batches:
    reg = arm(B, {})
    rec = {"call_ms": reg[0], "product_ms": reg[1]}
    if B >= 256:
        big = arm(B, {"PSF_TRMM_STREAM_MAX": "0"})
        rec.update({"batch_kernel_call_ms": big[0], "batch_kernel_product_ms": big[1], "same_rows_batch_kernel": bool(torch.equal(reg[2], big[2]))})
    out[str(B)] = rec
    print(B, json.dumps(rec), flush=True)
print(json.dumps({"config": "c3", "reps": reps, "batches": out}))
