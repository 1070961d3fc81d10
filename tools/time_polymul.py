#!/usr/bin/env python3
"""R_q products (PolynomialRingZq, gadget_ring.rs:78 / gpv_ring.rs:245-246) at q = 3329, n = 256 (BASELINE.json configs[3]) for `count` products per launch:
the wave-level NTT kernel through psf_poly_mul_negacyclic_dev on device buffers in both layouts (16-bit and 64-bit I/O), the product against a cached
image (psf_poly_mul_hat_dev), and the exact schoolbook kernel through the host entry point for reference.  Times are HIP events around `reps` back-to-back
launches; run it under `rocprofv3 --kernel-trace --stats` for the per-kernel summary kept as profiles/r05_kernel_stats_polymul.csv."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tools_amd import gadget as G

q = int(os.environ.get("POLY_Q", 3329))
n = int(os.environ.get("POLY_N", 256))
count = int(sys.argv[1]) if len(sys.argv) > 1 else 4096 * 13       # C4: 4096 preimages x (k + 2) ring elements
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
a = rng.integers(0, q, size=(count, n), dtype=np.uint64)
b = rng.integers(-q // 2, q // 2, size=(count, n), dtype=np.int64)
da64, db64 = torch.from_numpy(a.view(np.int64)).to(dev), torch.from_numpy(b).to(dev)
do64 = torch.empty((count, n), dtype=torch.int64, device=dev)
out = {"q": q, "n": n, "count": count, "reps": reps}


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps                         # us per launch


st = torch.cuda.current_stream().cuda_stream
us = timed(lambda: G.poly_mul_negacyclic_dev(da64.data_ptr(), db64.data_ptr(), do64.data_ptr(), q, n, count, 64, stream=st))
ref = do64.cpu().numpy().view(np.uint64)
out["ntt_io64"] = {"us_per_launch": round(us, 2), "products_per_s": round(count / us * 1e6), "GBps": round(count * n * 24 / us / 1e3, 1)}
if q < 2**14:
    da16 = torch.from_numpy(a.astype(np.uint16).view(np.int16)).to(dev)
    db16 = torch.from_numpy(b.astype(np.int16)).to(dev)
    do16 = torch.empty((count, n), dtype=torch.int16, device=dev)
    us = timed(lambda: G.poly_mul_negacyclic_dev(da16.data_ptr(), db16.data_ptr(), do16.data_ptr(), q, n, count, 16, stream=st))
    out["ntt_io16"] = {"us_per_launch": round(us, 2), "products_per_s": round(count / us * 1e6), "GBps": round(count * n * 6 / us / 1e3, 1)}
    out["io16_same_residues"] = bool((do16.cpu().numpy().view(np.uint16).astype(np.uint64) == ref).all())
    hat = torch.empty((n,), dtype=torch.int32, device=dev)
    G.ntt_forward_dev(da16.data_ptr(), hat.data_ptr(), q, n, 1, 16, stream=st)
    us = timed(lambda: G.poly_mul_hat_dev(hat.data_ptr(), 0, db16.data_ptr(), do16.data_ptr(), q, n, count, 16, stream=st))
    out["ntt_io16_cached_key"] = {"us_per_launch": round(us, 2), "products_per_s": round(count / us * 1e6), "GBps": round(count * n * 4 / us / 1e3, 1)}
if count <= 4096 * 13:
    t0 = time.perf_counter()
    sb = G.poly_mul_negacyclic(a, b, q, method=0)
    out["schoolbook_host_entry_ms_incl_copies"] = round((time.perf_counter() - t0) * 1e3, 2)
    out["same_residues"] = bool((sb == ref).all())
print(json.dumps(out))
