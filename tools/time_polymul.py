#!/usr/bin/env python3
"""R_q products (PolynomialRingZq, gadget_ring.rs:78 / gpv_ring.rs:245-246) through psf_poly_mul_negacyclic: the NTT kernel against the schoolbook kernel at
q = 3329, n = 256 (BASELINE.json configs[3]) for `count` products per call.  The entry point takes host buffers, so the wall time includes both copies; run it
under `rocprofv3 --kernel-trace --stats` for the kernels alone (tools/r4_profiles.sh keeps that summary as profiles/r04_kernel_stats_polymul.csv)."""
import json
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools_amd import gadget as G

q, n = 3329, 256
count = int(sys.argv[1]) if len(sys.argv) > 1 else 4096 * 13       # C4: 4096 preimages x (k + 2) ring elements
rng = np.random.default_rng(1)
a = rng.integers(0, q, size=(count, n), dtype=np.uint64)
b = rng.integers(-q // 2, q // 2, size=(count, n), dtype=np.int64)
out = {}
ref = None
for name, method in (("ntt", 1), ("schoolbook", 0)):
    r = G.poly_mul_negacyclic(a, b, q, method=method)             # warm-up
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        r = G.poly_mul_negacyclic(a, b, q, method=method)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    out[name] = {"wall_ms_median_incl_copies": round(ts[2] * 1e3, 3), "products_per_s_incl_copies": round(count / ts[2], 1)}
    if ref is None:
        ref = r
    else:
        out["same_residues"] = bool((ref == r).all())
out.update({"q": q, "n": n, "count": count})
print(json.dumps(out))
