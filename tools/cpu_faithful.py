#!/usr/bin/env python3
"""CPU legs BASELINE.md section 3 promises for the reference's own bench parameter sets (benches/psf.rs:27,52,79): one samp_p call, single
thread, (a) the flat-array port (oracle/psf_oracle.c) and (b) the GMP exact-rational "faithful mode" (oracle/psf_faithful_gmp.c: dense rational
mat-vec, dense nk x nk Gram-Schmidt matrix, [R; I] rebuilt per call -- the arithmetic style of the FLINT-backed reference).  Pure CPU: uses only
oracle/.  Prints one JSON object; commit it under profiles/."""
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return ""


def time_calls(fn, min_s=2.0, max_calls=200):
    fn(0)
    t0 = time.perf_counter()
    k = 0
    while True:
        fn(k + 1)
        k += 1
        dt = time.perf_counter() - t0
        if dt >= min_s or k >= max_calls:
            return dt / k, k


def main():
    O.build()
    out = {"cpu": cpu_model(), "threads": 1, "what": "seconds per PSF::samp_p call (one preimage), trapdoor outside the timed region (benches/psf.rs:36,61,88)", "sets": []}
    have_gmp = O.faithful_lib() is not None
    # benches/psf.rs:51-66 and :78-93
    for name, n, q, r, s in (("PSF Perturbation n=8", 8, 128, math.log2(8), 30.0), ("PSF Perturbation n=64", 64, 128, math.log2(64), 100.0)):
        gp = O.gadget_params_default(n, q)
        orc = O.PSFPerturbation(gp, r, s)
        assert orc.trap_gen(1) == 0
        u = O.uniform_targets(2, 1, n, q)[0]
        port, k1 = time_calls(lambda i: orc.samp_p_trace(100 + i, 0, u))
        rec = {"bench": name, "n": n, "q": q, "m": orc.m, "port_s_per_call": port, "port_calls": k1}
        if have_gmp:
            e = O.faithful_psfp_samp_p(orc, 5, 0, u)
            assert (orc.f_a(e.reshape(1, -1))[0] == u).all() and orc.check_domain(e)[0]
            fa, k2 = time_calls(lambda i: O.faithful_psfp_samp_p(orc, 100 + i, 0, u), max_calls=20)
            rec.update({"faithful_s_per_call": fa, "faithful_calls": k2, "faithful_over_port": fa / port})
        out["sets"].append(rec)
    # benches/psf.rs:26-39
    n, q = 8, 128
    s = 30.0 * math.log2(n)
    gp = O.gadget_params_default(n, q)
    orc = O.PSFGPV(gp, s)
    assert orc.trap_gen(1) == 0
    u = O.uniform_targets(2, 1, n, q)
    port, k1 = time_calls(lambda i: orc.samp_p(100 + i, u, percall=True, nthreads=1))
    rec = {"bench": "PSF GPV n=8", "n": n, "q": q, "m": orc.m, "port_s_per_call": port, "port_calls": k1}
    if have_gmp:
        A, bt, gt = orc.A, orc.basis_t, orc.gso_t
        e = O.faithful_gpv_samp_p(A, q, bt, gt, s, 5, 0, u[0])
        assert ((A.astype(object) @ e.astype(object)) % q == u[0].astype(object)).all()
        fa, k2 = time_calls(lambda i: O.faithful_gpv_samp_p(A, q, bt, gt, s, 100 + i, 0, u[0]), max_calls=50)
        rec.update({"faithful_s_per_call": fa, "faithful_calls": k2, "faithful_over_port": fa / port})
    out["sets"].append(rec)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
