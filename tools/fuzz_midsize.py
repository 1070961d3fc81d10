"""Seeded random PSFPerturbation configurations at the batch sizes the mid-size kernels of round 6 serve -- 5 ... 200 preimages with 2 049 ... 98 304 gadget problems:
k_trmm_stream_wg (33 ... 64 preimages), k_gadget_quad (<8> and <16>), k_recombine_wg (5 ... 64 preimages, K a multiple of 128, one or two digit planes) -- against the
oracle: sampled rows bit for bit, A e = u and check_domain on every row.  The suite's own random configurations keep n <= 12, where none of the three is chosen.
   python3 tools/fuzz_midsize.py <first case> <count>"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from oracle import oracle  # noqa: E402
import tools_amd as T  # noqa: E402

oracle.build()
PRIMES = [257, 3329, 7681, 12289, 65537, 1073741789, 2**31 - 1]


def one(case):
    rng = np.random.default_rng(77000 + case)
    while True:
        base = int(rng.choice([2, 2, 2, 3, 5, 32]))
        kind = int(rng.integers(0, 3))
        long_chain = base == 2 and rng.integers(0, 4) == 0          # chains of 33 ... 50 draws: k_gadget_quad<16>
        q = int(2 ** rng.integers(33, 51)) if long_chain else int(2 ** rng.integers(6, 50)) if kind == 0 else int(PRIMES[rng.integers(0, len(PRIMES))]) if kind == 1 else int(rng.integers(64, 2**24)) | 1
        k = 1
        while base**k < q:
            k += 1
        if k > 64:
            continue
        B = int(rng.choice([100, 130, 200])) if long_chain else int(rng.choice([5, 9, 16, 17, 31, 33, 40, 48, 63, 64, 65, 100, 130, 200]))
        n_lo, n_hi = 2049 // B + 1, min(98304 // B, 160)
        if n_lo > n_hi:
            continue
        n = int(rng.integers(n_lo, n_hi + 1))
        if n * k > 2600:                      # keep the key (and the oracle's time) small
            continue
        if rng.integers(0, 2):                # every other case: K = n k rounded up to 64 is a multiple of 128 (k_recombine_wg)
            while ((n * k + 63) // 64) % 2 and n < n_hi:
                n += 1
        m_bar = n * int(math.ceil(math.log2(q))) + int(rng.integers(0, 40))
        r = float(rng.choice([2.0, 3.0, 4.5, 30.0])) if base != 32 else 6.0
        bound = r * math.sqrt(base * base + 1) * (math.sqrt(m_bar) + math.sqrt(n * k) + 4.0)
        s = bound * float(rng.choice([1.1, 1.5, 3.0]))
        if s * r * math.sqrt(m_bar + n * k) < 2**23 * 0.9 and m_bar + n * k < 6000:
            break
    gp = T.GadgetParameters(n, k, m_bar, base, q)
    psf = T.PSFPerturbation(gp, r, s)
    A, (R, Lp, _) = psf.trap_gen(100 + case)
    orc = oracle.PSFPerturbation(oracle.GadgetParams(n, k, m_bar, base, q), r, s)
    orc.load_key(A, R, Lp)
    u = oracle.uniform_targets(case, B, n, q)
    first = int(rng.integers(0, 2**40))
    st = psf.samp_p_stages(u, seed=7 + case, first_index=first)
    e = st["e"]
    assert psf.last_status() == 0
    assert (psf.samp_p(u, seed=7 + case, first_index=first) == e).all()
    for b in sorted({0, 1, 15, 16, B // 2, B - 2, B - 1}):
        if b < B:
            assert (e[b] == orc.samp_p(7 + case, u[b:b + 1], first_index=first + b)[0]).all(), ("row", b, n, q, base, k, m_bar, r, s, B)
    assert (psf.f_a(e) == u).all() and psf.check_domain(e).all(), (n, q, base, k, m_bar, r, s, B)
    hi = bool(np.abs(st["z"]).max() > 127)
    psf.close()
    return (n, k, base, B, hi)


def main():
    first, count = int(sys.argv[1]), int(sys.argv[2])
    bad = 0
    t0 = time.time()
    seen = {"wg_product": 0, "quad16": 0, "recombine_wg": 0, "second_plane": 0}
    for case in range(first, first + count):
        try:
            n, k, base, B, hi = one(case)
            seen["wg_product"] += 33 <= B <= 64
            seen["quad16"] += k > 32
            seen["recombine_wg"] += B <= 64 and ((n * k + 63) // 64) % 2 == 0
            seen["second_plane"] += hi
        except BaseException as ex:      # noqa
            bad += 1
            print(f"FAIL case {case}: {type(ex).__name__}: {str(ex)[:300]}", flush=True)
    print(f"done: {count} cases from {first}, {bad} failures, {time.time() - t0:.0f} s, kernels hit {seen}", flush=True)


if __name__ == "__main__":
    main()
