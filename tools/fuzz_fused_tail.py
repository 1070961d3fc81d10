"""Seeded random configurations through the fused rounding + syndrome launch of a single call (k_round_syndrome_small: one or two preimages, q <= 2^32, n a multiple of 8): several calls per
key -- the first builds the compact copies and runs the separate kernels, the later ones the fused launch -- every call against the CPU oracle, bit for bit.
   python3 tools/fuzz_fused_tail.py <first case> <count>"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tools_amd as T
from oracle import oracle as O
O.build()
PRIMES = [257, 3329, 7681, 12289, 65537, 1073741789, 2**31 - 1, 4294967291]
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = fused_calls = calls = 0
t0 = time.time()
for case in range(first, first + count):
    rng = np.random.default_rng(50000 + case)
    while True:
        n = int(rng.choice([8, 8, 16, 24, 40, 64, 96]))
        kind = rng.integers(0, 3)
        q = int(2 ** rng.integers(6, 33)) if kind == 0 else (int(PRIMES[rng.integers(0, len(PRIMES))]) if kind == 1 else (int(rng.integers(65, 2**32)) | 1))
        base = int(rng.choice([2, 2, 2, 3, 5]))
        k = 1
        while base**k < q:
            k += 1
        m_bar = n * int(math.ceil(math.log2(q))) + int(rng.integers(0, 40))
        r = float(rng.choice([1.5, 3.0, 4.5, 30.0]))
        bound = r * math.sqrt(base * base + 1) * (math.sqrt(m_bar) + math.sqrt(n * k) + 4.0)
        s = bound * float(rng.choice([1.1, 1.5, 3.0]))
        if s * r * math.sqrt(m_bar + n * k) < 2**23 * 0.9 and m_bar + n * k > 256 and k <= 64:
            break
    B = int(rng.choice([1, 1, 2]))
    try:
        psf = T.PSFPerturbation(T.GadgetParameters(n, k, m_bar, base, q), r, s)
        A, (R, Lp, _) = psf.trap_gen(100 + case)
        orc = O.PSFPerturbation(O.GadgetParams(n, k, m_bar, base, q), r, s)
        orc.load_key(A, R, Lp)
        for it in range(5):
            u = O.uniform_targets(case * 7 + it, B, n, q)
            fi = int(rng.integers(0, 2**40))
            psf.enable_timing(True)
            e = psf.samp_p(u, seed=9 + it, first_index=fi)
            names = set(dict(psf.get_timing()))
            psf.enable_timing(False)
            calls += 1
            fused = "k_perturb_round" not in names and "k_trmm_f64" in names
            fused_calls += fused
            if not (e == orc.samp_p(9 + it, u, first_index=fi)).all():
                bad += 1
                print(f"FAIL case {case} call {it} fused={fused}: n={n} q={q} base={base} k={k} m_bar={m_bar} r={r} s={s} B={B}", flush=True)
            time.sleep(0.01)
        psf.close()
    except BaseException as ex:      # noqa
        bad += 1
        print(f"FAIL case {case}: {type(ex).__name__}: {str(ex)[:300]} (n={n} q={q} base={base} k={k} m_bar={m_bar} r={r} s={s} B={B})", flush=True)
print(f"done: {count} cases from {first}, {calls} calls ({fused_calls} through the fused launch), {bad} failures, {time.time() - t0:.0f} s", flush=True)
