"""Host-pointer samp_p against device-pointer samp_p_dev of the same seed on seeded random PSFPerturbation configurations (the menus of tests/test_gpu_random_configs.py,
--wide as in tools/fuzz_configs.py): no oracle in the loop, so thousands of calls per minute; on a mismatch prints which rows / coordinates differ.
    python3 tools/host_vs_device_fuzz.py <first case> <count> [--wide|--narrow] [calls per key] [--side-stream]"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import tools_amd as T
import test_gpu_random_configs as R

first, count = int(sys.argv[1]), int(sys.argv[2])
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 4
if "--wide" in sys.argv:
    R.R_MENU = [1.5, 3.0, 30.0, 100.0, 400.0]; R.S_FACTOR_MENU = [1.02, 1.5, 10.0]; R.BATCH_MENU = [1, 129, 300, 512, 777, 1024]
dev = torch.device("cuda:0")
bad = 0; ncalls = 0; t0 = time.time()
for case in range(first, first + count):
    rng = np.random.default_rng(1000 + case)
    while True:
        n = int(rng.integers(2, 13)); q = R.draw_modulus(rng); base = int(rng.choice([2, 2, 2, 3, 5, 7])); k = 1
        while base**k < q: k += 1
        if k > 64: base, k = 2, int(math.ceil(math.log2(q)))
        m_bar = n * int(math.ceil(math.log2(q))) + int(rng.integers(0, 40)); r = float(rng.choice(R.R_MENU))
        bound = r * math.sqrt(base * base + 1) * (math.sqrt(m_bar) + math.sqrt(n * k) + 4.0); s = bound * float(rng.choice(R.S_FACTOR_MENU)); B = R.draw_batch(rng)
        if s * r * math.sqrt(m_bar + n * k) < 2**23 * 0.9: break
    psf = T.PSFPerturbation(T.GadgetParameters(n, k, m_bar, base, q), r, s)
    psf.trap_gen(100 + case, export=False)
    m = m_bar + n * k
    u = np.random.default_rng(case).integers(0, q, size=(B, n), dtype=np.uint64)
    ud = torch.from_numpy(u.astype(np.int64)).to(dev); ed = torch.empty((B, m), dtype=torch.int64, device=dev)
    side = torch.cuda.Stream() if "--side-stream" in sys.argv else None      # a non-blocking stream for the device-pointer calls, issued FIRST (the first call after key generation)
    st = side.cuda_stream if side is not None else torch.cuda.current_stream().cuda_stream
    for c in range(calls):
        try:
            if side is not None:
                psf.samp_p_dev(ud.data_ptr(), ed.data_ptr(), B, seed=7 + case + 1000 * c, first_index=c, stream=st); torch.cuda.synchronize()
                e = psf.samp_p(u, seed=7 + case + 1000 * c, first_index=c)
            else:
                e = psf.samp_p(u, seed=7 + case + 1000 * c, first_index=c)
                psf.samp_p_dev(ud.data_ptr(), ed.data_ptr(), B, seed=7 + case + 1000 * c, first_index=c, stream=st); torch.cuda.synchronize()
        except T.PsfError as ex:
            print(f"case {case} call {c}: PsfError {ex.status} (n={n} q={q} base={base} k={k} m_bar={m_bar} r={r} s={s:.1f} B={B})", flush=True); break
        ncalls += 1
        d = np.asarray(e) != ed.cpu().numpy()
        if d.any():
            bad += 1
            rows = np.nonzero(d.any(axis=1))[0]; cols = np.nonzero(d.any(axis=0))[0]
            print(f"MISMATCH case {case} call {c}: n={n} q={q} base={base} k={k} m_bar={m_bar} r={r} s={s:.1f} B={B}: {int(d.sum())} entries, rows {rows[:8].tolist()}.. ({len(rows)}), cols {cols[:8].tolist()}.. ({len(cols)})", flush=True)
    psf.close()
print(f"done: {count} configurations from {first}, {ncalls} call pairs, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
