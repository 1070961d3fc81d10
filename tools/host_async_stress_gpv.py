"""Stress of the asynchronous host-pointer path of the nearest-plane types at full size (C2: PSFGPV n=256 q=3329 batch 1024; C4: PSFGPVRing batch 4096): N overlapped
samp_p_async calls with fresh seeds, two output buffers in turn; every call's rows are compared with the device-pointer call of the same seed (checksum of all entries and
64 sampled rows exactly).   python3 tools/host_async_stress_gpv.py [N]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools_amd as T

N = int(sys.argv[1]) if len(sys.argv) > 1 else 150


def stress(name, psf, B, n, q, shape):
    rng = np.random.default_rng(5)
    u = rng.integers(0, q, size=(B, n), dtype=np.uint64)
    dev = torch.device("cuda:0")
    ud = torch.from_numpy(u.astype(np.int64)).to(dev)
    cols = int(np.prod(shape[1:]))
    ed = torch.empty((B, cols), dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    outs = [np.zeros(shape, dtype=np.int64) for _ in range(2)]
    rows = np.sort(rng.choice(B, size=64, replace=False))
    bad = 0
    t0 = time.time()

    def verify(i, buf):
        nonlocal bad
        psf.samp_p_dev(ud.data_ptr(), ed.data_ptr(), B, seed=1000 + i, first_index=7 * i, stream=st)
        torch.cuda.synchronize()
        flat = buf.reshape(B, cols)
        ok = int(ed.sum().item()) == int(flat.sum(dtype=np.int64)) and bool((ed[torch.from_numpy(rows).to(dev)].cpu().numpy() == flat[rows]).all())
        if not ok:
            bad += 1
            print(f"{name} call {i}: MISMATCH", flush=True)

    for i in range(N):
        psf.samp_p_async(u, outs[i & 1], seed=1000 + i, first_index=7 * i)
        if i >= 1 and (i % 2 == 1):
            psf.wait()
            verify(i - 1, outs[(i - 1) & 1]); verify(i, outs[i & 1])
    psf.wait()
    if N % 2 == 1:
        verify(N - 1, outs[(N - 1) & 1])
    print(f"STRESS host async {name}: {N} calls of {B} preimages, {bad} mismatches, {time.time() - t0:.1f} s", flush=True)


gpv = T.PSFGPV(T.GadgetParameters.init_default(256, 3329), 1024.0); gpv.trap_gen(3, export=False)
stress("C2 PSFGPV", gpv, 1024, 256, 3329, (1024, gpv.m))
gpv.close()
n = 256; s = ((2 * 2 * 1.005 * math.sqrt(n) + 1) * 2) * 4
ring = T.PSFGPVRing(T.GadgetParametersRing.init_default(n, 3329), s, 1.005); ring.trap_gen(4)
stress("C4 PSFGPVRing", ring, 4096, n, 3329, (4096, ring.K, ring.n))
