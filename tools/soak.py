"""Soak of the full-size paths: many batches with fresh seeds; every batch is checked through A e = u and check_domain on the
device, and a slice of it bit for bit against the CPU oracle (rare events -- acceptance ties, screened-in-but-rejected
attempts, second sampling rounds, queue corner cases -- occur thousands of times over a run).
   python tools/soak.py c3|c2|c4|psfp:n:q:r:s:B [iterations] [oracle rows per iteration]"""
import ctypes as C, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools_amd as T
from tools_amd._ffi import lib, check
from oracle import oracle as O

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 32
O.build()
dev = torch.device("cuda:0")
if cfg == "c3" or cfg.startswith("psfp:"):
    n, q, r, s, B = 512, 2**30, 9.0, 512.0, 4096
    if cfg.startswith("psfp:"):                     # psfp:n:q:r:s:B
        f = cfg.split(":")
        n, q, r, s, B = int(f[1]), int(f[2]), float(f[3]), float(f[4]), int(f[5])
    psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s)
    check(lib().psfp_trap_gen(psf._h, C.c_uint64(3)), "trap_gen")
    A, (R, Lp, _) = psf.export_key()
    orc = O.PSFPerturbation(O.gadget_params_default(n, q), r, s); orc.load_key(A, R, Lp); del A, R, Lp
    m = psf.m
elif cfg == "c2":
    n, q, s, B = 256, 3329, 1024.0, 1024
    psf = T.PSFGPV(T.GadgetParameters.init_default(n, q), s)
    psf.trap_gen(3, export=False)
    A, (bt, gt) = psf.export_key()
    orc = O.PSFGPV(O.gadget_params_default(n, q), s); orc.load_key(A, bt, gt); del A, bt, gt
    m = psf.m
else:
    n, q, B = 256, 3329, 4096
    s = ((2 * 2 * 1.005 * math.sqrt(n) + 1) * 2) * 4
    psf = T.PSFGPVRing(T.GadgetParametersRing.init_default(n, q), s, 1.005)
    check(lib().psfring_trap_gen(psf._h, C.c_uint64(3)), "trap_gen")
    a, rr, ee, bt, gt = psf.export_key()
    orc = O.PSFGPVRing(O.gadget_params_ring_default(n, q), s, 1.005); orc.load_key(a, rr, ee, gso_t=gt)
    m = psf.d
u = torch.empty((B, n), dtype=torch.int64, device=dev)
e = torch.empty((B, m), dtype=torch.int64, device=dev)
u2 = torch.empty_like(u)
ok = torch.empty((B,), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
threads = O.num_threads()
t0 = time.time()
bad = 0
for it in range(iters):
    psf.uniform_targets_dev(u.data_ptr(), B, seed=500 + it, first_index=it * B, stream=st)
    psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=9000 + it, first_index=it * B, stream=st)
    status = psf.last_status()
    psf.f_a_dev(e.data_ptr(), u2.data_ptr(), ok.data_ptr(), B, stream=st)
    torch.cuda.synchronize()
    inv = bool((u2 == u).all().item()) and bool(ok.all().item()) and status == 0
    lo = (it * 37) % (B - rows) if B > rows else 0
    uh = u[lo:lo + rows].cpu().numpy().astype(np.uint64)
    e_cpu = orc.samp_p(9000 + it, uh, first_index=it * B + lo, nthreads=threads)
    same = bool((e_cpu.reshape(rows, -1) == e[lo:lo + rows].cpu().numpy()).all())
    if not (inv and same):
        bad += 1
        print(f"iteration {it}: invariants {inv} status {status} oracle rows equal {same}", flush=True)
print(f"SOAK {cfg}: {iters} batches of {B} ({iters * B} preimages), {iters * rows} rows compared with the oracle, failures {bad}, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
