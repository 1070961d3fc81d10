"""Scale check of PSFGPV (default C2: n=256 q=3329 s=1024 batch=1024)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tools_amd as T


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--q", type=int, default=3329)
    ap.add_argument("--s", type=float, default=1024.0)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    gp = T.GadgetParameters.init_default(a.n, a.q)
    print(gp, flush=True)
    psf = T.PSFGPV(gp, a.s)
    t0 = time.time()
    psf.trap_gen(3, export=False)
    torch.cuda.synchronize()
    print(f"trap_gen {time.time()-t0:.2f}s  m={psf.m}", flush=True)
    B = a.batch
    u = torch.empty((B, psf.n), dtype=torch.int64, device=dev)
    e = torch.empty((B, psf.m), dtype=torch.int64, device=dev)
    u2 = torch.empty_like(u)
    ok = torch.empty((B,), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    psf.uniform_targets_dev(u.data_ptr(), B, seed=3, stream=st)
    psf.enable_timing(True)
    for rep in range(a.reps):
        torch.cuda.synchronize(); t0 = time.time()
        psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=100 + rep, stream=st)
        rc = psf.last_status()
        dt = time.time() - t0
        print(f"rep {rep}: status {rc} {dt*1e3:.2f} ms -> {B/dt:.0f} preimages/s  {psf.get_timing()}", flush=True)
    psf.f_a_dev(e.data_ptr(), u2.data_ptr(), ok.data_ptr(), B, stream=st)
    torch.cuda.synchronize()
    nrm = (e.double() ** 2).sum(1).sqrt()
    print(f"A e == u: {bool((u2 == u).all())}  check_domain all: {bool(ok.all())}  |e| mean {nrm.mean().item():.1f} bound {a.s*psf.m**0.5:.1f}")


if __name__ == "__main__":
    main()
