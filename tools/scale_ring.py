"""Scale check of PSFGPVRing (default C4: R_q = Z_3329[X]/(X^256+1), batch 4096)."""
import argparse, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tools_amd as T


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--q", type=int, default=3329)
    ap.add_argument("--s", type=float, default=0.0)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=2)
    a = ap.parse_args()
    s = a.s or ((2 * 2 * 1.005 * math.sqrt(a.n) + 1) * 2) * 4      # gpv_ring.rs:296-298
    dev = torch.device("cuda:0")
    gp = T.GadgetParametersRing.init_default(a.n, a.q)
    print(gp, "s =", s, flush=True)
    psf = T.PSFGPVRing(gp, s, 1.005)
    t0 = time.time()
    psf.trap_gen(4)
    torch.cuda.synchronize()
    print(f"trap_gen {time.time()-t0:.2f}s  d={psf.d}", flush=True)
    B = a.batch
    u = torch.empty((B, psf.n), dtype=torch.int64, device=dev)
    e = torch.empty((B, psf.d), dtype=torch.int64, device=dev)
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
    print(f"a*sigma == u: {bool((u2 == u).all())}  check_domain all: {bool(ok.all())}  |sigma| mean {nrm.mean().item():.1f} bound {s*psf.d**0.5:.1f}")


if __name__ == "__main__":
    main()
