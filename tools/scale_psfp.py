"""Scale check of the PSFPerturbation path at a named configuration (default C3); prints per-kernel times."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tools_amd as T


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--q", type=int, default=2**30)
    ap.add_argument("--r", type=float, default=9.0)
    ap.add_argument("--s", type=float, default=512.0)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()

    dev = torch.device("cuda:0")
    gp = T.GadgetParameters.init_default(a.n, a.q)
    print(gp, flush=True)
    t0 = time.time()
    psf = T.PSFPerturbation(gp, a.r, a.s)
    print(f"create {time.time()-t0:.2f}s  m={psf.m}", flush=True)
    t0 = time.time()
    from tools_amd._ffi import lib, check
    import ctypes as C
    check(lib().psfp_trap_gen(psf._h, C.c_uint64(3)), "trap_gen")
    torch.cuda.synchronize()
    print(f"trap_gen {time.time()-t0:.2f}s", flush=True)
    B = a.batch
    u = torch.empty((B, psf.n), dtype=torch.int64, device=dev)
    e = torch.empty((B, psf.m), dtype=torch.int64, device=dev)
    u2 = torch.empty((B, psf.n), dtype=torch.int64, device=dev)
    ok = torch.empty((B,), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    psf.uniform_targets_dev(u.data_ptr(), B, seed=3, stream=st)
    psf.enable_timing(True)
    for rep in range(a.reps):
        torch.cuda.synchronize(); t0 = time.time()
        psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=100 + rep, first_index=0, stream=st)
        rc = psf.last_status()
        dt = time.time() - t0
        tm = psf.get_timing()
        print(f"rep {rep}: status {rc}  {dt*1e3:.1f} ms  -> {B/dt:.0f} preimages/s")
        for nm, ms in tm:
            print(f"    {nm:28s} {ms:9.3f} ms")
        trmm = dict(tm)["k_trmm_f64"]
        print(f"    trmm: {psf.m*(psf.m+1)*B/trmm*1e-9:.2f} TFLOP/s (algorithmic)", flush=True)
    psf.enable_timing(False)
    t0 = time.time()
    psf.f_a_dev(e.data_ptr(), u2.data_ptr(), ok.data_ptr(), B, stream=st)
    torch.cuda.synchronize()
    print(f"f_a {time.time()-t0:.2f}s  A e == u: {bool((u2 == u).all())}  check_domain all: {bool(ok.all())}")
    nrm = (e.double() ** 2).sum(1).sqrt()
    print(f"|e| mean {nrm.mean().item():.1f} max {nrm.max().item():.1f} bound {a.s*a.r*psf.m**0.5:.1f}")
    print(f"max mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB (torch side only)")


if __name__ == "__main__":
    main()
