"""Call time of PSFGPV (C2 shape) and PSFGPVRing (C4 shape) over the batch size: one key each, samp_p_dev through device pointers, median of `reps` calls, the walk form the
handle chose (psfgpv_get_nearest_plane_form) -- where the forms change and whether a batch size falls between them.   python tools/np_batch_sweep.py [reps=5]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import tools_amd as T  # noqa: E402


def sweep(name, psf, n, sizes, reps, force=None):
    if force is not None:
        psf._debug_set_walk(force, 0)
    dev = torch.device("cuda:0")
    m = psf.m if hasattr(psf, "m") else psf.d
    st = torch.cuda.current_stream().cuda_stream
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    Bmax = max(sizes)
    u = torch.empty((Bmax, n), dtype=torch.int64, device=dev)
    psf.uniform_targets_dev(u.data_ptr(), Bmax, seed=7, first_index=0)
    for B in sizes:
        e = torch.empty((B, m), dtype=torch.int64, device=dev)
        call = lambda: psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=5, first_index=100, stream=st)
        call(); call()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize(); ev0.record(); call(); ev1.record(); torch.cuda.synchronize()
            ts.append(ev0.elapsed_time(ev1))
        ts.sort()
        assert psf.last_status() == 0
        form = psf.nearest_plane_form() if hasattr(psf, "nearest_plane_form") else None
        t = ts[len(ts) // 2]
        print(json.dumps({"type": name, "B": B, "ms": round(t, 4), "us_per_preimage": round(t / B * 1000, 3), "form": form}), flush=True)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
    sizes = [1, 4, 16, 64, 128, 256, 512, 768, 1024, 1025, 1536, 1537, 1792, 2047, 2048, 2049, 3072, 4096, 4097, 6144, 8192]
    _, n, q, _, s, _ = bench.CONFIGS["c2"]
    psf = T.PSFGPV(T.GadgetParameters.init_default(n, q), s)
    psf.trap_gen(3, export=False)
    sweep("PSFGPV c2", psf, n, sizes, reps)
    if "--forms" in sys.argv:
        sweep("PSFGPV c2, per block", psf, n, [256, 1024], reps, force=0)
        psf._debug_set_walk(-1, 0)
    psf.close()
    _, n, q, _, _, _ = bench.CONFIGS["c4"]
    gpr = T.GadgetParametersRing.init_default(n, q)
    import math
    s4 = ((2 * 2 * 1.005 * math.sqrt(n) + 1) * 2) * 4               # compute_s, gpv_ring.rs:296-298
    psf = T.PSFGPVRing(gpr, s4, 1.005)
    from tools_amd._ffi import check, lib
    import ctypes as C
    check(lib().psfring_trap_gen(psf._h, C.c_uint64(3)), "trap_gen")
    sweep("PSFGPVRing c4", psf, n, sizes, reps)
    if "--forms" in sys.argv:      # the launch-per-block form forced where the one-launch walk is the default
        sweep("PSFGPVRing c4, per block", psf, n, [256, 1024, 1025, 1280, 1536, 1537, 1792, 2047, 2048], reps, force=0)
        psf._debug_set_walk(-1, 0)


if __name__ == "__main__":
    main()
