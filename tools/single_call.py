#!/usr/bin/env python3
"""single_call.py -- latency of ONE samp_p call (the reference's unit of work: psf.rs:48-80, benches/psf.rs:38,63-65,90-92).

(a) C3 (PSFPerturbation n=512 q=2^30) at batch 1 .. 128 through psfp_samp_p_dev (device pointers, HIP-event and host wall time per call) and through
    the host-pointer psfp_samp_p; per-kernel HIP-event times of one call; the streaming product against the batch kernel (same bits);
(b) the reference's three criterion sets (GPV n=8, Perturbation n=8, n=64; one preimage per call) as GPU microseconds per call.
Writes a JSON record (default profiles/r04_single_call.json).
"""
# the PSF_* switches this script sets are alive in the experiments build only (make -C tools_amd/csrc exp); the release library reads none of them
import os as _os
_os.environ.setdefault("PSF_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tools_amd", "lib", "libpsf_mi355x_exp.so"))
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed_calls(fn, sync, reps):
    """median / min host wall time per call, each call synchronised (a latency, not a throughput)"""
    ts = []
    for _ in range(reps):
        sync()
        t0 = time.perf_counter()
        fn()
        sync()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return {"median_ms": round(ts[len(ts) // 2] * 1e3, 4), "min_ms": round(ts[0] * 1e3, 4), "reps": reps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_single_call.json"))
    ap.add_argument("--batches", default="1,15,16,17,32,63,64,65,128")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--skip-c3", action="store_true")
    ap.add_argument("--skip-sets", action="store_true")
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--logq", type=int, default=30)
    ap.add_argument("--stream-max", default=None, help="PSF_TRMM_STREAM_MAX of the streaming arm (default: the library's)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import tools_amd as T
    from tools_amd._ffi import lib, check

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    sync = torch.cuda.synchronize
    stream = torch.cuda.current_stream().cuda_stream
    rec = {"what": "latency of one samp_p call; device-pointer entry point unless stated", "c3": {}, "reference_bench_sets": []}

    if not args.skip_c3:
        n, q, r, s = args.n, 2 ** args.logq, 9.0, 512.0
        psf = T.PSFPerturbation(T.GadgetParameters.init_default(n, q), r, s, device=0)
        t0 = time.time()
        check(lib().psfp_trap_gen(psf._h, C.c_uint64(3)), "trap_gen")
        sync()
        m = psf.m
        key_bytes = m * (m + 1) // 2 * 8
        rec["c3"]["shape"] = {"n": n, "q": q, "m": m, "key_bytes": key_bytes, "trap_gen_s": round(time.time() - t0, 2)}
        for B in [int(x) for x in args.batches.split(",")]:
            u = torch.empty((B, n), dtype=torch.int64, device=dev)
            psf.uniform_targets_dev(u.data_ptr(), B, seed=7, first_index=0, stream=stream)
            row = {}
            outs = {}
            for label, smax in (("stream", None), ("batch_kernel", "0")):
                if smax is None and args.stream_max is not None:
                    os.environ["PSF_TRMM_STREAM_MAX"] = args.stream_max
                elif smax is None:
                    os.environ.pop("PSF_TRMM_STREAM_MAX", None)
                else:
                    os.environ["PSF_TRMM_STREAM_MAX"] = smax
                e = torch.zeros((B, m), dtype=torch.int64, device=dev)
                call = lambda: psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=1000, first_index=0, stream=stream)
                call(); call()
                sync()
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                evs = []
                for _ in range(args.reps):
                    sync()
                    ev0.record(); call(); ev1.record()
                    sync()
                    evs.append(ev0.elapsed_time(ev1))
                evs.sort()
                wall = timed_calls(call, sync, args.reps)
                psf.enable_timing(True)
                call()
                tm = dict(psf.get_timing())
                psf.enable_timing(False)
                assert psf.last_status() == 0
                row[label] = {"event_ms_median": round(evs[len(evs) // 2], 4), "event_ms_min": round(evs[0], 4), "host_wall": wall,
                              "kernels_ms": {k: round(v, 4) for k, v in tm.items()}}
                outs[label] = e.clone()
            os.environ.pop("PSF_TRMM_STREAM_MAX", None)
            row["same_bits_as_batch_kernel"] = bool((outs["stream"] == outs["batch_kernel"]).all().item())
            # validity: A e = u and check_domain
            e = outs["stream"]
            u2 = torch.empty_like(u); ok = torch.empty((B,), dtype=torch.uint8, device=dev)
            psf.f_a_dev(e.data_ptr(), u2.data_ptr(), ok.data_ptr(), B, stream=stream)
            sync()
            row["valid"] = bool((u2 == u).all().item()) and bool(ok.all().item())
            # host-pointer entry point (what shim/src/lib.rs PSF::samp_p binds)
            uh = u.cpu().numpy().astype(np.uint64)
            hostcall = lambda: psf.samp_p(uh, seed=1000, first_index=0)
            eh = hostcall()
            row["host_pointer"] = timed_calls(hostcall, sync, max(5, args.reps // 2))
            row["host_pointer"]["same_rows"] = bool((torch.from_numpy(eh.astype(np.int64)).to(dev) == e).all().item())
            trmm = row["stream"]["kernels_ms"].get("k_trmm_f64")
            if trmm:
                row["stream"]["product_hbm_GBps"] = round(key_bytes / (trmm * 1e-3) / 1e9, 1)
                row["stream"]["product_frac_of_8TBps"] = round(key_bytes / (trmm * 1e-3) / 8e12, 4)
            rec["c3"][f"B{B}"] = row
            print(f"[c3] B={B}: stream {row['stream']['event_ms_median']} ms (product {trmm} ms), batch kernel {row['batch_kernel']['event_ms_median']} ms, "
                  f"host-pointer {row['host_pointer']['median_ms']} ms, same bits {row['same_bits_as_batch_kernel']}, valid {row['valid']}", flush=True)
            print("      kernels:", row["stream"]["kernels_ms"], flush=True)
        psf.close()

    if not args.skip_sets:
        cpu = {}
        try:
            with open(os.path.join(ROOT, "profiles", "r02_cpu_faithful_epyc.json")) as fh:
                for st in json.load(fh)["sets"]:
                    cpu[st["bench"]] = st
        except Exception:
            pass
        sets = [("PSF GPV n=8", "gpv", 8, 128, None, 30 * math.log2(8)),
                ("PSF Perturbation n=8", "pert", 8, 128, math.log2(8), 30.0),
                ("PSF Perturbation n=64", "pert", 64, 128, math.log2(64), 100.0)]
        for name, kind, n, q, r, s in sets:
            gp = T.GadgetParameters.init_default(n, q)
            if kind == "gpv":
                psf = T.PSFGPV(gp, s, device=0)
                psf.trap_gen(1, export=False)
            else:
                psf = T.PSFPerturbation(gp, r, s, device=0)
                check(lib().psfp_trap_gen(psf._h, C.c_uint64(1)), "trap_gen")
            m = psf.m
            u = torch.empty((1, n), dtype=torch.int64, device=dev)
            psf.uniform_targets_dev(u.data_ptr(), 1, seed=7, first_index=0, stream=stream)
            e = torch.zeros((1, m), dtype=torch.int64, device=dev)
            call = lambda: psf.samp_p_dev(u.data_ptr(), e.data_ptr(), 1, seed=5, first_index=0, stream=stream)
            for _ in range(5):
                call()
            sync()
            dev_wall = timed_calls(call, sync, 200)
            # back-to-back calls on one stream (a signing loop): launch throughput, no sync per call
            sync(); t0 = time.perf_counter()
            for i in range(500):
                psf.samp_p_dev(u.data_ptr(), e.data_ptr(), 1, seed=5 + i, first_index=0, stream=stream)
            sync()
            b2b = (time.perf_counter() - t0) / 500
            uh = u.cpu().numpy().astype(np.uint64)
            hostcall = lambda: psf.samp_p(uh, seed=5, first_index=0)
            hostcall()
            host = timed_calls(hostcall, sync, 200)
            row = {"bench": name, "n": n, "q": q, "m": m, "gpu_dev_ptr_us_per_call": round(dev_wall["median_ms"] * 1e3, 1),
                   "gpu_dev_ptr_back_to_back_us_per_call": round(b2b * 1e6, 1),
                   "gpu_host_ptr_us_per_call": round(host["median_ms"] * 1e3, 1)}
            if name in cpu:
                row["cpu_port_us_per_call"] = round(cpu[name]["port_s_per_call"] * 1e6, 1)
                row["cpu_gmp_faithful_us_per_call"] = round(cpu[name]["faithful_s_per_call"] * 1e6, 1)
            rec["reference_bench_sets"].append(row)
            print("[set]", row, flush=True)
            psf.close()

    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(rec, fh, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
