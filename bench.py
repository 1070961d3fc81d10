#!/usr/bin/env python3
"""bench.py -- preimages/sec of PSF::samp_p on MI355X (BASELINE.json metric), one process per GPU.

A "step" is one samp_p pass over one batch of synthetic uniform syndromes (benches/psf.rs:35,60,87: uniform
target, trapdoor generated outside the timed region).  Default workload = BASELINE.json configs[2], the one
the metric is quoted on: PSFPerturbation (MP12) over Z_q, n=512, q=2^30, batch=4096 per GPU, r=9, s=512.
Inputs (u) and the key are resident in HBM before the timed region; the preimages stay in HBM.  With N>1
ranks the batch is sharded by global preimage index (weak scaling: 4096 per GPU) and the result is gathered
to rank 0 over RCCL inside the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant kernel
(k_trmm_f64_big: x = sqrt(Sigma_2) d on the f64 MFMA pipe) and `cpu_baseline` (the CPU oracle timed on this
host's cores, rank 0, N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL on this driver
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (scheme, n, q, r, s, batch)
    "c3": ("PSFPerturbation", 512, 2**30, 9.0, 512.0, 4096),
    "c3prime": ("PSFPerturbation", 512, 1073741789, 9.0, 512.0, 4096),
    "bench64": ("PSFPerturbation", 64, 128, 6.0, 100.0, 4096),     # benches/psf.rs:78-93
    "c1": ("PSFPerturbation", 8, 64, 3.0, 25.0, 1),                 # README.md:62-66
    "c2": ("PSFGPV", 256, 3329, None, 1024.0, 1024),                # BASELINE.json configs[1]
    "c2s240": ("PSFGPV", 256, 3329, None, 240.0, 1024),             # the same at the reference bench's own rule s = 30 log2 n (benches/psf.rs:32), SURVEY.md 8d
    "c4": ("PSFGPVRing", 256, 3329, None, 0.0, 4096),               # BASELINE.json configs[3]; s = compute_s(256), gpv_ring.rs:296-298
    "c5": ("PSFPerturbation", 1024, 2**60, 10.0, 1024.0, 8192),     # BASELINE.json configs[4]: batch 65536 over 8 GPUs = 8192 per GPU
}                                                                   # (60.6 GB key per GPU, trap_gen ~10 s; run with --gpus 8 --config c5)
PEAK_F64_MFMA_TFLOPS = 78.6   # MI355X datasheet FP64 matrix; measured 77.3 by tools/probe_mfma_f64.hip (profiles/r01_probe_mfma_f64.log)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="preimages per GPU per step (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="preimages timed on the CPU (default: 32 per thread)")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL gather of the result (N>1)")
    ap.add_argument("--structured", action="store_true", help="PSFPerturbation with the structured square root of Sigma_2 (PSFP_FLAG_STRUCTURED_SQRT): a labelled, different algorithm; the headline stays on the dense path")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-call latency legs (batch 1 / 16 / 64)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short c2 / c4 / c3prime legs the default invocation appends (\"other_configs\")")
    ap.add_argument("--force-dist", action="store_true", help="run the N>1 code path (launcher, process group, barriers, gather, reductions) on a one-rank RCCL group")
    ap.add_argument("--oversubscribe", action="store_true", help="REHEARSAL of the N>1 path on a box with fewer GPUs: rank r uses device r %% (GPUs there are), the process group is gloo with a host-staged gather; "
                    "exercises launcher, sharding by global index, barriers, reductions and the gather with real kernels -- labelled in the line, never a scaling number")
    ap.add_argument("--verify-gather", action="store_true", help="rank 0 recomputes every rank's last step itself (same key, the rank's first_index) and compares with the gathered rows")
    ap.add_argument("--emulate-world", type=int, default=0, help="with --force-dist: size the gather's buffers as rank 0 of a job of this many ranks would (C5 readiness: 8)")
    ap.add_argument("--multi-handle", action="store_true", help="torch-free scaling mode: ONE process drives --gpus devices through psfp_samp_p_multi (host buffers, one worker thread per handle)")
    ap.add_argument("--launch-timeout", type=float, default=0.0, help="seconds after which the self-launcher stops every rank (0 = none)")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.multi_handle:
        return multi_handle_main(args)
    if (args.gpus > 1 or args.force_dist) and "WORLD_SIZE" not in os.environ:
        return launcher_main(args)                 # this process becomes the parent of N ranks and never touches the GPU
    rank_main(args)


def launcher_main(args):
    """`python3 bench.py --gpus N` is a complete command: without an external launcher (no WORLD_SIZE in the environment) the parent starts N fresh
    children of this script -- before anything here has imported torch.cuda or loaded the HIP library, and by subprocess, never exec -- one per GPU with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relays rank 0's output (its JSON line is the last line of stdout) and exits
    non-zero if any rank does.  `python -m torch.distributed.run ... bench.py --gpus N` keeps working: WORLD_SIZE is then set and this is skipped."""
    from tools_amd import launch
    have = launch.visible_gpu_count()
    if args.oversubscribe and have >= 1:
        have = args.gpus                             # ranks share the devices there are (rehearsal mode)
    if have < args.gpus:
        print(f"[bench] --gpus {args.gpus} but this host shows {have} GPU(s) (KFD topology / *_VISIBLE_DEVICES): not starting any rank", file=sys.stderr)
        sys.exit(2)
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    rc = launch.run_ranks(cmd, args.gpus, timeout=args.launch_timeout or None)
    sys.exit(rc)


def rank_main(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    import tools_amd as T

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: the launcher and the flag disagree", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("[bench] no GPU: tools_amd has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    if args.oversubscribe:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL refuses two ranks on one device; the rehearsal mode therefore runs its collectives over gloo (host tensors)
        dist.init_process_group(backend="gloo" if args.oversubscribe else "nccl", rank=rank, world_size=world)

    run = run_config(args.config, args.batch, args.steps, args.warmup, local_rank, rank, world, multi, dev, structured=args.structured,
                     gather=multi and not args.no_gather, force_dist=args.force_dist, alloc_world=args.emulate_world, verify_gather=args.verify_gather)
    psf, scheme, n, q, r, s, m, B, u, e = (run[k] for k in ("psf", "scheme", "n", "q", "r", "s", "m", "B", "u", "e"))
    stream, first_index, kern_ms, elapsed, valid, do_gather = (run[k] for k in ("stream", "first_index", "kern_ms", "elapsed", "valid", "do_gather"))

    # latency of ONE call with few preimages -- the reference's unit of work (psf.rs:48-80: one samp_p call = one preimage; benches/psf.rs:38,63-65,90-92)
    # and the regime where reading the key from HBM once, not the FP64 pipe, is the roof.  Outside the timed region of the metric.
    latency = None
    if rank == 0 and scheme == "PSFPerturbation" and not args.no_latency and not args.structured:
        latency = single_call_latency(psf, u, e, m, first_index, stream, args.config)
    elif rank == 0 and not args.no_latency:
        latency = nearest_plane_call_latency(psf, u, m, first_index, stream, args.config)

    total = B * world * args.steps
    value = total / elapsed
    out = None
    if rank == 0:
        roof = roofline(scheme, psf, m, B, kern_ms, args.config, args.structured, run["np_form"])
        out = {
            "metric": "preimages/sec (whole node) + HBM-BW% for samp_p, n=512 q~2^30 batch=4096",
            "value": round(value, 2), "unit": "preimages/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64+int64" if scheme == "PSFPerturbation" else "f64+int8+int64", "data": "synthetic",
            "config": {"workload": f"{scheme} samp_p n={n} q={q} k={run['gp'].k} m={m} r={r} s={s} batch={B}/GPU ({args.config})",
                       "global_batch": B * world, "parallelism": f"batch-sharded x{world}" + (" + overlapped RCCL gather of int32 rows to rank 0" if do_gather else "")},
            "valid": valid, "kernels_ms": {k: round(v, 3) for k, v in kern_ms.items()}, "trap_gen_s": round(run["trap_gen_s"], 2),
            "roofline": roof,
        }
        if multi:
            # what the process group really was: ranks_seen is dist.get_world_size(), the per-rank step times are each rank's own clock over the same K steps
            out["ranks_seen"] = run["ranks_seen"]
            out["ms_per_step_ranks"] = {"min": round(min(run["rank_elapsed"]) / args.steps * 1e3, 3), "max": round(max(run["rank_elapsed"]) / args.steps * 1e3, 3),
                                        "all": [round(x / args.steps * 1e3, 3) for x in run["rank_elapsed"]]}
            out["launcher"] = "bench.py (self-spawned ranks, tools_amd/launch.py)" if os.environ.get("PSF_LAUNCHED_BY") else "external (torch.distributed.run or equivalent)"
            if run["gather_info"]:
                out["gather"] = run["gather_info"]
            if args.oversubscribe:
                out["oversubscribed"] = f"{world} ranks on {torch.cuda.device_count()} GPU(s), gloo process group with a host-staged gather: a rehearsal of the N>1 code path, not a scaling measurement"
                out["config"]["parallelism"] = out["config"]["parallelism"].replace("RCCL gather", "gloo gather (host-staged)")
            if run.get("gather_verified") is not None:
                out["gather_verified"] = run["gather_verified"]
        if latency:
            out["latency"] = latency
        if args.config != "c3" or args.structured:
            out["metric"] = f"preimages/sec for samp_p ({args.config}" + (", structured sqrt(Sigma_2): labelled opt-in, not the parity path" if args.structured else "") + ")"
        if args.structured:
            out["config"]["workload"] += " [PSFP_FLAG_STRUCTURED_SQRT]"
        key_gb = (m * (m + 1) // 2) * 8 / 1e9 if scheme == "PSFPerturbation" else m * m * 12 / 1e9
        if world == 1 and not args.no_cpu_baseline and key_gb > 16:
            # the port would need the key twice in host memory (export buffer + its own copy): not timed at this size.  The same shape is
            # checked against the oracle stage by stage, with the factor streamed in row blocks, by tests/test_gpu_full_size.py
            out["cpu_baseline"] = {"value": None, "unit": "preimages/s", "cores": 0, "kind": "port",
                                   "sample": f"not timed: the key is {key_gb:.0f} GB and the port holds it twice in host memory"}
        elif world == 1 and not args.no_cpu_baseline and args.structured:
            out["cpu_baseline"] = {"value": None, "unit": "preimages/s", "cores": 0, "kind": "port",
                                   "sample": "not timed: the CPU leg measures the reference's algorithm (dense factor); see the line without --structured"}
        elif world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scheme, psf, n, q, r, s, u, e, first_index, 1000 + args.warmup + args.steps - 1, args.cpu_sample)

    # The other single-GPU configurations of BASELINE.json, a few steps each, so that the driver's record carries them too (default invocation only:
    # the headline above is timed first and alone; these legs run after it, each with its own key, outside every timed region of the metric).
    if (rank == 0 and world == 1 and not multi and args.config == "c3" and not args.structured and not args.no_other_configs and not args.batch):
        del psf, u, e
        run.clear()
        torch.cuda.empty_cache()
        out["other_configs"] = other_configs(local_rank, dev)

    def flush_c_stdio():                            # RCCL writes its version banner through C stdio (NCCL_DEBUG=VERSION on the GPU boxes)
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()

    if multi:                                       # every rank empties its buffers, then the group is torn down, then rank 0 speaks last
        flush_c_stdio()
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        if world > 1:
            time.sleep(0.5)                         # the other ranks have nothing buffered any more; let their last writes land
        print(json.dumps(out), flush=True)
    if not valid:
        sys.exit(4)


def run_config(cfg, batch, steps, warmup, local_rank, rank, world, multi, dev, structured=False, gather=False, force_dist=False, key_seed=3, alloc_world=0, verify_gather=False):
    """Key generation (outside the timed region, benches/psf.rs:36,61,88), `warmup` untimed steps, then exactly `steps` samp_p passes over one batch of
    uniform syndromes between two fences (device synchronise + barrier + device synchronise), and the correctness gate on the last step's rows."""
    import torch
    import torch.distributed as dist
    import tools_amd as T
    from tools_amd._ffi import lib, check
    from tools_amd.shard import shard_range, AsyncRowGather
    scheme, n, q, r, s, cfg_batch = CONFIGS[cfg]
    B = batch or cfg_batch
    t0 = time.time()
    if scheme == "PSFPerturbation":
        gp = T.GadgetParameters.init_default(n, q)
        psf = T.PSFPerturbation(gp, r, s, device=local_rank, structured=structured)
        check(lib().psfp_trap_gen(psf._h, C.c_uint64(key_seed)), "trap_gen")   # every rank: same seed -> same key
        m = psf.m
    elif scheme == "PSFGPV":
        gp = T.GadgetParameters.init_default(n, q)
        psf = T.PSFGPV(gp, s, device=local_rank)
        psf.trap_gen(key_seed, export=False)
        m = psf.m
    else:
        import math
        s = s or ((2 * 2 * 1.005 * math.sqrt(n) + 1) * 2) * 4               # compute_s, gpv_ring.rs:296-298
        gp = T.GadgetParametersRing.init_default(n, q)
        psf = T.PSFGPVRing(gp, s, 1.005, device=local_rank)
        check(lib().psfring_trap_gen(psf._h, C.c_uint64(key_seed)), "trap_gen")
        m = psf.d
    torch.cuda.synchronize()
    t_trapgen = time.time() - t0

    stream = torch.cuda.current_stream().cuda_stream
    first_index, _ = shard_range(rank, world, B)               # global preimage index of this rank's row 0
    u = torch.empty((B, n), dtype=torch.int64, device=dev)
    e = torch.empty((B, m), dtype=torch.int64, device=dev)
    psf.uniform_targets_dev(u.data_ptr(), B, seed=7, first_index=first_index, stream=stream)
    do_gather = bool(gather)
    gatherer = AsyncRowGather(B, m, dev, dst=0, force=force_dist, alloc_world=alloc_world) if do_gather else None     # step i's rows travel while step i+1 computes
    gather_info = None
    if gatherer is not None:
        free_b, total_b = torch.cuda.mem_get_info(dev)
        gather_info = {"depth": gatherer.depth, "buffers_sized_for_ranks": gatherer.alloc_world, "bytes_per_depth": AsyncRowGather.bytes_per_depth(B, m, gatherer.alloc_world, rank == 0),
                       "hbm_free_GB_after_allocation": round(free_b / 1e9, 2), "hbm_total_GB": round(total_b / 1e9, 2)}

    def step(i):
        psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=1000 + i, first_index=first_index, stream=stream)
        if do_gather:
            gatherer.submit(e)

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(warmup):
        step(i)
    if do_gather:
        gatherer.finish()
    if psf.last_status() != 0:
        raise RuntimeError("device-side sampler failure during warmup")
    psf.enable_timing(True)
    kern_ms = {}
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    gathered = None
    if do_gather:
        gathered = gatherer.finish()       # every step's rows have reached rank 0 inside the timed region
    fence()
    own = elapsed = time.perf_counter() - t0
    # per-kernel HIP-event times of the last step (events were recorded on the launch stream, no host sync in the loop)
    tm = psf.get_timing()
    for nm, ms in (tm.items() if isinstance(tm, dict) else tm):
        kern_ms[nm] = ms
    psf.enable_timing(False)
    status = psf.last_status()
    np_form = nearest_plane_form(psf) if hasattr(psf, "nearest_plane_form") else None      # of the timed steps (the latency legs launch other forms)
    rank_elapsed, ranks_seen = [own], 1
    cdev = torch.device("cpu") if (multi and dist.get_backend() == "gloo") else dev      # the small reductions of a gloo group run on host tensors
    if multi:
        ranks_seen = dist.get_world_size()
        mine = torch.tensor([own], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(ranks_seen)]
        dist.all_gather(every, mine)
        rank_elapsed = [float(t.item()) for t in every]
        elapsed = max(rank_elapsed)                  # MAX over ranks
    gather_verified = None
    if verify_gather and gathered is not None and rank == 0 and steps > 0:
        # rank 0 holds the same key: it recomputes every rank's last step from that rank's global indices and compares with what the gather delivered
        gather_verified = True
        u_r = torch.empty_like(u)
        e_r = torch.empty_like(e)
        for rr in range(world):
            fi = shard_range(rr, world, B)[0]
            psf.uniform_targets_dev(u_r.data_ptr(), B, seed=7, first_index=fi, stream=stream)
            psf.samp_p_dev(u_r.data_ptr(), e_r.data_ptr(), B, seed=1000 + warmup + steps - 1, first_index=fi, stream=stream)
            torch.cuda.synchronize()
            gather_verified = gather_verified and bool((gathered[rr].to(torch.int64).cpu() == e_r.cpu()).all().item())

    # correctness gate on the last step's output: A e == u and check_domain for every row
    u2 = torch.empty_like(u)
    ok = torch.empty((B,), dtype=torch.uint8, device=dev)
    psf.f_a_dev(e.data_ptr(), u2.data_ptr(), ok.data_ptr(), B, stream=stream)
    torch.cuda.synchronize()
    valid = bool((u2 == u).all().item()) and bool(ok.all().item()) and status == 0
    if gather_verified is False:
        valid = False
    if multi:                                       # every rank's rows must pass, not only rank 0's
        vt = torch.tensor([1 if valid else 0], dtype=torch.int32, device=cdev)
        dist.all_reduce(vt, op=dist.ReduceOp.MIN)
        valid = bool(vt.item())
    return {"psf": psf, "scheme": scheme, "n": n, "q": q, "r": r, "s": s, "m": m, "gp": gp, "B": B, "u": u, "e": e, "stream": stream,
            "first_index": first_index, "kern_ms": kern_ms, "elapsed": elapsed, "valid": valid, "do_gather": do_gather, "trap_gen_s": t_trapgen,
            "rank_elapsed": rank_elapsed, "ranks_seen": ranks_seen, "gather_info": gather_info, "np_form": np_form, "gather_verified": gather_verified}


def nearest_plane_form(psf):
    """The kernels the last PSFGPV / PSFGPVRing samp_p call really launched for the walk of gpv.rs:160, from the handle (psfgpv_get_nearest_plane_form)."""
    try:
        form, G, blocks, fallbacks = psf.nearest_plane_form()
    except Exception:
        return "nearest plane", {}
    walk = {1: f"k_np_walk<{G}> (one launch: sampler + updater workgroups)", 0: f"{blocks} x k_np_step<{G}> (one launch per 64-row block)",
            2: f"2 x {blocks} x k_np_step<{G}> (two column ranges side by side on two streams, one launch per 64-row block each)"}.get(form, f"form {form}")
    return f"nearest plane: k_np_project + {walk} + k_np_combine8_fused", {"form": {1: "one launch", 0: "launch per block", 2: "launch per block, two halves side by side"}.get(form, str(form)),
                                                                           "G": G, "blocks": blocks, "walks_rerun_without_waits": fallbacks}


def roofline(scheme, psf, m, B, kern_ms, cfg, structured=False, np_form=None):
    trmm = kern_ms.get("k_trmm_f64")
    mL = psf.m_bar if (scheme == "PSFPerturbation" and structured) else m
    flops_per_launch = float(mL) * (mL + 1) * B          # mL(mL+1)/2 fma per preimage (m^2 flop; structured: the m_bar x m_bar block)
    roof = None
    if trmm:
        ach = flops_per_launch / (trmm * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "k_trmm_f64_big", "achieved": round(ach, 3), "peak": PEAK_F64_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(ach / PEAK_F64_MFMA_TFLOPS, 4), "traffic": load_traffic(cfg + (":structured" if structured else ""), B),
                "traffic_source": "profiles/trmm_traffic.json (rocprofv3 PMC passes; null when psf_kernels.hpp changed since)",
                "launch_ms": round(trmm, 3), "flops_per_launch": flops_per_launch,
                "clock_note": "peak is priced at the nominal 2.4 GHz; under this kernel the shader clock sits at 2.25-2.37 GHz (tools/trmm_clock_probe.py, profiles/r02_probe_trmm.log)"}
        if roof["traffic"]:                      # the HBM side of the same launch, for the metric's "HBM-BW%"
            roof["hbm_GBps"] = round(roof["traffic"] / (trmm * 1e-3) / 1e9, 1)
            roof["hbm_frac_of_peak"] = round(roof["traffic"] / (trmm * 1e-3) / (PEAK_HBM_GBS * 1e9), 4)
    npl = kern_ms.get("nearest_plane")
    if npl and roof is None:
        # PSFGPV / PSFGPVRing: the nearest plane (gpv.rs:160) as a whole -- the walk (sampler workgroups + FP64-MFMA update tiles), then the int8-MFMA
        # recombination.  Algorithmic work per preimage: 2 d^2 FP64 flop (projection + update); the walk itself is d sequential draws per
        # preimage, so the phase is latency bound, not MFMA bound.
        flops = 2.0 * float(m) * float(m) * B
        ach = flops / (npl * 1e-3) / 1e12
        kname, form = np_form or nearest_plane_form(psf)
        roof = {"bound": "mfma", "kernel": kname, "achieved": round(ach, 3),
                "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F64_MFMA_TFLOPS, 4),
                "traffic": load_traffic(cfg, B, "np_traffic.json", ("psf_np_kernels.hpp",)),
                "traffic_source": "profiles/np_traffic.json: rocprofv3 PMC passes over tools/bin/np_harness (C++ over the C ABI), every k_np_* launch of one samp_p call; null when psf_np_kernels.hpp changed since",
                "launch_ms": round(npl, 3), "flops_per_launch": flops, "serial_steps": int(m),
                "us_per_serial_step": round(npl * 1e3 / m, 4), "walk": form,
                "note": "latency bound: d sequential SampleZ draws per preimage; frac is the FP64 share of the phase"}
    return roof


def other_configs(local_rank, dev, steps=5, warmup=1):
    """c2 (BASELINE configs[1]), c4 (configs[3]) and c3prime (configs[2] with a prime modulus): `steps` timed steps each, same fences and the same gate as
    the headline, with the per-kernel HIP-event times and the roofline object of each.  About 2 s in all (three key generations of 0.2-0.4 s + 0.4 s of steps)."""
    import torch
    res = {}
    for cfg in ("c2", "c4", "c3prime"):
        t0 = time.time()
        try:
            run = run_config(cfg, 0, steps, warmup, local_rank, 0, 1, False, dev)
            scheme, B = run["scheme"], run["B"]
            res[cfg] = {"workload": f"{scheme} n={run['n']} q={run['q']} m={run['m']} s={run['s']} batch={B}",
                        "ms_per_step": round(run["elapsed"] / steps * 1e3, 3), "value": round(B * steps / run["elapsed"], 2), "unit": "preimages/s",
                        "steps": steps, "warmup": warmup, "valid": run["valid"], "kernels_ms": {k: round(v, 3) for k, v in run["kern_ms"].items()},
                        "trap_gen_s": round(run["trap_gen_s"], 2), "roofline": roofline(scheme, run["psf"], run["m"], B, run["kern_ms"], cfg, np_form=run["np_form"])}
            run["psf"].close()
            run.clear()
        except Exception as exc:                    # a failed side leg is reported in the line, it does not take the headline with it
            res[cfg] = {"error": f"{type(exc).__name__}: {exc}"}
        torch.cuda.empty_cache()
        res[cfg]["wall_s"] = round(time.time() - t0, 2)
    return res


def multi_handle_main(args):
    """The second N-GPU route, without torch and without a process group: ONE process holds one handle per device, every handle the same key (same
    trap_gen seed), and psfp_samp_p_multi (include/psf_mi355x.h) cuts each step's rows into contiguous shares, one worker thread per handle -- the route a
    Rust caller of the shim would take.  Host buffers: the targets are uploaded and the rows come back over PCIe inside the timed region, so `value` here
    is the PCIe-inclusive rate of that entry point (said in the line), not the HBM-resident figure of the default mode."""
    import numpy as np
    import tools_amd as T
    from tools_amd._ffi import lib, check
    from tools_amd.psf import samp_p_multi, multi_timing
    scheme, n, q, r, s, batch = CONFIGS[args.config]
    if scheme != "PSFPerturbation":
        print("[bench] --multi-handle drives psfp_samp_p_multi: PSFPerturbation configurations only", file=sys.stderr)
        sys.exit(2)
    name = C.create_string_buffer(64)
    cus = C.c_int(0)
    for d in range(args.gpus):                      # fail fast and clearly when the node has fewer devices
        if lib().psf_device_info(d, name, 64, C.byref(cus)) != 0:
            print(f"[bench] --multi-handle --gpus {args.gpus}: device {d} is not available", file=sys.stderr)
            sys.exit(2)
    B = (args.batch or batch) * args.gpus           # weak scaling: the config's batch per device
    gp = T.GadgetParameters.init_default(n, q)
    t0 = time.time()
    handles = [T.PSFPerturbation(gp, r, s, device=d) for d in range(args.gpus)]
    for h in handles:
        check(lib().psfp_trap_gen(h._h, C.c_uint64(3)), "trap_gen")
    t_trapgen = time.time() - t0
    m = handles[0].m
    rng = np.random.default_rng(7)
    u = rng.integers(0, q, size=(B, n), dtype=np.uint64)
    e = np.zeros((B, m), dtype=np.int64)
    for i in range(args.warmup):
        samp_p_multi(handles, u, seed=1000 + i, out=e)
    windows = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        samp_p_multi(handles, u, seed=1000 + args.warmup + i, out=e)
        windows.append([multi_timing(h) for h in handles])
    elapsed = time.perf_counter() - t0
    chk = min(B, 256)                                # A e = u on the first rows of every handle's share, the domain check on the same rows
    rows = np.unique(np.concatenate([np.arange(d * (B // args.gpus), d * (B // args.gpus) + min(chk, B // args.gpus)) for d in range(args.gpus)]))
    valid = bool((handles[0].f_a(e[rows]) == u[rows]).all()) and bool(handles[0].check_domain(e[rows]).all())
    last = windows[-1]
    out = {"metric": f"preimages/sec for samp_p ({args.config}, psfp_samp_p_multi: one process, host buffers, PCIe inclusive)",
           "value": round(B * args.steps / elapsed, 2), "unit": "preimages/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64+int64", "data": "synthetic",
           "config": {"workload": f"{scheme} samp_p n={n} q={q} k={gp.k} m={m} r={r} s={s} batch={B // args.gpus}/GPU ({args.config})", "global_batch": B,
                      "parallelism": f"psfp_samp_p_multi over {args.gpus} handle(s), one worker thread per device, contiguous row shares (psf_shard_range)"},
           "valid": valid, "trap_gen_s": round(t_trapgen, 2), "transport": "host buffers: u uploaded, e downloaded inside the timed region",
           "handle_windows_ms": [{"device": d, "launched": round(w[0], 3), "done": round(w[1], 3)} for d, w in enumerate(last)],
           "windows_overlap": bool(max(w[0] for w in last) < min(w[1] for w in last)) if all(w[1] >= 0 for w in last) else None,
           "roofline": None, "cpu_baseline": None}
    print(json.dumps(out), flush=True)
    if not valid:
        sys.exit(4)


def nearest_plane_call_latency(psf, u, m, first_index, stream, cfg, reps=15):
    """One PSFGPV / PSFGPVRing samp_p call with 1 and 16 preimages (gpv.rs:152-161, gpv_ring.rs:160-212; what benches/psf.rs:26-39 times at n = 8) through the
    device-pointer entry point: median of `reps` synchronised HIP-event times.  The call is d dependent draws per preimage whatever the batch -- a latency
    chain, not a throughput problem -- so the figure beside it is the time per serial step."""
    import torch
    out = {"bound": "serial chain of d draws", "serial_steps": int(m), "entry_point": "samp_p_dev (device pointers)", "reps": reps}
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e = torch.empty((min(16, u.shape[0]), m), dtype=torch.int64, device=u.device)
    for B in (1, 16):
        if B > u.shape[0]:
            continue
        call = lambda: psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=77, first_index=first_index, stream=stream)
        call(); call()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            ev0.record(); call(); ev1.record()
            torch.cuda.synchronize()
            ts.append(ev0.elapsed_time(ev1))
        ts.sort()
        out[f"{cfg}_b{B}_ms"] = round(ts[len(ts) // 2], 4)
        out[f"{cfg}_b{B}_min_ms"] = round(ts[0], 4)
        out[f"us_per_serial_step_b{B}"] = round(ts[len(ts) // 2] * 1e3 / m, 4)
    if psf.last_status() != 0:
        out["status"] = "sampler failure"
    return out


def single_call_latency(psf, u, e, m, first_index, stream, cfg="c3", reps=30):
    """One samp_p call at batch 1 / 16 / 32 / 64 through the device-pointer entry point: median of `reps` HIP-event times, each call synchronised on both
    sides (a latency, not a throughput), plus the per-kernel HIP-event times of one call.  The product of these calls is k_trmm_stream, bound by
    reading the factor (m(m+1)/2 doubles) from HBM once: `product_frac_b1` = those bytes / its launch time / 8 TB/s, `call_frac_b1` the same for the whole call."""
    import torch
    key_bytes = m * (m + 1) // 2 * 8
    out = {"bound": "hbm", "bytes": key_bytes, "peak_GBps": PEAK_HBM_GBS, "entry_point": "psfp_samp_p_dev (device pointers)", "reps": reps}
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e = torch.empty((min(64, u.shape[0]), m), dtype=torch.int64, device=u.device)      # its own rows: the step's output is still to be checked against the oracle
    for B in (1, 16, 32, 64):      # 1-16: k_trmm_stream (bound by reading the factor); 32: k_trmm_stream_wg32; 64: k_trmm_stream_wg (LDS-shared tiles, bound by the matrix pipe)
        if B > u.shape[0]:
            continue
        call = lambda: psf.samp_p_dev(u.data_ptr(), e.data_ptr(), B, seed=77, first_index=first_index, stream=stream)
        call(); call()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            ev0.record(); call(); ev1.record()
            torch.cuda.synchronize()
            ts.append(ev0.elapsed_time(ev1))
        ts.sort()
        psf.enable_timing(True)
        call()
        tm = dict(psf.get_timing())
        psf.enable_timing(False)
        out[f"{cfg}_b{B}_ms"] = round(ts[len(ts) // 2], 4)
        out[f"{cfg}_b{B}_min_ms"] = round(ts[0], 4)
        out[f"kernels_ms_b{B}"] = {k: round(v, 4) for k, v in tm.items()}
        if B == 1 and tm.get("k_trmm_f64"):
            gbps = key_bytes / (tm["k_trmm_f64"] * 1e-3) / 1e9
            out["product_ms_b1"] = round(tm["k_trmm_f64"], 4)
            out["achieved_GBps"] = round(gbps, 1)
            out["product_frac_b1"] = round(gbps / PEAK_HBM_GBS, 4)      # the product KERNEL against the 8 TB/s spec; ~6.3 TB/s is what a streaming read achieves (MI355X_MICROARCH.md)
            out["call_frac_b1"] = round(key_bytes / (out[f"{cfg}_b1_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)   # the whole call against the same roof
    if psf.last_status() != 0:
        out["status"] = "sampler failure"
    return out


def load_traffic(config, B, fname="trmm_traffic.json", sources=("psf_kernels.hpp",)):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes (profiles/trmm_traffic.json), or None.  The figure is a
    committed measurement, not a counter read during this run: it is only returned while the kernel source it was measured on is unchanged
    (sha256 of the kernel headers recorded beside it), otherwise the line says null rather than carry a stale number."""
    path = os.path.join(ROOT, "profiles", fname)
    try:
        with open(path) as fh:
            rec = json.load(fh)
        ent = rec.get(f"{config}:B{B}")
        if not ent:
            return None
        want = ent.get("kernel_source_sha256")
        if want and want != kernel_source_hash(sources):
            return None
        return ent.get("hbm_bytes_per_launch", ent.get("hbm_bytes_per_call"))
    except Exception:
        return None


def kernel_source_hash(sources=("psf_kernels.hpp",)):
    import hashlib
    hsh = hashlib.sha256()
    for f in sources:
        with open(os.path.join(ROOT, "tools_amd", "csrc", f), "rb") as fh:
            hsh.update(fh.read())
    return hsh.hexdigest()[:16]


def cpu_baseline(scheme, psf, n, q, r, s, u, e, first_index, seed, sample):
    """The CPU oracle (oracle/psf_oracle.c, a port: the Rust/FLINT reference cannot be built here) timed on this host
    on a bounded sample of the same workload, same key, same seed; its output must equal the GPU's rows.  Two legs: all host
    threads, and one thread.  The port is a cache-blocked AVX-512 / AVX2 kernel for x = sqrt(Sigma_2) d (bit-identical to the scalar
    chain) but scalar C for the samplers; it is a reported baseline, not a tuned BLAS-class code."""
    import numpy as np
    from oracle import oracle as O
    O.build()
    threads, quota_note = effective_cpus(O.num_threads())
    if scheme == "PSFPerturbation":
        S = sample or min(u.shape[0], 512)           # two groups of 256 preimages x 241 row panels: enough tasks for every thread, ~3 s on 16 cores
        A, (R, Lp, _) = psf.export_key()
        orc = O.PSFPerturbation(O.gadget_params_default(n, q), r, s)
        orc.load_key(A, R, Lp)
        del A, R, Lp
        how = "OpenMP over (group of 256 preimages, panel of 128 rows) tasks"
    elif scheme == "PSFGPV":
        S = sample or min(u.shape[0], 8 * threads)
        A, (bt, gt) = psf.export_key()
        orc = O.PSFGPV(O.gadget_params_default(n, q), s)
        orc.load_key(A, bt, gt)
        del A, bt, gt
        how = "OpenMP over preimages, elimination factored once per key"
    else:
        S = sample or min(u.shape[0], 8 * threads)
        a, rr, ee, bt, gt = psf.export_key()
        orc = O.PSFGPVRing(O.gadget_params_ring_default(n, q), s, 1.005)
        orc.load_key(a, rr, ee, gso_t=gt)
        how = "OpenMP over preimages, basis / elimination / GSO once per key"
    uh = u[:S].cpu().numpy().astype(np.uint64)
    t0 = time.perf_counter()
    e_cpu = orc.samp_p(seed, uh, first_index=first_index, nthreads=threads)
    dt = time.perf_counter() - t0
    same = bool((e_cpu.reshape(S, -1) == e[:S].cpu().numpy()).all())
    # single-thread leg on a smaller sample (about 10-20 s of work)
    S1 = max(1, min(S, 64 if scheme == "PSFPerturbation" else 8))      # ~5 s: the CPU legs together stay a minor share of the run
    t0 = time.perf_counter()
    e_one = orc.samp_p(seed, uh[:S1], first_index=first_index, nthreads=1)
    dt1 = time.perf_counter() - t0
    same = same and bool((e_one.reshape(S1, -1) == e[:S1].cpu().numpy()).all())
    model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except Exception:
        pass
    return {"value": round(S / dt, 3), "unit": "preimages/s", "cores": threads, "kind": "port",
            "sample": f"{S} of the batch's preimages (same key, seed and targets), {dt:.1f} s wall, {how}",
            "single_thread": {"value": round(S1 / dt1, 3), "threads": 1, "sample": f"{S1} preimages, {dt1:.1f} s wall"},
            "cpu": model, "matches_gpu_bitwise": same,
            "note": "port, lightly tuned: blocked AVX-512/AVX2 triangular product, scalar samplers; see profiles/ for the GMP 'faithful mode' legs" + quota_note}


def effective_cpus(omp_threads):
    """Threads the CPU leg may really use: the OpenMP default, capped by the cgroup CPU quota of the box (the GPU boxes expose every
    hardware thread to nproc but run under a quota of 16 CPUs; 128 threads on 16 CPUs' worth of time only add scheduling overhead)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            cap = max(1, int(int(quota) / int(period)))
            if cap < omp_threads:
                return cap, f"; threads capped at the cgroup CPU quota ({cap} of {omp_threads} hardware threads)"
    except (OSError, ValueError):
        pass
    return omp_threads, ""


if __name__ == "__main__":
    main()
