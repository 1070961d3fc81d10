#!/usr/bin/env python3
"""Copies the round-5 measurement pass (gpurun_out/r5_final/, written by tools/r5_final.sh on the GPU box) into profiles/ under per-round names and refreshes the
hash-tied traffic records (profiles/trmm_traffic.json, profiles/np_traffic.json) that bench.py reads for `roofline.traffic`."""
import hashlib
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r5_final")
DST = os.path.join(ROOT, "profiles")


def sha(files):
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "tools_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def last_json(path, key=None):
    try:
        with open(path) as fh:
            lines = [ln for ln in fh.read().splitlines() if ln.strip().startswith("{") and (key is None or f'"{key}"' in ln)]
        return json.loads(lines[-1]) if lines else None
    except (OSError, ValueError):
        return None


COPY = {
    "bench_c3.json": "r05_bench_c3.json", "bench_c3prime.json": "r05_bench_c3prime.json", "bench_c2.json": "r05_bench_c2.json", "bench_c2s240.json": "r05_bench_c2s240.json",
    "bench_c4.json": "r05_bench_c4.json", "bench_c3_structured.json": "r05_bench_c3_structured.json", "bench_c5_one_gpu.json": "r05_bench_c5_one_gpu.json",
    "bench_c2_launch_per_block.json": "r05_bench_c2_launch_per_block.json", "bench_c4_walk2.json": "r05_bench_c4_walk2.json",
    "kernel_stats_c3.csv": "r05_kernel_stats_c3.csv", "kernel_stats_c2.csv": "r05_kernel_stats_c2.csv", "kernel_stats_c4.csv": "r05_kernel_stats_c4.csv",
    "kernel_stats_polymul.csv": "r05_kernel_stats_polymul.csv", "polymul.log": "r05_polymul.log",
    "single_call.json": "r05_single_call.json", "kernel_stats_single_call.csv": "r05_kernel_stats_single_call.csv", "trace_single_call.txt": "r05_trace_single_call.txt",
    "midsize.log": "r05_midsize.log", "traffic_c3.json": "r05_traffic_c3.json", "traffic_c2.json": "r05_traffic_c2.json", "traffic_c4.json": "r05_traffic_c4.json",
    "traffic_single_b16.json": "r05_traffic_single_b16.json", "host_path.log": "r05_host_path.log", "host_path_gpv.log": "r05_host_path_gpv.log",
    "host_async_stress_gpv.log": "r05_host_async_stress_gpv.log", "ring_fa.log": "r05_ring_fa.log", "probe_ldsdma_l2.log": "r05_probe_ldsdma_l2.log", "keygen.log": "r05_keygen.log",
    "keygen_phases.log": "r05_keygen_phases.log", "keygen_timeline_c3.txt": "r05_keygen_timeline_c3.txt", "keygen_timeline_c2.txt": "r05_keygen_timeline_c2.txt",
    "bench_c3_round4_stages.json": "r05_bench_c3_round4_stages.json", "bench_c2_combine_per_pair.json": "r05_bench_c2_combine_per_pair.json",
    "bench_c4_combine_per_pair.json": "r05_bench_c4_combine_per_pair.json",
}
for cfg in ("c3", "c2", "c4"):
    r = last_json(os.path.join(SRC, f"rocprof_{cfg}.log"), "metric")
    if r:
        with open(os.path.join(DST, f"r05_bench_under_rocprof_{cfg}.json"), "w") as fh:
            json.dump(r, fh)
for s, d in COPY.items():
    p = os.path.join(SRC, s)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(DST, d))
    else:
        print("missing:", s)

t = {}
c3 = last_json(os.path.join(SRC, "traffic_c3.json"), "hbm_bytes_per_launch")
if c3 and "hbm_bytes_per_launch" in c3:
    t["c3:B4096"] = {"hbm_bytes_per_launch": c3["hbm_bytes_per_launch"], "kernel": "k_trmm_f64_big", "kernel_source_sha256": sha(["psf_kernels.hpp"]),
                     "how": f"tools/pmc_traffic.sh c3 k_trmm_f64 (round 5): FETCH_SIZE {c3['FETCH_SIZE_KiB_avg']:.0f} KiB doubled + WRITE_SIZE {c3['WRITE_SIZE_KiB_avg']:.0f} KiB, "
                            "separate --pmc passes, averages over the launches of `bench.py --config c3 --steps 2 --warmup 1`",
                     "algorithmic_bytes_per_launch": 5840000000.0}
    try:
        with open(os.path.join(DST, "trmm_traffic.json")) as fh:
            old = json.load(fh)
        if "c3:structured:B4096" in old and old["c3:structured:B4096"].get("kernel_source_sha256") == sha(["psf_kernels.hpp"]):
            t["c3:structured:B4096"] = old["c3:structured:B4096"]
    except (OSError, ValueError):
        pass
    with open(os.path.join(DST, "trmm_traffic.json"), "w") as fh:
        json.dump(t, fh, indent=1)
npt = {}
for cfg, B, alg in (("c2", 1024, 515424256), ("c4", 4096, 279969792)):
    r = last_json(os.path.join(SRC, f"traffic_{cfg}.json"), "hbm_bytes_per_call")
    if r and "hbm_bytes_per_call" in r:
        npt[f"{cfg}:B{B}"] = {"hbm_bytes_per_call": r["hbm_bytes_per_call"], "kernels": "every k_np_* launch of one samp_p call",
                              "kernel_source_sha256": sha(["psf_np_kernels.hpp"]),
                              "how": f"tools/pmc_np.sh {cfg} (round 5): FETCH_SIZE {r['FETCH_SIZE_KiB']:.0f} KiB doubled + WRITE_SIZE {r['WRITE_SIZE_KiB']:.0f} KiB over the last call of tools/bin/np_harness",
                              "algorithmic_bytes_per_call": alg, "per_kernel_KiB": r.get("per_kernel_KiB"), "launches": r.get("launches")}
if npt:
    with open(os.path.join(DST, "np_traffic.json"), "w") as fh:
        json.dump(npt, fh, indent=1)
print("traffic records:", list(t), list(npt))
