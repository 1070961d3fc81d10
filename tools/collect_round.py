#!/usr/bin/env python3
"""Copies a round's measurement pass (gpurun_out/<tag>_final/, written by tools/final_round.sh on the GPU box) into profiles/ under per-round names and refreshes the
hash-tied traffic records (profiles/trmm_traffic.json, profiles/np_traffic.json) that bench.py reads for `roofline.traffic`: a record is only returned while the kernel
source it was measured on is unchanged (sha256 of the kernel headers beside it).  usage: python tools/collect_round.py r06"""
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
SRC = os.path.join(ROOT, "gpurun_out", f"{tag}_final")
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


for name in sorted(os.listdir(SRC)):
    p = os.path.join(SRC, name)
    if name.startswith("bench_under_rocprof") or name.endswith(".err") or name == "bench_default.log" or os.path.getsize(p) == 0:
        continue
    if name.startswith("rocprof_") and name.endswith(".log"):           # the bench line printed under rocprofv3 (stdout and the tool's log share the file)
        r = last_json(p, "metric")
        if r:
            with open(os.path.join(DST, f"{tag}_bench_under_rocprof_{name[8:-4]}.json"), "w") as fh:
                json.dump(r, fh)
        continue
    if name.startswith("bench_") and name.endswith(".log"):
        continue
    shutil.copy(p, os.path.join(DST, f"{tag}_{name}"))

c3 = last_json(os.path.join(SRC, "traffic_c3.json"), "hbm_bytes_per_launch")
if c3:
    m, B = 30801, 4096
    rec = {"c3:B4096": {"hbm_bytes_per_launch": c3["hbm_bytes_per_launch"], "kernel": "k_trmm_f64_big", "kernel_source_sha256": sha(["psf_kernels.hpp"]),
                        "how": f"tools/pmc_traffic.sh c3 k_trmm_f64_big ({tag}): FETCH_SIZE {c3['FETCH_SIZE_KiB_avg']:.0f} KiB doubled + WRITE_SIZE {c3['WRITE_SIZE_KiB_avg']:.0f} KiB, separate --pmc "
                               "passes, averages over the launches of `bench.py --config c3 --steps 2 --warmup 1`",
                        "algorithmic_bytes_per_launch": float(m * (m + 1) // 2 * 8 + 2 * m * B * 8)}}
    with open(os.path.join(DST, "trmm_traffic.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
np_rec = {}
# algorithmic bytes of one call (the Gram-Schmidt data in f64 and the basis once per batch, targets in, preimages out) -- the figures of rounds 3-5, kept
for cfg, key, alg in (("c2", "c2:B1024", 515424256), ("c4", "c4:B4096", 279969792)):
    r = last_json(os.path.join(SRC, f"traffic_{cfg}.json"), "hbm_bytes_per_call")
    if not r:
        continue
    np_rec[key] = {"hbm_bytes_per_call": r["hbm_bytes_per_call"], "kernels": "every k_np_* launch of one samp_p call", "kernel_source_sha256": sha(["psf_np_kernels.hpp"]),
                   "how": f"tools/pmc_np.sh {cfg} ({tag}): FETCH_SIZE {r['FETCH_SIZE_KiB']:.0f} KiB doubled + WRITE_SIZE {r['WRITE_SIZE_KiB']:.0f} KiB over the last call of tools/bin/np_harness",
                   "algorithmic_bytes_per_call": alg, "per_kernel_KiB": r["per_kernel_KiB"], "launches": r["launches"]}
if np_rec:
    with open(os.path.join(DST, "np_traffic.json"), "w") as fh:
        json.dump(np_rec, fh, indent=1)
print("collected", tag, "->", DST)
