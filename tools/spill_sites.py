#!/usr/bin/env python3
"""Where are a kernel's register spills?  Reads hipcc's device assembly (hipcc ... --cuda-device-only -S) and reports, for one kernel, how many spill stores / reloads
("Folded Spill" / "Folded Reload") sit inside the loops that contain a given instruction (the sampler's step loop: v_exp_f32; the updater's K loop: v_mfma_f64).
usage: tools/spill_sites.py <asm file> <mangled-name prefix>"""
import re
import sys

path, prefix = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and ":" in l and "@" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
stores = [i for i, l in enumerate(body) if "Folded Spill" in l]
reloads = [i for i, l in enumerate(body) if "Folded Reload" in l]
print(f"{prefix[:40]}...: {len(body)} lines, {len(loops)} loops, {len(stores)} spill stores, {len(reloads)} reloads")


def innermost(pos):
    c = [(a, b) for a, b in loops if a <= pos <= b]
    return min(c, key=lambda ab: ab[1] - ab[0]) if c else None


for what, pat in (("the sampler's step loop (v_exp_f32)", "v_exp_f32"), ("the updater's K loop (v_mfma_f64)", "v_mfma_f64")):
    sites = [i for i, l in enumerate(body) if pat in l]
    inner = sorted(set(innermost(p) for p in sites if innermost(p)))
    ins = [p for p in stores if any(a <= p <= b for a, b in inner)]
    inr = [p for p in reloads if any(a <= p <= b for a, b in inner)]
    print(f"  {what}: {len(inner)} innermost loop(s) of {[b - a for a, b in inner]} lines; spill stores inside: {len(ins)}, reloads inside: {len(inr)}")
depth = lambda p: len([1 for a, b in loops if a <= p <= b])
print("  loop depth of the spill stores:", {d: [depth(p) for p in stores].count(d) for d in sorted(set(depth(p) for p in stores))},
      "of the reloads:", {d: [depth(p) for p in reloads].count(d) for d in sorted(set(depth(p) for p in reloads))})
