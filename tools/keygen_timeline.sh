#!/bin/bash
# kernel timeline of the LAST trap_gen of tools/keygen_time.py <config>: busy time, idle gaps, per-kernel totals -> gpurun_out/<tag>_keygen_timeline_<config>.txt
export TMPDIR=/tmp
cfg=$1; tag=$2
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace -d $O/prof_${tag}_kg_$cfg -o t --output-format csv -- python3 $R/tools/keygen_time.py $cfg > $O/${tag}_keygen_trace_$cfg.log 2>&1
f=$(ls $O/prof_${tag}_kg_$cfg/*kernel_trace.csv 2>/dev/null | head -1)
python3 - "$f" > $O/${tag}_keygen_timeline_$cfg.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r): return r["Kernel_Name"].split("(")[0].replace("void ", "").replace("psf::", "")
# segments separated by idle gaps > 20 ms (create / between reps); keep the last one
segs, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) > 20_000_000: segs.append(cur); cur = []
    cur.append(b)
segs.append(cur)
seg = max(segs[-2:], key=len) if len(segs) > 1 else segs[-1]
t0 = int(seg[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in seg)
print(f"segment: {len(seg)} launches, {(t1 - t0) / 1e6:.2f} ms from first start to last end")
# union busy
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
busy, ce = 0, t0
gaps = []
for s, e in ev:
    if s > ce: gaps.append((s - ce, ce - t0, s - t0)); busy += e - s; ce = e
    elif e > ce: busy += e - ce; ce = e
print(f"busy (union of launches) {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms in {len(gaps)} gaps")
tot = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = nm(r); tot[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); tot[k][1] += 1
print("per kernel (sum of durations, launches):")
for k, (d, c) in sorted(tot.items(), key=lambda x: -x[1][0])[:25]: print(f"  {d / 1e6:9.3f} ms {c:6d}  {k}")
print("largest idle gaps (ms, at ms, before kernel):")
byend = sorted(seg, key=lambda r: int(r["Start_Timestamp"]))
for g, a, b in sorted(gaps, reverse=True)[:15]:
    nxt = next(r for r in byend if int(r["Start_Timestamp"]) - t0 == b)
    print(f"  {g / 1e6:8.3f} at {a / 1e6:8.2f}  -> {nm(nxt)}")
hist = collections.Counter(min(int(g / 1e3) // 10 * 10, 200) for g, _, _ in gaps)
print("gap histogram (us bucket: count, total ms):")
for b in sorted(hist): print(f"  {b:4d}+: {hist[b]:5d}  {sum(g for g, _, _ in gaps if min(int(g / 1e3) // 10 * 10, 200) == b) / 1e6:.2f}")
PY
rm -rf $O/prof_${tag}_kg_$cfg
cat $O/${tag}_keygen_timeline_$cfg.txt
