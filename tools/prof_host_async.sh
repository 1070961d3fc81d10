#!/bin/bash
# kernel + memcpy timeline of overlapped asynchronous host-pointer calls (C3): where is the GPU idle?
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace -d $O/prof_hostasync -o t --output-format csv -- python3 $R/tools/host_overlap_kernels.py > $O/r4_hostasync.log 2>&1
cd $R
python3 - $O/prof_hostasync <<'PY'
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0].replace("void ", "").replace("psf::", "")[:40]))
for f in glob.glob(d + "/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Name", "?")) + " " + str(r.get("Size", ""))))
ev.sort()
# the last ~3 calls: from the third-to-last k_normals on
starts = [i for i, e in enumerate(ev) if "k_normals_wave" in e[2]]
i0 = starts[-4]
t0 = ev[i0][0]
import collections
agg = collections.OrderedDict()
for s, e, n in ev[i0:]:
    # merge runs of the same name
    key = n
    if agg and list(agg.keys())[-1][0] == key:
        k = list(agg.keys())[-1]
        agg[k] = (agg[k][0], max(agg[k][1], e), agg[k][2] + 1, agg[k][3] + (e - s))
    else:
        agg[(key, s)] = (s, e, 1, e - s)
for (n, _), (s, e, cnt, busy) in agg.items():
    print(f"{(s - t0) / 1e6:9.2f} .. {(e - t0) / 1e6:9.2f} ms  x{cnt:<4d} busy {busy / 1e6:7.2f} ms  {n}")
PY
rm -rf $O/prof_hostasync
