#!/bin/bash
# kernel trace (start/end timestamps) of one bench configuration -> gpurun_out/<tag>_trace_<config>.csv (kernel name, stream/queue, start, end)
export TMPDIR=/tmp
cfg=$1; tag=$2; shift 2
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace -d $O/prof_${tag}_$cfg -o t --output-format csv -- python3 $R/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline "$@" > $O/${tag}_trace_$cfg.log 2>&1
f=$(ls $O/prof_${tag}_$cfg/*kernel_trace.csv 2>/dev/null | head -1)
python3 - "$f" $O/${tag}_trace_$cfg.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last samp_p call: from the last k_np_solve on
last = max(i for i, r in enumerate(rows) if "k_np_solve" in r["Kernel_Name"])
t0 = int(rows[last]["Start_Timestamp"])
with open(sys.argv[2], "w") as fh:
    fh.write("kernel,queue,start_us,end_us,dur_us\n")
    for r in rows[last:]:
        nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("psf::", "")
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        fh.write(f"{nm},{r.get('Queue_Id','')},{s/1e3:.1f},{e/1e3:.1f},{(e-s)/1e3:.1f}\n")
PY
rm -rf $O/prof_${tag}_$cfg
head -40 $O/${tag}_trace_$cfg.csv
