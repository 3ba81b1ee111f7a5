#!/bin/bash
# usage (ON THE GPU BOX): tools/trace_small.sh <tag>  -- kernel trace (start / end of every dispatch) of the one-frame call and of the 128-frame step
TAG=${1:?tag}; R=$(pwd); OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b1 -o b1 -- python3 $R/tools/profile_batch1.py > $OUT/b1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b128 -o b128 -- python3 $R/tools/profile_batch128.py > $OUT/b128.log 2>&1
cd $R
python3 - <<PY
import csv, glob
for tag in ("b1", "b128"):
    f = glob.glob("$OUT/%s/*kernel_trace.csv" % tag)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-40:]
    t0 = int(rows[0]["Start_Timestamp"])
    print("==", tag)
    prev_end = None
    for r in rows:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print("%-60s start %9.1f us  dur %8.1f us  gap %6.1f" % (r["Kernel_Name"].split("(")[0][-60:], s / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end is not None else 0.0))
        prev_end = e
PY
