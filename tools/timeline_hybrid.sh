#!/bin/bash
# kernel timeline (start / end per dispatch) of the last hybrid apply of tools/time_hybrid.py <mesh>: do the clean buckets' kernels overlap?
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/timeline_$1
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/time_hybrid.py $1 > $OUT/run.log 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
python3 - <<PY > $OUT/timeline.txt
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 14 dispatches that contain a direct / operator_mw kernel: one hybrid apply
idx = [i for i, r in enumerate(rows) if "faces_direct_kernel" in r["Kernel_Name"] or "operator_mw_kernel" in r["Kernel_Name"]]
last = idx[-1]
first = last
while first > 0 and int(rows[last]["End_Timestamp"]) - int(rows[first - 1]["Start_Timestamp"]) < 400000: first -= 1
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:last + 1]:
    print("%8.1f %8.1f us  queue %-4s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:90]))
PY
cat $OUT/timeline.txt
rm -rf $OUT/trace
