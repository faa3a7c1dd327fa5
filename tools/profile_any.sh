#!/bin/bash
# kernel trace of any python tool on the GPU box; usage: tools/profile_any.sh <tag> <script.py> [args]; summary -> gpurun_out/prof_<tag>/summary.txt
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $@ > $OUT/run.log 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
python3 - <<PY > $OUT/summary.txt
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-100s calls=%s avg_ns=%s pct=%s" % (r["Name"][:100], r["Calls"], r["AverageNs"], r["Percentage"]))
PY
cat $OUT/summary.txt; cat $OUT/run.log | grep -v amdgpu.ids
rm -rf $OUT/trace
