#!/bin/bash
# kernel trace of the Schwarz iterate (tools/time_schwarz.py); usage: tools/profile_schwarz.sh <tag> [time_schwarz args]
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_sz_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/time_schwarz.py $@ > $OUT/run.log 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
python3 - <<PY > $OUT/summary.txt
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-90s calls=%s avg_ns=%s pct=%s" % (r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"]))
PY
cat $OUT/summary.txt; grep "iterate\|apply_over" $OUT/run.log
