#!/bin/bash
# HBM traffic per kernel of any python tool: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (MI355X_MICROARCH.md: FETCH_SIZE in KB,
# x 2 on gfx950; WRITE_SIZE in KB), summed per kernel name with the call counts.  Usage: tools/pmc_traffic.sh <tag> <script.py> [args]
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/traffic_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 $@ > $OUT/f.log 2> $OUT/f.err || { tail -20 $OUT/f.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 $@ > $OUT/w.log 2> $OUT/w.err || { tail -20 $OUT/w.err; exit 1; }
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
tot = collections.defaultdict(lambda: [0, 0.0, 0.0])
for d, col in (("f", 1), ("w", 2)):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:100]
        if d == "f": tot[k][0] += 1
        tot[k][col] += float(r["Counter_Value"])
print("# per kernel: calls, FETCH_SIZE KB per call, WRITE_SIZE KB per call, HBM bytes per call = 2 * FETCH_SIZE + WRITE_SIZE (gfx950)")
for k, (n, fk, wk) in sorted(tot.items(), key=lambda kv: -(2 * kv[1][1] + kv[1][2])):
    if n == 0: continue
    print("%-100s calls=%d fetch_kb=%.1f write_kb=%.1f hbm_bytes=%.0f" % (k, n, fk / n, wk / n, (2 * fk + wk) * 1024 / n))
PY
cat $OUT/summary.txt; grep -v amdgpu $OUT/f.log | tail -5
rm -rf $OUT/f $OUT/w
