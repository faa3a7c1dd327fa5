#!/bin/bash
# Kernel trace + SQ counters (matrix-core busy, waits) + HBM bytes of the p = 15 stiffness kernels (tools/p15_run.py <tuning 4 value>)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=${1:-2}
OUT=$R/gpurun_out/pmc_p15_$V
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/p15_run.py $V > $OUT/trace.log 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p1 -- python3 tools/p15_run.py $V > $OUT/p1.log 2> $OUT/p1.err || { tail -20 $OUT/p1.err; exit 1; }
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES SQ_WAVES --output-format csv -d $OUT/p2 -- python3 tools/p15_run.py $V > $OUT/p2.log 2> $OUT/p2.err || { tail -20 $OUT/p2.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p3 -- python3 tools/p15_run.py $V > $OUT/p3.log 2> $OUT/p3.err || { tail -20 $OUT/p3.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p4 -- python3 tools/p15_run.py $V > $OUT/p4.log 2> $OUT/p4.err || { tail -20 $OUT/p4.err; exit 1; }
python3 - <<PY
import csv, glob, collections
print(open("$OUT/trace.log").read().strip())
ks = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(ks[0])))[:4]:
    print("%-70s calls=%s avg_ns=%s min_ns=%s max_ns=%s pct=%s" % (r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]))
for d in ("p1", "p2", "p3", "p4"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "stiffness" not in k: continue
        for c, vals in sorted(v.items()):
            print("   %-26s avg %.6g  (n=%d)  %s" % (c, sum(vals)/len(vals), len(vals), k[:40]))
PY
