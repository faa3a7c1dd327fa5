#!/bin/bash
# SQ counter passes of the whole-operator kernel (tools/time_whole.py <level> <deg> [key11]). Usage: tools/pmc_whole.sh <tag> <level> <deg> [key11]
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcw_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p1 -- python3 tools/time_whole.py $@ > $OUT/b1.log 2> $OUT/p1.err || { tail -20 $OUT/p1.err; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES --output-format csv -d $OUT/p2 -- python3 tools/time_whole.py $@ > $OUT/b2.log 2> $OUT/p2.err || { tail -20 $OUT/p2.err; exit 1; }
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -- python3 tools/time_whole.py $@ > $OUT/b3.log 2> $OUT/p3.err || { tail -5 $OUT/p3.err; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 tools/time_whole.py $@ > $OUT/b4.log 2> $OUT/kt.err || { tail -5 $OUT/kt.err; }
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
for d in ("p1","p2","p3"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True)
    if not f: print("no csv for", d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"][:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if not any(t in k for t in ("faces_direct", "operator_mw", "stiffness_wave")): continue
        print(d, k)
        for c, vals in sorted(v.items()):
            print("   %-24s avg %.5g  (n=%d)" % (c, sum(vals)/len(vals), len(vals)))
for f in glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("kernel-trace:", r["Name"][:90], "calls", r["Calls"], "avg ns", r["AverageNs"])
PY
cat $OUT/summary.txt
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/kt
