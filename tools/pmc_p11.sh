#!/bin/bash
# SQ counters of the multi-wave stiffness kernel (tools/config3.py <level> <deg>)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_p11
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/p1 -- python3 tools/config3.py $@ > $OUT/b1.log 2> $OUT/p1.err || { tail -20 $OUT/p1.err; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES --output-format csv -d $OUT/p2 -- python3 tools/config3.py $@ > $OUT/b2.log 2> $OUT/p2.err || { tail -20 $OUT/p2.err; exit 1; }
python3 - <<PY
import csv, glob, collections
for d in ("p1","p2"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "stiffness" not in k: continue
        print(d, k)
        for c, vals in sorted(v.items()):
            print("   %-24s avg %.5g  (n=%d)" % (c, sum(vals)/len(vals), len(vals)))
PY
