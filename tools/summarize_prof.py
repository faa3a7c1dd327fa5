"""Summarises a tools/profile_bench.sh output directory: per-kernel time stats and PMC HBM bytes."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    r = glob.glob(os.path.join(out, pattern), recursive=True)
    return r[0] if r else None


ks = find("trace/**/*kernel_stats.csv")
if ks:
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    rows = list(csv.DictReader(open(ks)))
    for r in rows[:12]:
        print("%-90s calls=%s avg_ns=%s min_ns=%s max_ns=%s pct=%s" % (r.get("Name", "")[:90], r.get("Calls"), r.get("AverageNs"), r.get("MinNs"), r.get("MaxNs"), r.get("Percentage")))
kt = find("trace/**/*kernel_trace.csv")
if kt:
    d = defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(kt)):
        n = r["Kernel_Name"]
        d[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[n] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Workgroup_Size"), r.get("Grid_Size"))
    print("== per-kernel durations from the trace (ns) ==")
    for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:8]:
        v2 = sorted(v)
        print("%-80s n=%d avg=%.0f median=%d min=%d  vgpr/sgpr/lds/wg/grid=%s" % (n[:80], len(v), sum(v) / len(v), v2[len(v2) // 2], v2[0], meta[n]))
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    f = find("pmc_%s/**/*counter_collection.csv" % ("fetch" if name == "FETCH_SIZE" else "write"))
    if not f:
        continue
    d = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    print("== %s per dispatch (raw counter value, KB per the rocprof convention) ==" % name)
    for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:6]:
        v2 = sorted(v)
        print("%-80s n=%d avg=%.1f median=%.1f" % (n[:80], len(v), sum(v) / len(v), v2[len(v2) // 2]))
