set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
python bench.py > gpurun_out/bench_r04d.json 2> gpurun_out/bench_r04d.err
tail -1 gpurun_out/bench_r04d.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value', d['value'], 'frac', d['roofline']['frac'], 'kernel_ms', d['roofline']['kernel_avg_ms'])
for k,v in d['secondary'].items():
    if isinstance(v,dict):
        print(k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('apply_aij_GDoF_per_s','apply_aij_ms','roofline_frac_hbm','GDoF_per_s','ms','cheby_5_iterations_ms')})
"
mkdir -p gpurun_out/prof_r04_final
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04_final/trace -- python3 bench.py --no-check > gpurun_out/prof_r04_final/bench.json 2> gpurun_out/prof_r04_final/bench.err
python3 - <<PY > gpurun_out/prof_r04_final/summary.txt
import csv, glob
f = glob.glob("gpurun_out/prof_r04_final/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:60]:
    print("%-110s calls=%s avg_ns=%s pct=%s" % (r["Name"][:110], r["Calls"], r["AverageNs"], r["Percentage"]))
PY
rm -rf gpurun_out/prof_r04_final/trace
head -5 gpurun_out/prof_r04_final/summary.txt
