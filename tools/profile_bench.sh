#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace + separate PMC passes of the SAME bench command.
# Usage: tools/profile_bench.sh <tag> [bench args...]; outputs under gpurun_out/prof_<tag>/
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-check --no-secondary $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err || { tail -20 $OUT/fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err || { tail -20 $OUT/write.err; exit 1; }
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
