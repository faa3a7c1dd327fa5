#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace + separate FETCH_SIZE / WRITE_SIZE passes of the whole-operator apply (tools/time_whole.py).
# Usage: tools/profile_whole.sh <tag> <level> <deg> [key11 reps]; summary -> gpurun_out/prof_<tag>/summary.txt
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/time_whole.py $@ > $OUT/run_trace.log 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/time_whole.py $@ > $OUT/run_fetch.log 2> $OUT/fetch.err || { tail -20 $OUT/fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/time_whole.py $@ > $OUT/run_write.log 2> $OUT/write.err || { tail -20 $OUT/write.err; exit 1; }
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt
grep level $OUT/run_trace.log >> $OUT/summary.txt
cat $OUT/summary.txt
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write
