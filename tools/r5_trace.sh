#!/bin/bash
# rocprofv3 kernel trace of a short cfg3 bench run: per-kernel stats + the in-situ split-pass table
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r5}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-exact-check > $OUT/trace_run.log 2>&1
f=$(find $OUT/p -name "*kernel_stats.csv" | head -1)
t=$(find $OUT/p -name "*kernel_trace.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
python3 $R/tools/split_insitu.py "$t" 5 > $OUT/split_insitu.log 2>&1
python3 $R/tools/trace_summary.py "$t" 40 > $OUT/trace_summary.log 2>&1
rm -rf $OUT/p
