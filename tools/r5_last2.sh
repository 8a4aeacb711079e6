#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5_last; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-exact-check > $OUT/tr.log 2>&1
TR=$(find $OUT/tr -name "*kernel_trace.csv" | head -1)
python $R/tools/split_insitu.py $TR 10 > $OUT/r05_split_insitu_cfg3.log 2>&1
rm -rf $OUT/tr; tail -3 $OUT/r05_split_insitu_cfg3.log
