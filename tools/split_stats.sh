#!/bin/bash
# kernel-level durations of the split path on the cfg3 layer-1 shapes
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/split_stats_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -- python $R/tools/gemm_split_bench.py --only="cfg3 L1" > $OUT/run.log 2>&1
f=$(find $OUT/p -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("%-60s calls %4s avg %10.1f us min %10.1f max %10.1f" % (r["Name"].replace("(anonymous namespace)::", "")[:60],
          r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
rm -rf $OUT/p
