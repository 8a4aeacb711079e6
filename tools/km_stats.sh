#!/bin/bash
# per-kernel totals of the GEMM / split / recurrence kernels over a short cfg3 bench run (A/B of ASRK_KMAJOR)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 1 0; do
  rm -rf /tmp/kms
  ASRK_KMAJOR=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kms -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-exact-check > /tmp/kms.log 2>&1
  f=$(find /tmp/kms -name "*kernel_stats.csv" | head -1)
  echo "ASRK_KMAJOR=$v"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    if any(k in n for k in ('gemm_km', 'gemm_bf16x6', 'split_panel', 'lstm_rec')):
        print("  %-46s calls %4d  avg %9.1f us  per step %8.3f ms" % (n[:46], int(r['Calls']), float(r['AverageNs']) / 1e3,
                                                                      float(r['TotalDurationNs']) / 5e6))
PY
done
