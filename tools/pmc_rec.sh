#!/bin/bash
# SQ counters of the persistent LSTM recurrence kernels (cfg2 layer-0 shape, one forward + one
# backward launch through tools/rec_timeline.py); usage: tools/pmc_rec.sh <outdir under gpurun_out> [T B D H]
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/$1; mkdir -p $OUT; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python $R/tools/rec_timeline.py "$@" > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/p2 -- python $R/tools/rec_timeline.py "$@" > $OUT/p2.log 2>&1
find $OUT -name "*counter_collection.csv" | sort | while read f; do python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:44]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in agg.items():
    if 'lstm_rec' in k:
        print(k, {c: '%.4g' % v for c, v in d.items()})
PY
done
