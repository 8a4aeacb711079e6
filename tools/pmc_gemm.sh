#!/bin/bash
# PMC passes over one GEMM shape; usage: tools/pmc_gemm.sh NT 16000 4096 2048 <outdir>
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/$5; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python $R/tools/gemm_one.py $1 $2 $3 $4 > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/p2 -- python $R/tools/gemm_one.py $1 $2 $3 $4 > $OUT/p2.log 2>&1
find $OUT -name "*counter_collection.csv" | while read f; do echo "== $f"; python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in agg.items():
    if 'gemm' in k:
        print(k, {c: '%.4g' % v for c, v in d.items()})
PY
done
