#!/bin/bash
# kernel durations + PMC passes over one GEMM shape on the bf16x6 split path
# usage: tools/pmc_split.sh NT 25600 8192 4096 <outdir>
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
OUT=$R/gpurun_out/$5; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ASRK_GEMM_SPLIT=2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p0 -- python $R/tools/gemm_one.py $1 $2 $3 $4 5 > $OUT/p0.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python $R/tools/gemm_one.py $1 $2 $3 $4 > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/p2 -- python $R/tools/gemm_one.py $1 $2 $3 $4 > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $OUT/p3 -- python $R/tools/gemm_one.py $1 $2 $3 $4 > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p4 -- python $R/tools/gemm_one.py $1 $2 $3 $4 > $OUT/p4.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/p5 -- python $R/tools/gemm_one.py $1 $2 $3 $4 > $OUT/p5.log 2>&1
{
echo "shape $1 $2 $3 $4"
f=$(find $OUT/p0 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-8 "$f" | head -8
find $OUT -name "*counter_collection.csv" | sort | while read f; do python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in rows:
    k = r['Kernel_Name'][:40]
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
for k, d in agg.items():
    if 'gemm' in k or 'split' in k:
        print(k, 'launches', len(n[k]), {c: '%.5g' % (v / len(n[k])) for c, v in d.items()})
PY
done
} > $OUT/summary.txt 2>&1
rm -rf $OUT/p0 $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5
cat $OUT/summary.txt
