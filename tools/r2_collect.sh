#!/bin/bash
# Round-2 evidence pass on the GPU box -> gpurun_out/r2_collect/ (copy what should be judged into profiles/).
# usage: tools/r2_collect.sh [tag]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-final}
OUT=$R/gpurun_out/r2_collect_$TAG; mkdir -p $OUT
cd $R
python bench.py --steps 20 --warmup 5 > $OUT/r02_bench_cfg3.json 2> $OUT/bench_cfg3.err
python bench.py --workload cfg2 --steps 40 --warmup 5 > $OUT/r02_bench_cfg2.json 2> $OUT/bench_cfg2.err
python tools/decode_bench.py --cpu-baseline 2> $OUT/decode.err | grep '^{' > $OUT/r02_decode_cfg5.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- \
    python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-exact-check > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/r02_cfg3_kernel_stats.csv
TR=$(find $OUT/stats -name "*kernel_trace.csv" | head -1)
python $R/tools/step_timeline.py $TR 300 4 > $OUT/r02_cfg3_step_timeline.log 2>&1
python $R/tools/step_timeline.py $TR 8 4 | head -400 > $OUT/r02_cfg3_step_timeline_fine_head.log 2>&1
rm -rf $OUT/stats
cd $R
tools/pmc_hbm.sh cfg3 r2_pmc_hbm_cfg3 > $OUT/pmc_cfg3.log 2>&1
cp $R/gpurun_out/r2_pmc_hbm_cfg3/hbm_traffic_cfg3.json $OUT/r02_hbm_traffic_cfg3.json
rm -rf $R/gpurun_out/r2_pmc_hbm_cfg3/FETCH_SIZE $R/gpurun_out/r2_pmc_hbm_cfg3/WRITE_SIZE
# SQ counters of the recurrence kernels at the cfg3 layer-1 shape (T=800, B=32, Din=4096, H=1024)
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python $R/tools/rec_timeline.py 800 32 4096 1024 > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/p2 -- python $R/tools/rec_timeline.py 800 32 4096 1024 > $OUT/p2.log 2>&1
find $OUT/p1 $OUT/p2 -name "*counter_collection.csv" | sort | while read f; do python3 - "$f" <<'PY'
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
done > $OUT/r02_rec_pmc_sq.log 2>&1
rm -rf $OUT/p1 $OUT/p2
cd $R
python tools/rec_timeline.py 800 32 4096 1024 2>&1 | grep -v amdgpu.ids > $OUT/r02_rec_timeline_h1024.log
python tools/rec_timeline.py 1000 32 80 512 2>&1 | grep -v amdgpu.ids > $OUT/r02_rec_timeline_h512.log
# split GEMM: per-shape rates + accuracy vs float64 beside the exact-f32 kernel, kernel durations and counters
python tools/gemm_split_bench.py --acc 2>&1 | grep -v amdgpu.ids > $OUT/r02_gemm_split_bench.log
tools/pmc_split.sh NT 25600 8192 4096 r2_pmc_split > /dev/null 2>&1
cp $R/gpurun_out/r2_pmc_split/summary.txt $OUT/r02_gemm_split_pmc.log
python tools/gru_bench.py 2>&1 | grep '^{' > $OUT/r02_gru_layer.json
head -c 600 $OUT/r02_bench_cfg3.json; echo; cat $OUT/r02_rec_pmc_sq.log
