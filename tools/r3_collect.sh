#!/bin/bash
# Round-3 evidence pass on the GPU box -> gpurun_out/r3_collect_<tag>/ (copy what should be judged into profiles/).
# usage: tools/r3_collect.sh [tag]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-final}
OUT=$R/gpurun_out/r3_collect_$TAG; mkdir -p $OUT
cd $R
python bench.py --steps 20 --warmup 5 > $OUT/r03_bench_cfg3.json 2> $OUT/bench_cfg3.err
python tools/input_pipeline_bench.py --out $OUT/r03_input_pipeline.json > /dev/null 2>&1
python tools/ctc_beam_bench.py --out $OUT/r03_ctc_beam.json > /dev/null 2> $OUT/ctc_beam.err
python tools/gemm_shapes.py cfg3 2>&1 | grep -v amdgpu.ids > $OUT/r03_gemm_census_cfg3.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- \
    python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-exact-check > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/r03_cfg3_kernel_stats.csv
TR=$(find $OUT/stats -name "*kernel_trace.csv" | head -1)
python $R/tools/step_timeline.py $TR 300 4 > $OUT/r03_cfg3_step_timeline.log 2>&1
rm -rf $OUT/stats
cd $R
tools/pmc_hbm.sh cfg3 r3_pmc_hbm_cfg3 > $OUT/pmc_cfg3.log 2>&1
cp $R/gpurun_out/r3_pmc_hbm_cfg3/hbm_traffic_cfg3.json $OUT/r03_hbm_traffic_cfg3.json
rm -rf $R/gpurun_out/r3_pmc_hbm_cfg3/FETCH_SIZE $R/gpurun_out/r3_pmc_hbm_cfg3/WRITE_SIZE
python tools/rec_timeline.py 800 32 4096 1024 2>&1 | grep -v amdgpu.ids > $OUT/r03_rec_timeline_h1024.log
head -c 500 $OUT/r03_bench_cfg3.json; echo
