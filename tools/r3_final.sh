#!/bin/bash
# Round-3 closing evidence pass on the GPU box: HBM traffic (stamped) FIRST, so that the bench line that follows
# carries roofline.traffic measured on exactly these kernel sources.  -> gpurun_out/r3_final/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3_final; mkdir -p $OUT
cd $R
tools/pmc_hbm.sh cfg3 r3_pmc_hbm_cfg3 > $OUT/pmc_cfg3.log 2>&1
cp $R/gpurun_out/r3_pmc_hbm_cfg3/hbm_traffic_cfg3.json $OUT/r03_hbm_traffic_cfg3.json
cp $OUT/r03_hbm_traffic_cfg3.json $R/profiles/r03_hbm_traffic_cfg3.json
rm -rf $R/gpurun_out/r3_pmc_hbm_cfg3/FETCH_SIZE $R/gpurun_out/r3_pmc_hbm_cfg3/WRITE_SIZE
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- \
    python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-exact-check > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/r03_cfg3_kernel_stats.csv
TR=$(find $OUT/stats -name "*kernel_trace.csv" | head -1)
python $R/tools/step_timeline.py $TR 300 4 > $OUT/r03_cfg3_step_timeline.log 2>&1
rm -rf $OUT/stats
cd $R
python bench.py --steps 20 --warmup 5 > $OUT/r03_bench_cfg3.json 2> $OUT/bench_cfg3.err
python tools/gemm_shapes.py cfg3 2>&1 | grep -v amdgpu.ids > $OUT/r03_gemm_census_cfg3.log
head -c 600 $OUT/r03_bench_cfg3.json; echo; tail -3 $OUT/pmc_cfg3.log
