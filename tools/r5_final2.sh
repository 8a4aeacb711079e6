#!/bin/bash
# Round-5 closing evidence, trimmed re-run after the dG^T panel change (most important files first).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5_final; mkdir -p $OUT
cd $R
tools/pmc_hbm.sh cfg3 r5_pmc_hbm_cfg3 > $OUT/pmc_cfg3.log 2>&1
cp $R/gpurun_out/r5_pmc_hbm_cfg3/hbm_traffic_cfg3.json $OUT/r05_hbm_traffic_cfg3.json
cp $OUT/r05_hbm_traffic_cfg3.json $R/profiles/r05_hbm_traffic_cfg3.json
rm -rf $R/gpurun_out/r5_pmc_hbm_cfg3/FETCH_SIZE $R/gpurun_out/r5_pmc_hbm_cfg3/WRITE_SIZE
python bench.py --steps 20 --warmup 5 > $OUT/r05_bench_cfg3.json 2> $OUT/bench_cfg3.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg3 -- \
    python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exact-check > $OUT/stats_cfg3.log 2>&1
cp $(find $OUT/stats_cfg3 -name "*kernel_stats.csv" | head -1) $OUT/r05_cfg3_kernel_stats.csv
TR=$(find $OUT/stats_cfg3 -name "*kernel_trace.csv" | head -1)
python $R/tools/step_timeline.py $TR 300 4 > $OUT/r05_cfg3_step_timeline.log 2>&1
python $R/tools/split_insitu.py $TR 12 > $OUT/r05_split_insitu_cfg3.log 2>&1
rm -rf $OUT/stats_cfg3
cd $R
tools/pmc_hbm.sh shipped r5_pmc_hbm_shipped > $OUT/pmc_shipped.log 2>&1
cp $R/gpurun_out/r5_pmc_hbm_shipped/hbm_traffic_shipped.json $OUT/r05_hbm_traffic_shipped.json
cp $OUT/r05_hbm_traffic_shipped.json $R/profiles/r05_hbm_traffic_shipped.json
rm -rf $R/gpurun_out/r5_pmc_hbm_shipped/FETCH_SIZE $R/gpurun_out/r5_pmc_hbm_shipped/WRITE_SIZE
python bench.py --workload shipped --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r05_bench_shipped.json 2> $OUT/bench_shipped.err
ASRK_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-check 2> $OUT/bench_dist.err | grep '^{' > $OUT/r05_bench_cfg3_rccl_world1.json
python bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r05_bench_cfg2.json 2> $OUT/bench_cfg2.err
python bench.py --workload cnn --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r05_bench_cnn.json 2> $OUT/bench_cnn.err
python tools/rec_timeline.py 800 32 4096 1024 2>&1 | grep -v amdgpu.ids > $OUT/r05_rec_timeline_h1024.log
python tools/gemm_shapes.py cfg3 2>&1 | grep -v amdgpu.ids > $OUT/r05_gemm_census_cfg3.log
python tools/solver_bench.py --warmup 45 --steps 30 2> $OUT/solver_bench.err | tail -1 > $OUT/r05_solver_loop_cfg3.json
for f in r05_bench_cfg3 r05_bench_shipped r05_bench_cfg3_rccl_world1 r05_bench_cfg2 r05_bench_cnn r05_solver_loop_cfg3; do head -c 300 $OUT/$f.json; echo; done
