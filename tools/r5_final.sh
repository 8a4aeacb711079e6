#!/bin/bash
# Round-5 closing evidence pass on the GPU box -> gpurun_out/r5_final/ (what is to be judged is copied into profiles/).
# HBM traffic (stamped with the kernel-source digest) FIRST, so that the bench lines that follow carry
# roofline.traffic measured on exactly these kernel sources.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r5_final; mkdir -p $OUT
cd $R
for W in cfg3 shipped; do
  tools/pmc_hbm.sh $W r5_pmc_hbm_$W > $OUT/pmc_$W.log 2>&1
  cp $R/gpurun_out/r5_pmc_hbm_$W/hbm_traffic_$W.json $OUT/r05_hbm_traffic_$W.json
  cp $OUT/r05_hbm_traffic_$W.json $R/profiles/r05_hbm_traffic_$W.json
  rm -rf $R/gpurun_out/r5_pmc_hbm_$W/FETCH_SIZE $R/gpurun_out/r5_pmc_hbm_$W/WRITE_SIZE
done
cd /tmp && export TMPDIR=/tmp
for W in cfg3 shipped; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -- \
      python $R/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-exact-check > $OUT/stats_$W.log 2>&1
  cp $(find $OUT/stats_$W -name "*kernel_stats.csv" | head -1) $OUT/r05_${W}_kernel_stats.csv
  if [ $W = cfg3 ]; then
    TR=$(find $OUT/stats_$W -name "*kernel_trace.csv" | head -1)
    python $R/tools/step_timeline.py $TR 300 4 > $OUT/r05_cfg3_step_timeline.log 2>&1
    python $R/tools/split_insitu.py $TR 7 > $OUT/r05_split_insitu_cfg3.log 2>&1
  fi
  rm -rf $OUT/stats_$W
done
cd $R
python bench.py --steps 20 --warmup 5 > $OUT/r05_bench_cfg3.json 2> $OUT/bench_cfg3.err
python bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r05_bench_cfg2.json 2> $OUT/bench_cfg2.err
python bench.py --workload shipped --steps 20 --warmup 5 > $OUT/r05_bench_shipped.json 2> $OUT/bench_shipped.err
python bench.py --workload cnn --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r05_bench_cnn.json 2> $OUT/bench_cnn.err
ASRK_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-check 2> $OUT/bench_dist.err | grep '^{' > $OUT/r05_bench_cfg3_rccl_world1.json
python tools/solver_bench.py --warmup 45 --steps 30 2> $OUT/solver_bench.err | tail -1 > $OUT/r05_solver_loop_cfg3.json
python tools/gemm_shapes.py cfg3 2>&1 | grep -v amdgpu.ids > $OUT/r05_gemm_census_cfg3.log
python tools/rec_timeline.py 800 32 4096 1024 2>&1 | grep -v amdgpu.ids > $OUT/r05_rec_timeline_h1024.log
python tools/split_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/r05_split_bench.log
for f in r05_bench_cfg3 r05_bench_cfg2 r05_bench_shipped r05_bench_cnn r05_bench_cfg3_rccl_world1 r05_solver_loop_cfg3; do head -c 420 $OUT/$f.json; echo; done
tail -3 $OUT/pmc_cfg3.log
