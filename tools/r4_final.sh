#!/bin/bash
# Round-4 closing evidence pass on the GPU box -> gpurun_out/r4_final/ (copy what is to be judged into profiles/).
# HBM traffic (stamped with the kernel-source digest) FIRST, so that the bench lines that follow carry
# roofline.traffic measured on exactly these kernel sources.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4_final; mkdir -p $OUT
cd $R
tools/pmc_hbm.sh cfg3 r4_pmc_hbm_cfg3 > $OUT/pmc_cfg3.log 2>&1
cp $R/gpurun_out/r4_pmc_hbm_cfg3/hbm_traffic_cfg3.json $OUT/r04_hbm_traffic_cfg3.json
cp $OUT/r04_hbm_traffic_cfg3.json $R/profiles/r04_hbm_traffic_cfg3.json
rm -rf $R/gpurun_out/r4_pmc_hbm_cfg3/FETCH_SIZE $R/gpurun_out/r4_pmc_hbm_cfg3/WRITE_SIZE
cd /tmp && export TMPDIR=/tmp
for W in cfg3 shipped; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -- \
      python $R/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-exact-check > $OUT/stats_$W.log 2>&1
  cp $(find $OUT/stats_$W -name "*kernel_stats.csv" | head -1) $OUT/r04_${W}_kernel_stats.csv
  if [ $W = cfg3 ]; then
    TR=$(find $OUT/stats_$W -name "*kernel_trace.csv" | head -1)
    python $R/tools/step_timeline.py $TR 300 4 > $OUT/r04_cfg3_step_timeline.log 2>&1
  fi
  rm -rf $OUT/stats_$W
done
cd $R
python bench.py --steps 20 --warmup 5 > $OUT/r04_bench_cfg3.json 2> $OUT/bench_cfg3.err
python bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r04_bench_cfg2.json 2> $OUT/bench_cfg2.err
python bench.py --workload shipped --steps 20 --warmup 5 > $OUT/r04_bench_shipped.json 2> $OUT/bench_shipped.err
python tools/solver_bench.py --warmup 45 --steps 30 2> $OUT/solver_bench.err | tail -1 > $OUT/r04_solver_loop_cfg3.json
python tools/gemm_shapes.py cfg3 2>&1 | grep -v amdgpu.ids > $OUT/r04_gemm_census_cfg3.log
python tools/rec_timeline.py 800 32 4096 1024 2>&1 | grep -v amdgpu.ids > $OUT/r04_rec_timeline_h1024.log
python tools/decode_bench.py --cpu-baseline 2> $OUT/decode.err | tail -1 > $OUT/r04_decode_cfg5.json
python tools/ctc_beam_bench.py --out $OUT/r04_ctc_beam.json > /dev/null 2> $OUT/ctc_beam.err
python tools/input_pipeline_bench.py --out $OUT/r04_input_pipeline.json > /dev/null 2> $OUT/input.err
for f in r04_bench_cfg3 r04_bench_cfg2 r04_bench_shipped r04_solver_loop_cfg3; do head -c 420 $OUT/$f.json; echo; done
tail -3 $OUT/pmc_cfg3.log
