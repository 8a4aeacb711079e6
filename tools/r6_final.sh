#!/bin/bash
# Round-6 closing evidence pass on the GPU box -> gpurun_out/r6_final/ (what is to be judged is copied into profiles/).
# usage: tools/r6_final.sh pmc     HBM traffic per kernel (PMC passes, stamped with the kernel-source digest) for all four
#                                  training workloads -> profiles/r06_hbm_traffic_<w>.json
#        tools/r6_final.sh bench   kernel stats + step timeline + the bench lines (run AFTER pmc: the lines then carry
#                                  roofline.traffic measured on exactly these kernel sources)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r6_final; mkdir -p $OUT
cd $R
if [ "$1" = "pmc" ]; then
  for W in cfg3 shipped cfg2 cnn; do
    tools/pmc_hbm.sh $W r6_pmc_hbm_$W > $OUT/pmc_$W.log 2>&1
    cp $R/gpurun_out/r6_pmc_hbm_$W/hbm_traffic_$W.json $OUT/r06_hbm_traffic_$W.json
    cp $OUT/r06_hbm_traffic_$W.json $R/profiles/r06_hbm_traffic_$W.json   # the bench lines of the same call read it
    rm -rf $R/gpurun_out/r6_pmc_hbm_$W/FETCH_SIZE $R/gpurun_out/r6_pmc_hbm_$W/WRITE_SIZE
    tail -4 $OUT/pmc_$W.log
  done
  exit 0
fi
cd /tmp && export TMPDIR=/tmp
for W in cfg3 shipped; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -- \
      python $R/bench.py --workload $W --steps 12 --warmup 3 --no-cpu-baseline --no-exact-check > $OUT/stats_$W.log 2>&1
  cp $(find $OUT/stats_$W -name "*kernel_stats.csv" | head -1) $OUT/r06_${W}_kernel_stats.csv
  if [ $W = cfg3 ]; then
    TR=$(find $OUT/stats_$W -name "*kernel_trace.csv" | head -1)
    python $R/tools/step_timeline.py $TR 300 4 > $OUT/r06_cfg3_step_timeline.log 2>&1
    python $R/tools/split_insitu.py $TR 7 > $OUT/r06_split_insitu_cfg3.log 2>&1
  fi
  rm -rf $OUT/stats_$W
done
cd $R
python bench.py --steps 20 --warmup 5 > $OUT/r06_bench_cfg3.json 2> $OUT/bench_cfg3.err
python bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r06_bench_cfg2.json 2> $OUT/bench_cfg2.err
python bench.py --workload shipped --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r06_bench_shipped.json 2> $OUT/bench_shipped.err
python bench.py --workload cnn --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r06_bench_cnn.json 2> $OUT/bench_cnn.err
python bench.py --workload cfg5 > $OUT/r06_decode_cfg5.json 2> $OUT/bench_cfg5.err
ASRK_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-check 2> $OUT/bench_dist.err | grep '^{' > $OUT/r06_bench_cfg3_rccl_world1.json
python tools/solver_bench.py --warmup 45 --steps 30 2> $OUT/solver_bench.err | tail -1 > $OUT/r06_solver_loop_cfg3.json
python tools/rec_timeline.py 800 32 4096 1024 2>&1 | grep -v amdgpu.ids > $OUT/r06_rec_timeline_h1024.log
python tools/rec_timeline.py 500 32 2048 512 2>&1 | grep -v amdgpu.ids > $OUT/r06_rec_timeline_h512.log
python tools/speller_timeline.py 2>&1 | grep -v amdgpu.ids > $OUT/r06_speller_timeline.log
python tools/prenet_bench.py vgg 2>/dev/null | grep '^{' > $OUT/r06_prenet_vgg.json
ASRK_CONV_DIRECT=0 python tools/prenet_bench.py vgg 2>/dev/null | grep '^{' > $OUT/r06_prenet_vgg_im2col.json
python tools/prenet_bench.py cnn 2>/dev/null | grep '^{' > $OUT/r06_prenet_cnn.json
python tools/decode_launches.py 32 2>/dev/null | grep '^{' > $OUT/r06_decode_position_32.json
python tools/decode_launches.py 1 2>/dev/null | grep '^{' > $OUT/r06_decode_position_1.json
(cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_prenet -o vgg -- python $R/tools/prenet_bench.py vgg --steps 10 > /dev/null 2>&1; cp $OUT/stats_prenet/vgg_kernel_stats.csv $OUT/r06_prenet_vgg_kernel_stats.csv; rm -rf $OUT/stats_prenet)
(cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_dec -o d -- python $R/tools/decode_launches.py 32 > /dev/null 2>&1; cp $OUT/stats_dec/d_kernel_stats.csv $OUT/r06_decode_position_32_kernel_stats.csv; rm -rf $OUT/stats_dec)
python tools/residue_trace.py 2>&1 | grep -v amdgpu.ids > $OUT/r06_residue_after.log
for f in r06_bench_cfg3 r06_bench_cfg2 r06_bench_shipped r06_bench_cnn r06_decode_cfg5 r06_bench_cfg3_rccl_world1 r06_solver_loop_cfg3 r06_prenet_vgg r06_prenet_vgg_im2col r06_prenet_cnn r06_decode_position_32 r06_decode_position_1; do grep '^{' $OUT/$f.json | tail -1 | head -c 420; echo; done
