#!/bin/bash
# Collect the round's measured evidence on the GPU box into gpurun_out/profiles_new/ (copy what should
# be judged into profiles/ afterwards).  usage: tools/collect_profiles.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_new; mkdir -p $OUT
# the stand-alone microbenchmarks (sources in tools/*.hip; the binaries are git-ignored)
for t in mfma_peak clock_probe pingpong mfma_ring; do
  [ -x $R/tools/$t ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $R/tools/$t.hip -o $R/tools/$t
done
cd /tmp && export TMPDIR=/tmp
for W in cfg2 cfg3; do
  ST=3; [ $W = cfg2 ] && ST=5
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -- \
      python $R/bench.py --workload $W --steps $ST --warmup 2 --no-cpu-baseline --no-exact-check > $OUT/stats_$W.log 2>&1
  cp $(find $OUT/stats_$W -name "*kernel_stats.csv" | head -1) $OUT/r01_${W}_kernel_stats.csv
  NL=2; [ $W = cfg3 ] && NL=4
  python $R/tools/step_timeline.py $(find $OUT/stats_$W -name "*kernel_trace.csv" | head -1) 8 $NL \
      > $OUT/r01_${W}_step_timeline.log 2>&1
done
cd $R
python bench.py > $OUT/r01_bench_cfg2.json 2> $OUT/bench_cfg2.err
python bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-exact-check > $OUT/r01_bench_cfg3.json 2> $OUT/bench_cfg3.err
python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/r01_gemm_vs_vendor.log
python tools/skinny_bench.py 2>&1 | grep -v amdgpu.ids >> $OUT/r01_gemm_vs_vendor.log
(./tools/mfma_peak; ./tools/clock_probe) > $OUT/r01_mfma_peak_clock.log 2>&1
python tools/rec_timeline.py 2>&1 | grep -v amdgpu.ids > $OUT/r01_rec_timeline_final.log
python tools/rec_timeline.py 1600 32 320 1024 2>&1 | grep -v amdgpu.ids >> $OUT/r01_rec_timeline_final.log
./tools/mfma_ring > $OUT/r01_mfma_ring.log 2>&1
python tools/corun_check.py 2>&1 | grep -v amdgpu.ids > $OUT/r01_corun_fwd_gemm.log
tools/pmc_hbm.sh cfg3 pmc_hbm_cfg3 > $OUT/pmc_cfg3.log 2>&1
cp $R/gpurun_out/pmc_hbm_cfg3/hbm_traffic_cfg3.json $OUT/r01_hbm_traffic_cfg3.json
tools/pmc_hbm.sh cfg2 pmc_hbm_cfg2 > $OUT/pmc_cfg2.log 2>&1
cp $R/gpurun_out/pmc_hbm_cfg2/hbm_traffic_cfg2.json $OUT/r01_hbm_traffic_cfg2.json
tail -1 $OUT/r01_bench_cfg2.json | cut -c1-400
