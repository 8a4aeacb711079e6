#!/bin/bash
# round 4, first GPU pass: the new parity tests, the cfg3 line on this box, the shipped workload with kernel stats
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4_first; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_shipped_cfg2_gpu.py tests/test_optim_gpu.py tests/test_e2e_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
python bench.py --workload cfg3 --steps 10 --warmup 3 --no-cpu-baseline --no-exact-check > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python bench.py --workload shipped --steps 10 --warmup 3 --no-cpu-baseline --no-exact-check > $OUT/bench_shipped.json 2> $OUT/bench_shipped.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_shipped -- \
   python $R/bench.py --workload shipped --steps 3 --warmup 2 --no-cpu-baseline --no-exact-check > $OUT/stats_shipped.log 2>&1
cp $(find $OUT/stats_shipped -name "*kernel_stats.csv" | head -1) $OUT/r04_shipped_kernel_stats.csv
rm -rf $OUT/stats_shipped
cut -c1-300 $OUT/bench_cfg3.json; cut -c1-300 $OUT/bench_shipped.json
