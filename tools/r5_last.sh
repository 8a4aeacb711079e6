#!/bin/bash
# one traced cfg3 run -> split launches of a training step; then the recurrence-facing part of the GPU suite on the final sources
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5_last; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-exact-check > $OUT/tr.log 2>&1
TR=$(find $OUT/tr -name "*kernel_trace.csv" | head -1)
python $R/tools/split_insitu.py $TR 10 > $OUT/r05_split_insitu_cfg3.log 2>&1
rm -rf $OUT/tr
cd $R
(timeout 330 python -m pytest tests/test_kernels_gpu.py tests/test_parallel_gpu.py tests/test_shipped_cfg2_gpu.py -m gpu -x -q -k "lstm or gru or rec or panel or encoder or rccl or solver or shipped or cnn" > $OUT/r05_pytest_gpu_after_dgt.log 2>&1; echo rc=$? >> $OUT/r05_pytest_gpu_after_dgt.log)
tail -4 $OUT/r05_split_insitu_cfg3.log; tail -4 $OUT/r05_pytest_gpu_after_dgt.log
