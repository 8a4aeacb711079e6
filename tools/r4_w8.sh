#!/bin/bash
# round 4: two-waves-per-SIMD forward recurrence (ASRK_FWD_W8 = 0 baseline | 1 MT4 half tiles | 2 MT2/NT2 split batch tiles)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r4_w8; mkdir -p $OUT
cd $R
for W in 0 1 2; do
  export ASRK_FWD_W8=$W
  echo "=== ASRK_FWD_W8=$W" | tee -a $OUT/timeline.log
  python tools/rec_timeline.py 800 32 4096 1024 2>&1 | grep -v amdgpu.ids | grep -A4 "== fwd" >> $OUT/timeline.log
  timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "cfg3_layer1_full_length or bf16x6_recurrence or lstm_repeatable" 2>&1 | tail -2 | tee -a $OUT/pytest_$W.log
  python bench.py --workload cfg3 --steps 10 --warmup 3 --no-cpu-baseline --no-exact-check > $OUT/bench_w$W.json 2> $OUT/bench_w$W.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_w$W.json").read().strip().splitlines()[-1])
print("W8=$W ms/step %.2f  lstm_fwd %.2f lstm_bwd %.2f gemm %.2f speller %.2f" % (d["ms_per_step"], d["kernel_families"]["lstm_fwd"]["ms_per_step"], d["kernel_families"]["lstm_bwd"]["ms_per_step"], d["kernel_families"]["gemm"]["ms_per_step"], d["kernel_families"]["speller"]["ms_per_step"]))
PY
done
cat $OUT/timeline.log
