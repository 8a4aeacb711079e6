#!/bin/bash
# same-box A/B of the bf16x6 split policy on the H = 512 workloads: tools/r6_split_ab.sh [workloads...]
# ASRK_GEMM_SPLIT=1: where asrk_gemm_takes_split says it pays (default); 2: wherever the shape allows
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for W in ${@:-cfg2 shipped}; do for S in 1 2; do
  ASRK_GEMM_SPLIT=$S python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-exact-check 2>/dev/null | grep '^{' | \
    python -c "
import sys, json
d = json.loads(sys.stdin.read()); kf = d['kernel_families']
print('$W ASRK_GEMM_SPLIT=$S', round(d['ms_per_step'], 2), 'ms/step', {k: round(v['ms_per_step'], 2) for k, v in kf.items() if v['ms_per_step'] > 0.05})"
done; done
