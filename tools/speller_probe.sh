#!/bin/bash
# timing experiments on the speller backward kernels: per-kernel average under ASRK_SPELLER_DBG skip masks
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for D in 0; do
  rm -rf /tmp/sp_$D
  ASRK_SPELLER_DBG=$D rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_$D -- \
     python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  f=$(find /tmp/sp_$D -name "*kernel_stats.csv" | head -1)
  grep -E "energy_bwd|conv_bwd" $f | awk -F, "{print \$1, \$(NF-4)}"
done
