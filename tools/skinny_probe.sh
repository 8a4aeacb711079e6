#!/bin/bash
# timing experiments on the speller's skinny kernels: per-kernel average under ASRK_SKINNY_DBG skip masks
# (1 no weight loads, 2 no x loads, 4 no MFMAs, 8 no LDS staging; results are wrong)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for D in 0 1 2 3 4 8 15; do
  rm -rf /tmp/sk_$D
  ASRK_SKINNY_DBG=$D rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sk_$D -- \
     python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-check > /dev/null 2>&1
  f=$(find /tmp/sk_$D -name "*kernel_stats.csv" | head -1)
  echo "mask $D: $(python3 - "$f" <<'PY'
import csv, sys
out = []
for r in csv.DictReader(open(sys.argv[1])):
    if 'skinny_kernel' in r['Name']:
        out.append("%s %.1f" % (r['Name'].split('skinny_kernel')[1][:9], float(r['AverageNs']) / 1e3))
print("  ".join(sorted(out)))
PY
)"
done
