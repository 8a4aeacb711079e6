#!/bin/bash
# round 6: rocprofv3 kernel stats of a short cfg3 bench run under a list of environment settings (same box)
# usage: tools/r6_ab.sh TAG "ENV1=.. ENV2=.." "ENV=.." ...   ("-" = no extra environment)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for E in "$@"; do
  i=$((i+1))
  [ "$E" = "-" ] && E=""
  rm -rf /tmp/p_$i
  env $E rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$i -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-exact-check > $OUT/run_$i.log 2>&1
  f=$(find /tmp/p_$i -name "*kernel_stats.csv" | head -1)
  cp "$f" $OUT/kernel_stats_$i.csv
  echo "== [$i] '$E'" >> $OUT/summary.log
  grep '^{"metric"' $OUT/run_$i.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], {k: round(v['ms_per_step'], 2) for k, v in d['kernel_families'].items()}, d.get('pools'))" >> $OUT/summary.log 2>&1
  python3 - "$f" >> $OUT/summary.log <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keys = ('skinny_kernel', 'attend_energy', 'softmax_context', 'dattn_kernel', 'energy_bwd', 'conv_bwd', 'dvalue')
for r in rows:
    n = r['Name']
    if any(k in n for k in keys):
        n = n.replace('(anonymous namespace)::', '').replace('void ', '')
        print("  %-44s calls %6s avg %8.1f us  total %8.2f ms" % (n.split('(')[0][:44], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
PY
  rm -rf /tmp/p_$i
done
cat $OUT/summary.log
