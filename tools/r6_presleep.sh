#!/bin/bash
# round 6 experiment: idle time before a recurrence step's first canary poll (ASRK_FWD_PRESLEEP / ASRK_BWD_PRESLEEP, x64 cycles)
# and the poll flavours (ASRK_FWD_POLL / ASRK_BWD_POLL: 1 pipelined polls, 2 one polling wave per workgroup) against the
# recurrence families of cfg2 (H = 512) and cfg3 (H = 1024), same box.  usage: tools/r6_presleep.sh [sweep2]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() {  # workload env...
  W=$1; shift
  env "$@" python bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-exact-check 2>/dev/null | grep '^{"metric"' | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); f = d['kernel_families']
print('  %-6s %-44s ms/step %7.2f  fwd %6.2f  bwd %6.2f' % ('$W', '$*', d['ms_per_step'], f['lstm_fwd']['ms_per_step'], f['lstm_bwd']['ms_per_step']))"
}
if [ "$1" = "sweep2" ]; then
  for W in cfg3 cfg2; do
    run $W X=0
    for F in 6 10 12; do run $W ASRK_FWD_PRESLEEP=$F; done
    for P in 1 2 3; do run $W ASRK_FWD_POLL=$P; done
    for P in 0 2 3; do run $W ASRK_BWD_POLL=$P; done
    run $W X=0
  done
  exit 0
fi
for W in cfg2 cfg3; do
  run $W X=0
  for F in 0 4 8 24; do run $W ASRK_FWD_PRESLEEP=$F; done
  for B in 4 8 16; do run $W ASRK_BWD_PRESLEEP=$B; done
  run $W X=0
done
