cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in 0 1 2; do
  rm -rf /tmp/fp$m
  ASRK_FILL_MODE=$m rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp$m -- python $R/tools/rec_timeline.py 800 32 4096 1024 > /tmp/fp$m.log 2>&1
  f=$(find /tmp/fp$m -name "*kernel_stats.csv" | head -1)
  echo "mode $m: $(grep sentinel_fill $f | cut -d, -f1-6 | cut -c1-150)"
done
