#!/bin/bash
# the decode evidence of the round's closing pass (cfg5 line, one decode position at 32 / 1 utterances, kernel table):
#   tools/r6_decode_final.sh  -> gpurun_out/r6_final/
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r6_final; mkdir -p $OUT; cd $R
python bench.py --workload cfg5 > $OUT/r06_decode_cfg5.json 2> $OUT/bench_cfg5.err
python tools/decode_launches.py 32 2>/dev/null | grep '^{' > $OUT/r06_decode_position_32.json
python tools/decode_launches.py 1 2>/dev/null | grep '^{' > $OUT/r06_decode_position_1.json
(cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_dec -o d -- python $R/tools/decode_launches.py 32 > /dev/null 2>&1; cp $OUT/stats_dec/d_kernel_stats.csv $OUT/r06_decode_position_32_kernel_stats.csv; rm -rf $OUT/stats_dec)
for f in r06_decode_cfg5 r06_decode_position_32 r06_decode_position_1; do grep '^{' $OUT/$f.json | tail -1 | head -c 900; echo; done
