#!/bin/bash
# HBM traffic per kernel launch from the PMC counters (separate --pmc passes, kernel trace only),
# as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes:  bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024
# (FETCH_SIZE under-reports wide coalesced reads by exactly 2x on gfx950; WRITE_SIZE uncalibrated).
# usage: tools/pmc_hbm.sh <workload> <outname>     (run on the GPU box from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=${1:-cfg2}; NAME=${2:-pmc_hbm_$W}
OUT=$R/gpurun_out/$NAME; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -- \
      python $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-exact-check > $OUT/$C.log 2>&1
done
python3 - "$OUT" "$W" "$R" <<'PY'
import csv, sys, glob, collections, json, os
out, wl = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: {'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0, 'n': 0})
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(os.path.join(out, c, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != c:
                continue
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('asrk_rec::', '').replace('void ', '').split('(')[0]
            agg[k][c] += float(r['Counter_Value'])
            if c == 'FETCH_SIZE':
                agg[k]['n'] += 1
res = {}
for k, d in agg.items():
    n = max(d['n'], 1)
    res[k] = {'launches': d['n'], 'fetch_kb_per_launch': d['FETCH_SIZE'] / n, 'write_kb_per_launch': d['WRITE_SIZE'] / n,
              'hbm_bytes_per_launch': (2 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024 / n}
top = sorted(res.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:14]
import subprocess
lines = subprocess.run([sys.executable, os.path.join(sys.argv[3], 'bench.py'),
                        '--print-kernel-digest'], capture_output=True, text=True).stdout.strip().splitlines()
digest = lines[-1]
conv_digest = [l.split()[1] for l in lines if l.startswith('conv ')]
top = sorted(res.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:20]
json.dump({'workload': wl, 'formula': 'bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024', 'kernel_source_digest': digest,
           'conv_source_digest': conv_digest[0] if conv_digest else None,
           'kernels': dict(top)},
          open(os.path.join(out, 'hbm_traffic_%s.json' % wl), 'w'), indent=1)
for k, v in top:
    print('%-60s launches %4d  %10.1f MB/launch' % (k[:60], v['launches'], v['hbm_bytes_per_launch'] / 1e6))
PY
