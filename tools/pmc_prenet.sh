#!/bin/bash
# MFMA utilisation and LDS bank conflicts of the prenet kernels (PMC passes, kernel trace only):
#   tools/pmc_prenet.sh [vgg|cnn]   -> gpurun_out/pmc_prenet/summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; W=${1:-vgg}
OUT=$R/gpurun_out/pmc_prenet; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/p1 -- python $R/tools/prenet_bench.py $W --steps 3 > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python $R/tools/prenet_bench.py $W --steps 3 > $OUT/p2.log 2>&1
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
# units (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES = MFMA cycles summed over the 1024 SIMDs (64 per 32x32x2 f32 MFMA);
# GRBM_GUI_ACTIVE = active cycles summed over the 8 XCDs, so GRBM_GUI_ACTIVE / 8 / wall time = the effective clock
import csv, glob, os, sys, collections
out = sys.argv[1]
val = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, 'p*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        val[k][r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9)
m = lambda v: sum(v) / max(len(v), 1)
print('%-30s %5s %9s %10s %10s %12s %10s' % ('kernel', 'calls', 'us/call', 'clock GHz', 'MFMA util', 'LDS conflict', 'wait LDS'))
for k, d in sorted(val.items(), key=lambda kv: -m(kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', [0])) * len(kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', [0]))):
    if 'conv3x3' not in k or 'reduce' in k or 'weight' in k:
        continue
    gui = m(d['GRBM_GUI_ACTIVE']) / 8.0
    print('%-30s %5d %9.1f %10.2f %10.3f %12.3f %10.4f' % (k[:30], len(d['GRBM_GUI_ACTIVE']), m(dur[k]) * 1e6, gui / m(dur[k]) / 1e9,
          m(d['SQ_VALU_MFMA_BUSY_CYCLES']) / 1024.0 / gui, m(d['SQ_LDS_BANK_CONFLICT']) / max(m(d['SQ_LDS_IDX_ACTIVE']), 1.0),
          m(d['SQ_WAIT_INST_LDS']) / max(m(d['SQ_WAVE_CYCLES']), 1.0)))
print('MFMA util = MFMA-busy cycles per SIMD / active cycles; LDS conflict = conflict cycles / LDS-active cycles; '
      'wait LDS = wave cycles waiting on LDS / wave cycles (both passes are profiled runs: clocks ~3 % under an unprofiled run)')
PY
