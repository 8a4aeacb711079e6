"""Print the kernel timeline (start, duration, queue, name) of the last full training step in a
rocprofv3 kernel_trace.csv (steps are delimited by the first lstm_rec_fwd launch of each step).
usage: step_timeline.py <kernel_trace.csv> [min_us] [fwd recurrence launches per step, default 2]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r['Kernel_Name']) for r in rows))
fw = [i for i, e in enumerate(ev) if 'lstm_rec_fwd' in e[3]]
# cfg2 has two forward recurrence launches per step (L0, L1), cfg3 four
per_step = int(sys.argv[3]) if len(sys.argv) > 3 else 2
starts = fw[::per_step]
a, b = starts[-2], starts[-1]
t0 = ev[a][0]
print("step span %.2f ms" % ((ev[b][0] - t0) / 1e6))
for s, e, q, k in ev[a:b]:
    d = (e - s) / 1e3
    if d >= min_us:
        name = k.replace('(anonymous namespace)::', '').replace('void ', '')[:70]
        print("%9.3f ms  +%8.1f us  q%-3s %s" % ((s - t0) / 1e6, d, q, name))
