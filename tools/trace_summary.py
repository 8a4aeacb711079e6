"""Summarise a rocprofv3 kernel_trace.csv: per-kernel totals, busy time (union of intervals) and
idle gaps for the last `steps` training steps.  usage: trace_summary.py <kernel_trace.csv> [n_top]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ntop = int(sys.argv[2]) if len(sys.argv) > 2 else 25
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows))
t0, t1 = ev[0][0], max(e[1] for e in ev)
# take the last 40% of the trace (steady state)
cut = t0 + (t1 - t0) * 0.6
ev = [e for e in ev if e[0] >= cut]
span = max(e[1] for e in ev) - ev[0][0]
busy, cur_s, cur_e = 0, None, None
for s, e, _ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k in ev:
    a = agg[k[:90]]; a[0] += e - s; a[1] += 1
print("window %.2f ms, GPU busy (union) %.2f ms = %.1f%%, sum of kernel durations %.2f ms" % (
    span / 1e6, busy / 1e6, 100.0 * busy / span, sum(a[0] for a in agg.values()) / 1e6))
for k, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:ntop]:
    print("%9.3f ms %5d x %8.1f us  %s" % (t / 1e6, n, t / n / 1e3, k))
