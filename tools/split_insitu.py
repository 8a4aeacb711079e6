"""Every split-pass launch of the LAST training step in a rocprofv3 kernel trace: duration, elements (from the grid)
and effective TB/s (4 B read + 6 B written per element), next to the same kernels' isolated rates (tools/split_bench.py).
usage: split_insitu.py <kernel_trace.csv> [steps_in_trace]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
              int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", 256) or 256))
             for r in rows))
sp = [e for e in ev if "split_panel" in e[2]]
opt = [e for e in ev if "sqnorm_final_kernel" in e[2]]        # one gradient-norm launch closes every training step
last = [e for e in sp if opt[-2][1] <= e[0] <= opt[-1][0]] if len(opt) >= 2 else []
if not last:
    per = len(sp) // steps if steps else len(sp)
    last = sp[-per:] if per else sp
tot_t = tot_b = 0.0
prev_end = {}
for s, e, k, grid, wgs in last:
    n_wg = grid // wgs if grid >= wgs else grid
    elems = n_wg * 4 * 64 * 32                       # 4 waves x (64 rows x 32 k) per workgroup, padding included
    us = (e - s) / 1e3
    before = max((x for x in ev if x[1] <= s), key=lambda x: x[1], default=None)
    tag = "T" if "split_panel_t_kernel" in k else "N"
    print("%s %9.1f us  %8.1f M elements  %5.2f TB/s   after %s" % (
        tag, us, elems / 1e6, elems * 10 / us * 1e-6, before[2].split("(")[0][-48:] if before else "-"))
    tot_t += us
    tot_b += elems * 10
print("last step: %d split launches, %.3f ms, %.2f TB/s overall" % (len(last), tot_t / 1e3, tot_b / tot_t * 1e-6))
