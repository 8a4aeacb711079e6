"""Which Python lines put non-asrk device work on a training step?  torch.profiler (with_stack) over a few cfg3 steps:
every ATen op that launched a device kernel or a memcpy / memset, grouped by the innermost frame inside this repository.
usage: python tools/residue_trace.py [--workload cfg3] [--steps 3]"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model, step = bench.build_step(a.workload, dev)
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0, set()])
    for ev in prof.events():
        if not ev.name.startswith("aten::") and "Memcpy" not in ev.name and "Memset" not in ev.name:
            continue
        dt = getattr(ev, "self_device_time_total", None)
        if dt is None:
            dt = getattr(ev, "self_cuda_time_total", 0)
        if not dt:
            continue
        where = "?"
        for fr in (ev.stack or []):
            if ROOT in fr and "tools/residue_trace.py" not in fr:
                where = fr.replace(ROOT + "/", "")
                break
        shapes = str(ev.input_shapes)[:60] if ev.input_shapes else ""
        k = (ev.name, where)
        agg[k][0] += 1
        agg[k][1] += dt
        agg[k][2].add(shapes)
    print("%-28s %-70s %7s %10s" % ("op", "innermost repo frame", "n/step", "us/step"))
    tot = 0.0
    for (name, where), (n, us, shapes) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        tot += us
        print("%-28s %-70s %7.1f %10.1f  %s" % (name, where[:70], n / a.steps, us / a.steps, sorted(shapes)[:2]))
    print("total device time of ATen / memcpy / memset work: %.1f us per step" % (tot / a.steps))


if __name__ == "__main__":
    main()
