"""Which Python lines put non-asrk device work on a training step?  A TorchDispatchMode over a few steps of a bench.py
workload records every ATen op that touches a device tensor (copies, fills, element-wise arithmetic - the library's own
kernels are not ATen ops), with the bytes it moves and the innermost frames inside this repository.
usage: python tools/residue_trace.py [--workload cfg3] [--steps 2]"""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SKIP = ("aten.view", "aten.detach", "aten.t.", "aten.transpose", "aten.reshape", "aten.as_strided", "aten.select",
        "aten.slice", "aten.unsqueeze", "aten.squeeze", "aten.expand", "aten.alias", "aten._unsafe_view", "aten.permute",
        "aten.empty", "aten.unbind", "aten.split", "aten.narrow", "aten.lift_fresh", "aten._local_scalar_dense",
        "aten.is_", "aten.size", "aten.stride", "aten.numel", "aten.item", "aten.empty_like", "aten.new_empty",
        "aten.resize_", "aten.set_", "aten.record_stream", "aten.unflatten", "aten.flatten", "aten.view_as")


class Recorder(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.agg = collections.defaultdict(lambda: [0, 0])

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if name.startswith(SKIP):
            return out
        tens = [a for a in list(args) + [out] if isinstance(a, torch.Tensor) and a.is_cuda]
        if not tens:
            return out
        nbytes = max(t.numel() * t.element_size() for t in tens)
        frames = [f for f in traceback.extract_stack()[:-1]
                  if ("end-to-end-asr-pytorch_amd" in f.filename or f.filename.endswith("bench.py"))]
        where = " < ".join("%s:%d" % (f.filename.split("end-to-end-asr-pytorch_amd/")[-1].split("/")[-1] if "amd/" not in f.filename
                                       else f.filename.split("end-to-end-asr-pytorch_amd/")[-1], f.lineno)
                           for f in reversed(frames[-3:])) or "(autograd engine / no repo frame)"
        k = (name, where)
        self.agg[k][0] += 1
        self.agg[k][1] += nbytes
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--steps", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model, step = bench.build_step(a.workload, dev)
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    rec = Recorder()
    with rec:
        for _ in range(a.steps):
            step()
    torch.cuda.synchronize()
    print("%-34s %7s %12s  %s" % ("ATen op on device tensors", "n/step", "MB/step", "innermost repo frames"))
    for (name, where), (n, nb) in sorted(rec.agg.items(), key=lambda kv: -kv[1][1]):
        print("%-34s %7.1f %12.2f  %s" % (name[:34], n / a.steps, nb / a.steps / 1e6, where))


if __name__ == "__main__":
    main()
