"""Debug tool (GPU): phase timeline of the decoder loop's kernels (csrc/speller.hip SP_STAMP) over one cfg3 training step.
For every kernel of the step pair: mean over the decode steps of the shader-clock deltas between its phase stamps (first
workgroup, wave 0), the start of its last workgroup, and the gap to the kernel before it.  ~2100 cycles per us.
    python tools/speller_timeline.py [--workload cfg3]"""
import argparse
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

PKG = "end-to-end-asr-pytorch_amd"
NAMES = ["F1 query", "F2a energy", "F2b softmax+ctx", "F3 cell", "B2 dG W", "B3 dattn", "B4 energy bwd", "B5 conv bwd",
         "B6 dh+cell bwd"]
CYC = 2100.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    a = ap.parse_args()
    lib = importlib.import_module(PKG + "._lib").load()
    lib.asrk_speller_set_debug_.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.asrk_speller_set_debug_.restype = None
    dev = torch.device("cuda:0")
    model, step = bench.build_step(a.workload, dev)
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    L = bench.WORKLOADS[a.workload]["L"]
    buf = torch.zeros(L * 16 * 16, dtype=torch.int64, device=dev)
    lib.asrk_speller_set_debug_(ctypes.c_void_p(buf.data_ptr()), L * 16)
    step()
    torch.cuda.synchronize()
    lib.asrk_speller_set_debug_(None, 0)
    s = buf.cpu().numpy().astype(np.int64).reshape(L, 16, 16)
    print("phase stamps per kernel (us, mean over %d decode steps; '-' = not stamped)" % L)
    order_f, order_b = [0, 1, 2, 3], [4, 5, 6, 7, 8]
    for kid in range(9):
        st = s[:, kid, :]
        ok = st[:, 0] > 0
        if not ok.any():
            print("%-16s (no stamps)" % NAMES[kid])
            continue
        st = st[ok]
        t0 = st[:, 0]
        cols = []
        prev = t0
        for ph in range(1, 10):
            v = st[:, ph]
            if (v > 0).all():
                cols.append("%d:+%.2f" % (ph, float((v - prev).mean()) / CYC))
                prev = v
        last_start = (st[:, 12] - t0).mean() / CYC if (st[:, 12] > 0).all() else float("nan")
        last_end = (st[:, 13] - t0).mean() / CYC if (st[:, 13] > 0).all() else float("nan")
        total = (st[:, 9] - t0).mean() / CYC if (st[:, 9] > 0).all() else float("nan")
        # gap: this kernel's first stamp minus the previous kernel's end stamp (same decode step)
        grp = order_f if kid in order_f else order_b
        i = grp.index(kid)
        gap = float("nan")
        if i > 0:
            pe = s[:, grp[i - 1], :][ok]
            pend = np.maximum(pe[:, 9], pe[:, 13])
            good = pend > 0
            if good.any():
                gap = float((t0[good] - pend[good]).mean()) / CYC
        print("%-16s wg0 total %6.2f | last wg starts +%5.2f ends +%5.2f | gap after previous kernel %5.2f | %s" % (
            NAMES[kid], total, last_start, last_end, gap, " ".join(cols)))
    # step-pair time from the stamps: F1 start of step t+1 minus F1 start of step t; B2 start of t-1 minus B2 start of t
    f = s[:, 0, 0]
    if (f > 0).sum() > 2:
        d = np.diff(f[f > 0])
        print("forward step (F1 start -> next F1 start): %.2f us" % (np.median(d) / CYC))
    bb = s[:, 4, 0]
    if (bb > 0).sum() > 2:
        d = -np.diff(bb[bb > 0])
        print("backward step (B2 start -> next B2 start): %.2f us" % (np.median(d) / CYC))


if __name__ == "__main__":
    main()
