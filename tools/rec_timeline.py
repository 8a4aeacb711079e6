"""Debug tool (GPU): per-phase timeline of workgroup 0 inside the persistent LSTM kernels.

    python tools/rec_timeline.py [T B D H]

phases: 0 step start | 1 exchange fragments loaded sentinel-free | 2 MFMAs done | 3 partials in LDS
        4 past the barrier | 5 cell update done, exchange (sc1) stores issued | 6 saved-tensor
        stores + next-step prefetch issued
"""
import ctypes
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "end-to-end-asr-pytorch_amd"
ops = importlib.import_module(PKG + ".ops")
lib = importlib.import_module(PKG + "._lib").load()

T, B, D, H = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else (1000, 32, 80, 512)
dev = "cuda"
g = torch.Generator().manual_seed(0)
x = torch.randn(T, B, D, generator=g).to(dev).requires_grad_(True)
ps = [torch.randn(4 * H, D, generator=g) / D ** 0.5, torch.randn(4 * H, H, generator=g) / H ** 0.5,
      torch.zeros(4 * H), torch.zeros(4 * H)]
pf = tuple(p.to(dev).requires_grad_(True) for p in ps)
pr = tuple((p * 0.9).to(dev).requires_grad_(True) for p in ps)

STEPS = min(T, 400)
lib.asrk_lstm_set_debug_.argtypes = [ctypes.c_void_p, ctypes.c_int]


def report(tag, buf, ms):
    a = buf.cpu().numpy().astype(np.int64).reshape(STEPS, 4, 8)
    span = a[STEPS - 1, 0, 6] - a[1, 0, 0]
    cyc_per_us = span / (ms * 1e3 * (STEPS - 1) / T) if ms > 0 else float("nan")
    print("== %s: kernel %.3f ms for T=%d (%.2f us/step); ~%.0f cycles/us" % (
        tag, ms, T, ms * 1e3 / T, cyc_per_us))
    names = ["canary(0-7)", "bulk ld(7-1)", "mfma(1-2)", "lds wr(2-3)", "barrier(3-4)",
             "cell+xchg st(4-5)", "tail st/ld(5-6)"]
    for w in range(4):
        t = a[50:STEPS - 1, w, :][:, [0, 7, 1, 2, 3, 4, 5, 6]]
        d = np.diff(t, axis=1).astype(np.float64).mean(0)
        nxt = (a[51:STEPS, w, 0] - a[50:STEPS - 1, w, 6]).mean()
        tot = (a[51:STEPS, w, 0] - a[50:STEPS - 1, w, 0]).mean()
        print(" wave %d: %s | loop %5.0f | step %6.0f cyc" % (
            w, " ".join("%s %5.0f" % (n, v) for n, v in zip(names, d)), nxt, tot))


for tag in ("fwd", "bwd"):
    buf = torch.zeros(STEPS * 4 * 8, dtype=torch.int64, device=dev)
    if tag == "fwd":
        lib.asrk_lstm_set_debug_(ctypes.c_void_p(buf.data_ptr()), STEPS)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        lib.asrk_profile_reset(); lib.asrk_profile_enable(1)
        y = ops.lstm_layer(x, pf, pr)
        lib.asrk_profile_enable(0)
        ms, n = ctypes.c_double(0), ctypes.c_int64(0)
        lib.asrk_profile_get(1, ctypes.byref(ms), ctypes.byref(n))
        lib.asrk_lstm_set_debug_(None, 0)
        torch.cuda.synchronize()
        report("fwd", buf, ms.value)
    else:
        gy = torch.randn(y.shape, generator=g).to(dev)
        lib.asrk_lstm_set_debug_(ctypes.c_void_p(buf.data_ptr()), STEPS)
        lib.asrk_profile_reset(); lib.asrk_profile_enable(1)
        y.backward(gy)
        lib.asrk_profile_enable(0)
        ms, n = ctypes.c_double(0), ctypes.c_int64(0)
        lib.asrk_profile_get(2, ctypes.byref(ms), ctypes.byref(n))
        lib.asrk_lstm_set_debug_(None, 0)
        torch.cuda.synchronize()
        report("bwd", buf, ms.value)
ops.check_errors()
