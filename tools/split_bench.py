"""The split pass alone (asrk_split_panel_f32) on the cfg3 operand shapes, both storage orders: ms and effective
TB/s (4 B read + 6 B written per element).   python tools/split_bench.py"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ops = importlib.import_module("end-to-end-asr-pytorch_amd.ops")
SHAPES = [("dG^T L0", True, 8192, 51200), ("dG^T L1", True, 8192, 25600), ("Y^T  L0", True, 2048, 51200),
          ("X^T  L1", True, 4096, 25600), ("W^T  ih", True, 4096, 8192), ("dG   L1", False, 25600, 8192),
          ("X    L1", False, 25600, 4096), ("W    ih", False, 8192, 4096)]
for tag, trans, rows, K in SHAPES:
    src = torch.randn((K, rows) if trans else (rows, K), device="cuda")
    ld = rows if trans else K
    f = lambda: ops.SplitPanel(src, ld, rows, K, trans)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%s %s rows=%6d K=%6d  %7.3f ms  %5.2f TB/s" % (tag, "T" if trans else "N", rows, K, ms,
                                                         rows * K * 10.0 / ms * 1e-9), flush=True)
