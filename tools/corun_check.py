"""Debug tool (GPU): can a GEMM share the chip with the forward LSTM recurrence?

The forward recurrence occupies every CU (one persistent workgroup each) but keeps the matrix pipe
busy only ~25 % of a step; the next layer's input projection could in principle run in the shadow.
Times the cfg2 layer-0 recurrence alone, a cfg2 layer-1-sized GEMM alone, and both at once on two
streams.    python tools/corun_check.py [T B D H]
"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("end-to-end-asr-pytorch_amd.ops")
lib = importlib.import_module("end-to-end-asr-pytorch_amd._lib").load()

T, B, D, H = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else (1000, 32, 80, 512)
dev = "cuda"
g = torch.Generator().manual_seed(0)
x = torch.randn(T, B, D, generator=g).to(dev)
ps = [torch.randn(4 * H, D, generator=g) / D ** 0.5, torch.randn(4 * H, H, generator=g) / H ** 0.5,
      torch.zeros(4 * H), torch.zeros(4 * H)]
pf = tuple(p.to(dev) for p in ps)
pr = tuple((p * 0.9).to(dev) for p in ps)
M, N, K = 16000, 2048, 2048
A = torch.randn(M, K, device=dev)
W = torch.randn(N, K, device=dev)
C = torch.empty(M, N, device=dev)
side = torch.cuda.Stream()
REP = 3   # GEMMs per recurrence


def rec():
    with torch.no_grad():
        return ops.lstm_layer(x, pf, pr)


def gemms():
    for _ in range(REP):
        ops.gemm(0, 1, M, N, K, A, K, W, K, C, N)


def timed(fn_main, fn_side):
    for _ in range(2):
        if fn_main: fn_main()
        if fn_side:
            with torch.cuda.stream(side): fn_side()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    if fn_main:
        e[0].record(); fn_main(); e[1].record()
    if fn_side:
        with torch.cuda.stream(side):
            e[2].record(side); fn_side(); e[3].record(side)
    torch.cuda.synchronize()
    return (e[0].elapsed_time(e[1]) if fn_main else 0.0, e[2].elapsed_time(e[3]) if fn_side else 0.0)


for hint in (0, 64):
    ops._gemm_state['lds_hint'] = hint
    a, _ = timed(rec, None)
    _, b = timed(None, gemms)
    c, d = timed(rec, gemms)
    print("launch hint %3d KiB: layer alone %.2f ms | %d GEMMs alone %.2f ms | together: layer %.2f ms, GEMMs %.2f ms"
          % (hint, a, REP, b, c, d))
ops._gemm_state['lds_hint'] = 0
ops.check_errors()
