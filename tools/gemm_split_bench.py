"""Throughput and accuracy of the bf16x6 split GEMM (default) beside the exact-f32 MFMA kernel on the cfg3
contraction shapes.  GPU only.   python tools/gemm_split_bench.py [--acc]"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ops = importlib.import_module("end-to-end-asr-pytorch_amd.ops")
_lib = importlib.import_module("end-to-end-asr-pytorch_amd._lib")
L = _lib.load()

SHAPES = [  # (tag, mode, M, N, K)
    ("cfg3 L1 ih fwd", "NT", 25600, 8192, 4096), ("cfg3 L1 dX", "NN", 25600, 4096, 8192),
    ("cfg3 L1 dW_ih", "TN", 8192, 4096, 25600), ("cfg3 L2 ih fwd", "NT", 12800, 8192, 4096),
    ("cfg3 L3 ih fwd", "NT", 6400, 8192, 4096), ("cfg3 L0 dW_hh", "TN", 4096, 1024, 51200),
    ("cfg3 L3 dW_hh", "TN", 4096, 1024, 6400), ("cfg3 ctc head", "NT", 6400, 5000, 2048),
    ("cfg3 char", "NT", 2048, 5000, 1024), ("cfg2 L1 ih fwd", "NT", 16000, 4096, 2048),
    ("sq 4096", "NT", 4096, 4096, 4096), ("sq 8192", "NT", 8192, 8192, 8192),
    ("dec 2048x1024", "NT", 2048, 1024, 1024), ("dec 2048x4096", "NT", 2048, 4096, 3072),
    ("dec dW", "TN", 4096, 3072, 2048), ("key proj", "NT", 6400, 1024, 2048),
    ("small 1024", "NT", 1024, 1024, 1024), ("small 512", "NT", 512, 512, 2048),
    ("L0 ih fwd", "NT", 51200, 8192, 80), ("L0 dW_ih", "TN", 8192, 80, 51200),
]


def time_it(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


acc = "--acc" in sys.argv
only = [a[7:] for a in sys.argv if a.startswith("--only=")]
for tag, mode, M, N, K in SHAPES:
    if only and not any(o in tag for o in only):
        continue
    g = torch.Generator(device="cuda").manual_seed(1)
    if mode == "NT":
        A, B = torch.randn(M, K, device="cuda", generator=g), torch.randn(N, K, device="cuda", generator=g)
        run = lambda: ops.gemm(0, 1, M, N, K, A, K, B, K, C, N)
        ref64 = lambda r: A[:r].double() @ B.double().t()
    elif mode == "NN":
        A, B = torch.randn(M, K, device="cuda", generator=g), torch.randn(K, N, device="cuda", generator=g)
        run = lambda: ops.gemm(0, 0, M, N, K, A, K, B, N, C, N)
        ref64 = lambda r: A[:r].double() @ B.double()
    else:
        A, B = torch.randn(K, M, device="cuda", generator=g), torch.randn(K, N, device="cuda", generator=g)
        run = lambda: ops.gemm(1, 0, M, N, K, A, M, B, N, C, N)
        ref64 = lambda r: A[:, :r].double().t() @ B.double()
    C = torch.empty(M, N, device="cuda")
    fl = 2.0 * M * N * K
    out = []
    for split in (0, 2):
        ops.set_gemm_split(split)
        t = time_it(run)
        msg = "%s %7.3f ms %6.1f TF/s" % ("bf16x6" if split else "f32  ", t, fl / t * 1e-9)
        if acc:
            r = min(M, 512)
            ref = ref64(r)
            err = (C[:r].double() - ref).abs().max().item() / ref.abs().max().item()
            msg += " err %.2e" % err
        out.append(msg)
    ops.set_gemm_split(1)
    print("%-16s %s M=%6d N=%5d K=%6d | %s" % (tag, mode, M, N, K, " | ".join(out)), flush=True)
