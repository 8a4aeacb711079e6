"""Per-shape GEMM throughput of asrk_gemm_f32 on the cfg2 / cfg3 contraction shapes, beside the
vendor library (torch.matmul fp32 -> hipBLASLt/rocBLAS) as a yardstick.  GPU only.
    python tools/gemm_bench.py"""
import importlib
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ops = importlib.import_module("end-to-end-asr-pytorch_amd.ops")
torch.backends.cuda.matmul.allow_tf32 = False

SHAPES = [  # (tag, mode, M, N, K)
    ("cfg2 L1 ih fwd", "NT", 16000, 4096, 2048), ("cfg2 L1 dX", "NN", 16000, 2048, 4096),
    ("cfg2 L1 dW_ih", "TN", 4096, 2048, 16000), ("cfg2 L0 ih fwd", "NT", 32000, 4096, 80),
    ("cfg2 L0 dW_hh", "TN", 2048, 512, 32000), ("cfg2 head fwd", "NT", 8000, 5000, 2048),
    ("cfg2 head dW", "TN", 5000, 2048, 8000), ("cfg2 head dX", "NN", 8000, 2048, 5000),
    ("cfg3 L1 ih fwd", "NT", 25600, 8192, 4096), ("cfg3 L1 dW_ih", "TN", 8192, 4096, 25600),
    ("cfg3 dec cell", "NT", 32, 4096, 3072), ("cfg3 char", "NT", 2048, 5000, 1024),
]


def time_it(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for tag, mode, M, N, K in SHAPES:
    if mode == "NT":
        A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
        mine = lambda: ops.gemm(0, 1, M, N, K, A, K, B, K, C, N)
        ref = lambda: torch.matmul(A, B.t())
    elif mode == "NN":
        A, B = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda")
        mine = lambda: ops.gemm(0, 0, M, N, K, A, K, B, N, C, N)
        ref = lambda: torch.matmul(A, B)
    else:
        A, B = torch.randn(K, M, device="cuda"), torch.randn(K, N, device="cuda")
        mine = lambda: ops.gemm(1, 0, M, N, K, A, M, B, N, C, N)
        ref = lambda: torch.matmul(A.t(), B)
    C = torch.empty(M, N, device="cuda")
    t_m, t_r = time_it(mine), time_it(ref)
    fl = 2.0 * M * N * K
    print("%-16s %s M=%6d N=%5d K=%6d  asrk %7.3f ms %6.1f TF/s | vendor %7.3f ms %6.1f TF/s" % (
        tag, mode, M, N, K, t_m, fl / t_m * 1e-9, t_r, fl / t_r * 1e-9))
