"""Skinny-M GEMM shapes of the decoder loop / beam search, asrk vs vendor."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("end-to-end-asr-pytorch_amd.ops")
torch.backends.cuda.matmul.allow_tf32 = False

def time_it(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for mode, M, N, K in (("NT", 32, 4096, 3072), ("NN", 32, 3072, 4096), ("NN", 32, 1024, 4096), ("NT", 32, 4096, 1024),
                      ("NT", 32, 300, 1024), ("NN", 32, 1024, 300), ("NT", 16, 5000, 1024), ("NT", 16, 4096, 3072),
                      ("NT", 64, 4096, 3072), ("NN", 64, 3072, 4096)):
    if mode == "NT":
        A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
        mine = lambda: ops.gemm(0, 1, M, N, K, A, K, B, K, C, N)
        ref = lambda: torch.matmul(A, B.t())
    else:
        A, B = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda")
        mine = lambda: ops.gemm(0, 0, M, N, K, A, K, B, N, C, N)
        ref = lambda: torch.matmul(A, B)
    C = torch.empty(M, N, device="cuda")
    tm, tv = time_it(mine), time_it(ref)
    gb = (N * K + M * K + M * N) * 4 / 1e9
    print("%s M=%3d N=%5d K=%5d  asrk %6.1f us (%5.2f TB/s) | vendor %6.1f us (%5.2f TB/s)" % (
        mode, M, N, K, tm, gb / tm * 1e3, tv, gb / tv * 1e3))
