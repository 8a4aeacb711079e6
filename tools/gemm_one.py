"""Run asrk_gemm_f32 a few times on one shape (for rocprofv3 --pmc passes).
    python tools/gemm_one.py NT 16000 4096 2048 [iters]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("end-to-end-asr-pytorch_amd.ops")
mode, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
C = torch.empty(M, N, device="cuda")
if mode == "NT":
    A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    f = lambda: ops.gemm(0, 1, M, N, K, A, K, B, K, C, N)
elif mode == "NN":
    A, B = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda")
    f = lambda: ops.gemm(0, 0, M, N, K, A, K, B, N, C, N)
else:
    A, B = torch.randn(K, M, device="cuda"), torch.randn(K, N, device="cuda")
    f = lambda: ops.gemm(1, 0, M, N, K, A, M, B, N, C, N)
for _ in range(iters):
    f()
torch.cuda.synchronize()
