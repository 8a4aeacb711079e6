"""Cycles per K-tile iteration of asrk_gemm_f32's main loop at controlled occupancy
(1 WG on the chip / 1 WG per CU / 2 WGs per CU), splitk forced to 1."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("end-to-end-asr-pytorch_amd.ops")

def time_it(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for mode in ("NT", "TN", "NN"):
    for (M, N, K, tag) in ((128, 128, 65536, "1 WG"), (2048, 2048, 16384, "256 WG"), (4096, 2048, 16384, "512 WG"),
                           (4096, 4096, 16384, "1024 WG"), (8192, 4096, 8192, "2048 WG")):
        if mode == "NT":
            A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
            f = lambda: ops.gemm(0, 1, M, N, K, A, K, B, K, C, N, splitk=1)
        elif mode == "NN":
            A, B = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda")
            f = lambda: ops.gemm(0, 0, M, N, K, A, K, B, N, C, N, splitk=1)
        else:
            A, B = torch.randn(K, M, device="cuda"), torch.randn(K, N, device="cuda")
            f = lambda: ops.gemm(1, 0, M, N, K, A, M, B, N, C, N, splitk=1)
        C = torch.empty(M, N, device="cuda")
        t = time_it(f)
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        rounds = max(1.0, tiles / 512.0) if tiles > 256 else 1.0
        cyc = t * 1e-3 * 2.4e9 / (K / 32) / rounds
        print("%s %-8s M=%5d N=%5d K=%6d %8.3f ms  %6.1f TF/s  ~%6.0f cycles per K-tile per WG-slot (ideal 4096)" % (
            mode, tag, M, N, K, t, 2.0 * M * N * K / t * 1e-9, cyc))
