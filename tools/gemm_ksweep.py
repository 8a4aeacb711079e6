"""t(K) = a + b*K decomposition of asrk_gemm_f32 vs the vendor GEMM (fixed M, N)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("end-to-end-asr-pytorch_amd.ops")
torch.backends.cuda.matmul.allow_tf32 = False

def time_it(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for (M, N) in ((16000, 4096), (8192, 8192), (4096, 2048)):
    for K in (256, 512, 1024, 2048, 4096, 8192):
        A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
        C = torch.empty(M, N, device="cuda")
        tm = time_it(lambda: ops.gemm(0, 1, M, N, K, A, K, B, K, C, N))
        tv = time_it(lambda: torch.matmul(A, B.t()))
        fl = 2.0 * M * N * K
        print("M=%5d N=%5d K=%5d  asrk %7.3f ms %6.1f TF/s | vendor %7.3f ms %6.1f TF/s" % (
            M, N, K, tm, fl / tm * 1e-9, tv, fl / tv * 1e-9))
