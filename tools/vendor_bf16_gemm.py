"""Yardstick only (never on the product path): what the vendor library (hipBLASLt through torch.matmul) sustains on
this chip for plain bf16 GEMMs of the cfg3 extents - the practical ceiling of a power-throttled MI355X that the
bf16x6 kernel's 417 TF/s-equivalent nominal peak (2500 / 6) has to be read against.   GPU only."""
import torch

for (M, N, K) in ((25600, 8192, 4096), (25600, 8192, 24576), (8192, 8192, 8192), (8192, 4096, 25600)):
    A = torch.randn(M, K, device="cuda").bfloat16()
    B = torch.randn(N, K, device="cuda").bfloat16()
    C = A @ B.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        C = A @ B.t()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    tf = 2.0 * M * N * K / ms * 1e-9
    print("vendor bf16 NT M=%6d N=%5d K=%6d  %8.3f ms  %7.1f TF/s  (= %.1f TF/s-equivalent at 6 products per f32 product)"
          % (M, N, K, ms, tf, tf / 6), flush=True)
