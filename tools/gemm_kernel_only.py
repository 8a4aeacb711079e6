"""The split GEMM kernel alone (operands split once, only asrk_gemm_panels_f32 timed) on one NT shape, for the
routing knobs (ASRK_SPLIT_W256 / ASRK_SPLIT_TAIL).   python tools/gemm_kernel_only.py [M N K]"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ops = importlib.import_module("end-to-end-asr-pytorch_amd.ops")
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
M, N, K = nums if len(nums) == 3 else (25600, 8192, 4096)
A, B = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
C = torch.empty(M, N, device="cuda")
pa, pb = ops.SplitPanel(A, K, M, K, False), ops.SplitPanel(B, K, N, K, False)
run = lambda: ops.gemm_panels(M, N, K, pa, 0, 0, pb, 0, 0, C, N)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("DBG=%s CFG=%s %s M=%d N=%d K=%d kernel only: %.3f ms  %.1f TF/s-equivalent" % (
    os.environ.get("ASRK_SPLIT_DBG", "0"), os.environ.get("ASRK_SPLIT_CFG", "0"),
    "bf16x6", M, N, K, ms, 2.0 * M * N * K / ms * 1e-9), flush=True)
