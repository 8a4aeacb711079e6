"""CPU ORACLE (test infrastructure only - never imported by the product path).

Restates the arithmetic of csrc/gemm_split.hip (and of the bf16x6 recurrence kernels in csrc/lstm_rec.hip):
an f32 operand is split into three bf16 pieces with round-to-nearest-even conversions, and a product is
formed from the six partial products a_i * b_j with i + j <= 2.  This is not a restatement of reference code
(the reference reaches ATen's f32 GEMM, src/module.py:131): it pins the CLAIM the kernels rest on - the split
is exact and the three dropped partial products are below one f32 rounding of the product - on the CPU, where
`pytest -m "not gpu"` runs (tests/test_split_arith_cpu.py).  Parity of the kernels themselves is pinned on the
GPU against float64 (tests/test_kernels_gpu.py::test_gemm_split_*) and against the reference goldens.
"""
import numpy as np
import torch


def split3(a):
    """f32 tensor -> (a0, a1, a2) f32 tensors holding bf16-representable values, a0 + a1 + a2 == a"""
    a = a.to(torch.float32)
    a0 = a.to(torch.bfloat16).to(torch.float32)
    r1 = a - a0                                   # exact in f32
    a1 = r1.to(torch.bfloat16).to(torch.float32)
    r2 = r1 - a1                                  # exact in f32, <= 8 significant bits
    a2 = r2.to(torch.bfloat16).to(torch.float32)
    return a0, a1, a2


def gemm6(A, B):
    """A [M,K], B [N,K] f32 -> A B^T from the six partial products (float64 accumulation: isolates the error of
    the DROPPED terms from accumulation-order noise)"""
    a = [t.to(torch.float64) for t in split3(A)]
    b = [t.to(torch.float64) for t in split3(B)]
    out = torch.zeros(A.shape[0], B.shape[0], dtype=torch.float64)
    for i, j in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)):
        out += a[i] @ b[j].t()
    return out
