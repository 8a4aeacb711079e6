"""The arithmetic claim behind csrc/gemm_split.hip and the bf16x6 recurrence kernels, checked on the CPU
(oracle/split_oracle.py): exact 3-way bf16 split of f32, and the size of the three dropped partial products."""
import numpy as np
import torch

from oracle import split_oracle as S


def _random_f32(n, seed, emin=-100, emax=100):
    g = np.random.default_rng(seed)
    mant = g.integers(0, 1 << 23, n, dtype=np.uint32)
    expo = g.integers(emin + 127, emax + 127, n, dtype=np.uint32)
    sign = g.integers(0, 2, n, dtype=np.uint32)
    bits = (sign << 31) | (expo << 23) | mant
    return torch.from_numpy(bits.view(np.float32).copy())


def test_three_bf16_pieces_reproduce_f32_exactly():
    a = torch.cat([_random_f32(200000, 1), torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e38, 1.17549435e-38 * 2 ** 30,
                                                          0.1, 1 / 3, 16777215.0, 1 + 2.0 ** -23])])
    a0, a1, a2 = S.split3(a)
    for p in (a0, a1, a2):                         # every piece is a bf16 number
        assert torch.equal(p.to(torch.bfloat16).to(torch.float32), p)
    assert torch.equal((a0.double() + a1.double() + a2.double()), a.double())
    nz = a != 0
    assert (a1[nz].abs() <= a[nz].abs() * 2.0 ** -8).all()      # |a1| <= half an ulp of bf16(a)
    assert (a2[nz].abs() <= a[nz].abs() * 2.0 ** -16).all()


def test_dropped_partial_products_are_below_one_f32_rounding():
    a, b = _random_f32(100000, 2, -30, 30), _random_f32(100000, 3, -30, 30)
    (a0, a1, a2), (b0, b1, b2) = S.split3(a), S.split3(b)
    kept = sum(x.double() * y.double() for x, y in ((a0, b0), (a0, b1), (a1, b0), (a1, b1), (a0, b2), (a2, b0)))
    exact = a.double() * b.double()
    rel = ((kept - exact).abs() / exact.abs()).max().item()
    assert rel <= 2.0 ** -23.9, rel                 # dropped a1 b2 + a2 b1 + a2 b2: below one f32 rounding (2^-24 .. 2^-23)
    assert rel > 0                                  # ... and they do exist


def test_six_term_gemm_matches_float64_like_an_f32_gemm():
    g = torch.Generator().manual_seed(4)
    A = torch.randn(64, 512, generator=g) * torch.exp(torch.randn(512, generator=g) * 2)
    B = torch.randn(48, 512, generator=g)
    ref = A.double() @ B.double().t()
    mag = (A.double().abs() @ B.double().abs().t()).max().item()
    err6 = (S.gemm6(A, B) - ref).abs().max().item() / mag
    err32 = ((A @ B.t()).double() - ref).abs().max().item() / mag        # ATen f32 GEMM on the same inputs
    assert err6 < 2.0 ** -24                       # truncation of the product only (float64 accumulation)
    assert err6 < err32                            # smaller than what f32 accumulation alone costs
