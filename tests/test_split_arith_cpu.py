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


# ------------------------------------------------------------------------------ the OPT-IN fp16x4 variant
def _split2_f16(a):
    """csrc/gemm_split.hip split8_f16 / row_exp_of, restated: every ROW is scaled by the exact power of two that
    brings its largest magnitude into [2^13, 2^14), then split into two fp16 numbers (round-to-nearest)"""
    a = a.double()
    amax = a.abs().amax(dim=1, keepdim=True)
    e = torch.where(amax > 0, 13 - torch.floor(torch.log2(amax.clamp_min(1e-300))), torch.zeros_like(amax))
    scaled = a * torch.pow(2.0, e)
    h0 = scaled.to(torch.float16).double()
    h1 = (scaled - h0).to(torch.float16).double()
    return h0, h1, scaled, e


def test_two_fp16_pieces_keep_22_bits_relative_to_the_row_maximum():
    g = torch.Generator().manual_seed(7)
    a = torch.randn(512, 1024, generator=g) * torch.exp(3 * torch.randn(512, 1024, generator=g))   # wide dynamic range in a row
    a[3] = 0.0
    a[5, :100] = 1e-30                               # far below the row maximum
    h0, h1, scaled, _ = _split2_f16(a)
    assert h0.abs().max() < 2.0 ** 14 and torch.isfinite(h0).all() and torch.isfinite(h1).all()   # nothing overflows fp16
    r = (scaled - h0 - h1).abs()
    # elements near the row maximum: relative 2^-22; small elements (h1 subnormal): absolute 2^-25 of the SCALED row,
    # i.e. <= 2^-38 of the row maximum
    assert (r <= torch.maximum(scaled.abs() * 2.0 ** -22, torch.full_like(r, 2.0 ** -25))).all()
    # every partial product h_i * h_j has <= 22 significant bits: exact in the f32 accumulator of the MFMA
    p = (h0[:64, :64].float() * h1[:64, :64].t().float()).double()
    assert torch.equal(p, h0[:64, :64] * h1[:64, :64].t())


def test_four_term_gemm_error_is_below_the_f32_accumulation_error_for_deep_contractions():
    g = torch.Generator().manual_seed(8)
    K = 4096
    A = torch.randn(48, K, generator=g) * torch.exp(torch.randn(K, generator=g) * 2)
    B = torch.randn(40, K, generator=g)
    a0, a1, sa, ea = _split2_f16(A)
    b0, b1, sb, eb = _split2_f16(B)
    four = (a0 @ b0.t() + a0 @ b1.t() + a1 @ b0.t() + a1 @ b1.t()) * torch.pow(2.0, -ea) * torch.pow(2.0, -eb).t()
    ref = A.double() @ B.double().t()
    mag = (A.double().abs() @ B.double().abs().t()).max().item()
    err4 = (four - ref).abs().max().item() / mag               # operand rounding only (float64 accumulation here)
    err32 = ((A @ B.t()).double() - ref).abs().max().item() / mag
    assert err4 < 2.0 ** -21 and err4 < err32, (err4, err32)
