"""GPU parity: every HIP kernel (through the C ABI / ops layer) vs the CPU oracle on the same
seeded inputs.  Tolerance: 1e-3 relative fp32 (BASELINE.json north_star) unless tighter is noted;
integer outputs exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import asr_oracle as O
from helpers import load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def t(x):
    return torch.as_tensor(x).to(DEV)


# ------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (257, 131, 83), (1000, 2048, 80), (64, 5000, 2048),
                                   (4096, 512, 1000), (1, 1, 1), (33, 17, 5)])
@pytest.mark.parametrize("mode", ["NT", "NN", "TN"])
def test_gemm_matches_fp64(ops, M, N, K, mode):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(K, N, generator=g)
    bias = torch.randn(N, generator=g)
    ref = (a.double() @ b.double() + bias.double()).numpy()
    C = torch.empty(M, N, device=DEV)
    if mode == "NT":
        A, Bm = t(a), t(b.t().contiguous())
        ops.gemm(0, 1, M, N, K, A, K, Bm, K, C, N, bias=t(bias))
    elif mode == "NN":
        A, Bm = t(a), t(b)
        ops.gemm(0, 0, M, N, K, A, K, Bm, N, C, N, bias=t(bias))
    else:
        A, Bm = t(a.t().contiguous()), t(b)
        ops.gemm(1, 0, M, N, K, A, M, Bm, N, C, N, bias=t(bias))
    scale = np.abs(a.double().numpy()) @ np.abs(b.double().numpy()) + 1.0
    assert np.max(np.abs(C.cpu().numpy() - ref) / scale) < 2e-6


def test_gemm_splitk_beta_strided(ops):
    g = torch.Generator().manual_seed(5)
    M, N, K = 96, 200, 4096
    a = torch.randn(K, M, generator=g)       # TN
    b = torch.randn(K, N + 8, generator=g)   # ldb > N
    c0 = torch.randn(M, N + 4, generator=g)  # ldc > N
    ref = c0.double().clone()
    ref[:, :N] += 0.5 * (a.double().t() @ b.double()[:, :N])
    C = t(c0.clone())
    ops.gemm(1, 0, M, N, K, t(a), M, t(b), N + 8, C, N + 4, alpha=0.5, beta=1.0, splitk=8)
    assert rel_err(C.cpu(), ref) < 1e-5
    C2 = t(c0.clone())
    ops.gemm(1, 0, M, N, K, t(a), M, t(b), N + 8, C2, N + 4, alpha=0.5, beta=0.0, splitk=0)
    ref2 = c0.double().clone()
    ref2[:, :N] = 0.5 * (a.double().t() @ b.double()[:, :N])
    assert rel_err(C2.cpu(), ref2) < 1e-5


def test_linear_autograd(ops):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(7, 13, 50, generator=g)
    w = torch.randn(31, 50, generator=g)      # N=31: scalar (unaligned) epilogue/loader paths
    b = torch.randn(31, generator=g)
    xr, wr, br = [v.clone().requires_grad_(True) for v in (x, w, b)]
    yr = F.linear(xr, wr, br)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)
    xg, wg, bg = [v.clone().to(DEV).requires_grad_(True) for v in (x, w, b)]
    y = ops.linear(xg, wg, bg)
    y.backward(gy.to(DEV))
    assert rel_err(y.detach().cpu(), yr.detach()) < 1e-5
    assert rel_err(xg.grad.cpu(), xr.grad) < 1e-5
    assert rel_err(wg.grad.cpu(), wr.grad) < 1e-5
    assert rel_err(bg.grad.cpu(), br.grad) < 1e-5


# ------------------------------------------------------------------------------ row ops
@pytest.mark.parametrize("rows,cols", [(37, 5000), (5, 31), (1, 4), (260, 16000)])
def test_log_softmax_fwd_bwd(ops, rows, cols):
    g = torch.Generator().manual_seed(rows + cols)
    x = (torch.randn(rows, cols, generator=g) * 3).requires_grad_(True)
    gy = torch.randn(rows, cols, generator=g)
    yr = F.log_softmax(x, dim=-1)
    yr.backward(gy)
    xg = x.detach().clone().to(DEV).requires_grad_(True)
    y = ops.log_softmax(xg)
    y.backward(gy.to(DEV))
    assert torch.max(torch.abs(y.detach().cpu() - yr.detach())).item() < 2e-5
    assert rel_err(xg.grad.cpu(), x.grad) < 1e-4


@pytest.mark.parametrize("T0", [37, 36])   # 36: every frame is overwritten (no zero fill in backward)
def test_swap_and_pyramid(ops, T0):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(5, T0, 24, generator=g)           # [B,T,F]
    xt = ops.swap_bt(t(x))
    assert torch.equal(xt.cpu(), x.transpose(0, 1).contiguous())
    for rate, style in [(2, "concat"), (3, "concat"), (2, "drop"), (4, "drop")]:
        xg = xt.detach().clone().requires_grad_(True)
        y = ops.pyramid(xg, rate, style)
        B, T, Fd = x.shape
        if style == "concat":
            ref = x[:, :T - T % rate].contiguous().view(B, T // rate, Fd * rate)
        else:
            ref = x[:, ::rate].contiguous()
        assert torch.equal(y.detach().cpu(), ref.transpose(0, 1).contiguous())
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy.to(DEV))
        xr = x.clone().requires_grad_(True)
        if style == "concat":
            yr = xr[:, :T - T % rate].contiguous().view(B, T // rate, Fd * rate)
        else:
            yr = xr[:, ::rate].contiguous()
        yr.backward(gy.transpose(0, 1))
        assert torch.equal(xg.grad.cpu(), xr.grad.transpose(0, 1).contiguous())


@pytest.mark.parametrize("n0,n1,n2", [(7, 5, 256), (33, 3, 512), (250, 32, 1024), (1, 1, 260), (3, 70000, 256)])
def test_copy3d_wide_rows(ops, n0, n1, n2):
    """rows of >= 64 float4: the row-walking kernel (pyramid concat / un-concat shapes), with strides
    that leave gaps on both sides, and accumulate=True"""
    g = torch.Generator().manual_seed(n0 + n1 + n2)
    ss1, ds1 = n2 + 8, 2 * n2
    ss0, ds0 = n1 * ss1 + 16, n1 * ds1 + 4
    src = torch.randn(n0 * ss0, generator=g)
    dst0 = torch.randn(n0 * ds0, generator=g)
    def view(flat, s0, s1):
        return torch.as_strided(flat, (n0, n1, n2), (s0, s1, 1))
    for acc in (False, True):
        ref = dst0.clone()
        view(ref, ds0, ds1).copy_(view(src, ss0, ss1) + (view(dst0, ds0, ds1) if acc else 0))
        d = t(dst0.clone())
        ops.copy3d(t(src), d, n0, n1, n2, ss0, ss1, ds0, ds1, accumulate=acc)
        assert torch.equal(d.cpu(), ref)


def test_colsum(ops):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(3001, 517, generator=g)
    out = torch.empty(517, device=DEV)
    ops.colsum(t(x), 3001, 517, 517, out)
    assert rel_err(out.cpu(), x.double().sum(0)) < 1e-5


@pytest.mark.parametrize("M,N,ld,off", [(3001, 516, 1040, 0), (37, 260, 520, 260), (1, 4, 8, 4),
                                         (4099, 2048, 4096, 2048)])
def test_colsum_vector_path(ops, M, N, ld, off):
    """16-B aligned, N % 4 == 0: float4 kernel; a column block of a wider matrix (the per-direction
    gate-gradient blocks the bias gradients are summed from), plus accumulate=True."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, ld, generator=g)
    xd = t(x)
    out = torch.empty(N, device=DEV)
    ops.colsum(xd.view(-1)[off:], M, N, ld, out)
    ref = x.double()[:, off:off + N].sum(0)
    assert rel_err(out.cpu(), ref) < 1e-5
    ops.colsum(xd.view(-1)[off:], M, N, ld, out, accumulate=True)
    assert rel_err(out.cpu(), 2 * ref) < 1e-5


# ------------------------------------------------------------------------------ CTC
def test_ctc_loss_golden(ops):
    g = load_golden("ctc_loss")
    lp = t(g["log_probs"]).requires_grad_(True)
    loss = ops.CTCLoss(blank=0)(lp, t(g["targets"]), t(g["input_lengths"]), t(g["target_lengths"]))
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    assert rel_err(lp.grad.cpu(), g["grad"]) < 1e-3


@pytest.mark.parametrize("T,B,V,L", [(250, 32, 5000, 64), (50, 7, 31, 20), (200, 4, 100, 1)])
def test_ctc_loss_vs_oracle(ops, T, B, V, L):
    g = torch.Generator().manual_seed(T + B)
    logits = torch.randn(B, T, V, generator=g)
    lp_bm = logits.log_softmax(-1)                      # [B,T,V] like ctc_output
    txt = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(1, L + 1, (1,), generator=g))
        txt[b, :n] = torch.randint(1, min(V, 6), (n,), generator=g)   # few symbols -> many repeats
    tl = (txt != 0).sum(-1)
    il = torch.randint(2 * L + 2, T + 1, (B,), generator=g)
    il[0] = T
    lr = lp_bm.clone().requires_grad_(True)
    ref = F.ctc_loss(lr.transpose(0, 1), txt, il, tl, blank=0, reduction="mean")
    ref.backward()
    lg = lp_bm.clone().to(DEV).requires_grad_(True)
    # non-contiguous [T,B,V] view exactly as bin/train_asr.py:123 passes it
    loss = ops.CTCLoss(blank=0)(lg.transpose(0, 1), t(txt), t(il), t(tl))
    loss.backward()
    assert abs(loss.item() - ref.item()) < 1e-4 * abs(ref.item())
    assert rel_err(lg.grad.cpu(), lr.grad) < 1e-3


# ------------------------------------------------------------------------------ LSTM layer
def _lstm_case(ops, T, B, D, H, bidir, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, D, generator=g)
    names = ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")
    shapes = ((4 * H, D), (4 * H, H), (4 * H,), (4 * H,))
    sd = {}
    for sfx in ([""] + (["_reverse"] if bidir else [])):
        for n, s in zip(names, shapes):
            sd["p." + n + sfx] = (torch.randn(*s, generator=g) / np.sqrt(s[-1] if len(s) > 1 else 4.0)
                                  ).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    yr = O.lstm_layer(xr, sd, "p.", bidir, impl="aten")
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy)

    xg = x.transpose(0, 1).contiguous().to(DEV).requires_grad_(True)   # time-major
    pf = [sd["p." + n].detach().clone().to(DEV).requires_grad_(True) for n in names]
    pr = [sd["p." + n + "_reverse"].detach().clone().to(DEV).requires_grad_(True)
          for n in names] if bidir else None
    y = ops.lstm_layer(xg, tuple(pf), tuple(pr) if pr else None)
    y.backward(gy.transpose(0, 1).contiguous().to(DEV))
    ops.check_errors()
    assert rel_err(y.detach().cpu().transpose(0, 1), yr.detach()) < 1e-3, "forward"
    assert rel_err(xg.grad.cpu().transpose(0, 1), xr.grad) < 1e-3, "dx"
    for i, n in enumerate(names):
        assert rel_err(pf[i].grad.cpu(), sd["p." + n].grad) < 1e-3, n
        if bidir:
            assert rel_err(pr[i].grad.cpu(), sd["p." + n + "_reverse"].grad) < 1e-3, n + "_reverse"


@pytest.mark.parametrize("T,B,D,H,bidir", [
    (9, 3, 12, 16, True),        # tiny, ragged tiles everywhere
    (37, 5, 20, 32, True),
    (50, 32, 80, 512, True),     # cfg2 layer-0 shape, short T
    (40, 32, 256, 1024, True),   # cfg3 width (H=1024 plan: MT=2,NT=2 / UB=8)
    (33, 17, 24, 64, False),     # unidirectional, odd batch
    (21, 40, 16, 128, True),     # B > 32 -> batch groups
    (12, 4, 8, 20, True),        # H % 16 != 0 (K padding), H % 4 == 0
])
def test_lstm_layer_fwd_bwd(ops, T, B, D, H, bidir):
    _lstm_case(ops, T, B, D, H, bidir, seed=T * 100 + H)


@pytest.mark.parametrize("T,B,D,H,bidir", [
    (12, 64, 32, 1024, True),    # directions x batch groups x slices > 256 CUs: one launch per direction
    (9, 130, 16, 512, True),     # 9 batch groups of 16: several launches over batch-group ranges
    (7, 200, 8, 1024, False),    # unidirectional, B > what fits beside each other
])
def test_lstm_layer_oversize_shapes_run_as_several_launches(ops, T, B, D, H, bidir):
    """maximum sizes: shapes whose independent (direction, batch-group) recurrences do not fit the
    chip at once are split over several launches of the persistent kernels; results are unchanged"""
    _lstm_case(ops, T, B, D, H, bidir, seed=T * 1000 + B)


@pytest.mark.parametrize("T,B,D,H,bidir", [(64, 32, 96, 1024, True), (48, 20, 64, 512, True), (30, 32, 64, 512, False)])
def test_lstm_bf16x6_recurrence_equals_f32_recurrence(ops, monkeypatch, T, B, D, H, bidir):
    """forward recurrence on the bf16 matrix cores (exact 3-way operand split, lstm_rec_fwd_bf_kernel) vs the
    f32-MFMA kernel (ASRK_REC_BF=0) on the same layer: the outputs agree to f32 rounding, far inside the
    1e-3 bar (both are also checked against the oracle in test_lstm_layer_fwd_bwd)"""
    g = torch.Generator().manual_seed(T + H)
    x = torch.randn(T, B, D, generator=g).to(DEV)
    shapes = ((4 * H, D), (4 * H, H), (4 * H,), (4 * H,))
    pf = tuple((torch.randn(*s, generator=g) / np.sqrt(s[-1] if len(s) > 1 else 4.0)).to(DEV) for s in shapes)
    pr = tuple((torch.randn(*s, generator=g) / np.sqrt(s[-1] if len(s) > 1 else 4.0)).to(DEV) for s in shapes) \
        if bidir else None
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("ASRK_REC_BF", flag)
        with torch.no_grad():
            outs.append(ops.lstm_layer(x, pf, pr).cpu())
        ops.check_errors()
    assert torch.isfinite(outs[0]).all()
    if bidir:
        assert not torch.equal(outs[0], outs[1])      # two different kernels did run (MT = 2 plans)
    assert (outs[0] - outs[1]).abs().max().item() < 2e-5 * max(1.0, outs[1].abs().max().item())


@pytest.mark.parametrize("T,B,D,H", [(48, 32, 96, 1024), (40, 20, 64, 512)])
def test_lstm_bf16x6_bptt_equals_f32_bptt(ops, monkeypatch, T, B, D, H):
    """BPTT on the bf16 matrix cores (lstm_rec_bwd_bf_kernel) vs the f32-MFMA BPTT kernel (ASRK_REC_BF_BWD=0):
    input and parameter gradients of the same layer agree to f32 rounding"""
    g = torch.Generator().manual_seed(T * 7 + H)
    x0 = torch.randn(T, B, D, generator=g)
    shapes = ((4 * H, D), (4 * H, H), (4 * H,), (4 * H,))
    p0 = [torch.randn(*s, generator=g) / np.sqrt(s[-1] if len(s) > 1 else 4.0) for s in shapes * 2]
    dy = torch.randn(T, B, 2 * H, generator=g).to(DEV)
    grads = []
    for flag in ("1", "0"):
        monkeypatch.setenv("ASRK_REC_BF_BWD", flag)
        x = x0.clone().to(DEV).requires_grad_(True)
        ps = [q.clone().to(DEV).requires_grad_(True) for q in p0]
        y = ops.lstm_layer(x, tuple(ps[:4]), tuple(ps[4:]))
        y.backward(dy)
        ops.join_deferred()
        ops.check_errors()
        grads.append([x.grad.cpu()] + [q.grad.cpu() for q in ps])
    assert any(not torch.equal(a, b) for a, b in zip(*grads))      # two different kernels did run
    for a, b in zip(*grads):
        assert torch.isfinite(a).all()
        assert (a - b).abs().max().item() < 3e-5 * max(1e-3, b.abs().max().item())


def _lstm_case_pyramid(ops, T, B, D, H, rate, style, seed, share_panels=None, monkeypatch=None):
    """bidirectional layer + the time reduction of src/module.py:141-153 through the FUSED store path
    (asrk_lstm_rec_{fwd,bwd}_pyr_f32) against ATen lstm + the reference's reshape / slicing on the host"""
    if share_panels is not None:
        monkeypatch.setenv("ASRK_SHARE_PANELS", share_panels)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, D, generator=g)
    names = ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")
    shapes = ((4 * H, D), (4 * H, H), (4 * H,), (4 * H,))
    sd = {}
    for sfx in ("", "_reverse"):
        for n, s in zip(names, shapes):
            sd["p." + n + sfx] = (torch.randn(*s, generator=g) / np.sqrt(s[-1] if len(s) > 1 else 4.0)
                                  ).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    yr = O.lstm_layer(xr, sd, "p.", True, impl="aten")                  # [B,T,2H]
    if style == "concat":                                                 # src/module.py:146-150
        Tr = T - T % rate
        yr2 = yr[:, :Tr].contiguous().view(B, Tr // rate, rate * 2 * H)
    else:                                                                 # src/module.py:143-144
        yr2 = yr[:, ::rate]
    gy = torch.randn(yr2.shape, generator=g)
    yr2.backward(gy)

    xg = x.transpose(0, 1).contiguous().to(DEV).requires_grad_(True)
    pf = [sd["p." + n].detach().clone().to(DEV).requires_grad_(True) for n in names]
    pr = [sd["p." + n + "_reverse"].detach().clone().to(DEV).requires_grad_(True) for n in names]
    y = ops.lstm_layer(xg, tuple(pf), tuple(pr), pyramid=(rate, style))
    y.backward(gy.transpose(0, 1).contiguous().to(DEV))
    ops.join_deferred()
    ops.check_errors()
    assert y.shape == (yr2.shape[1], B, yr2.shape[2])
    assert rel_err(y.detach().cpu().transpose(0, 1), yr2.detach()) < 1e-3, "forward"
    assert rel_err(xg.grad.cpu().transpose(0, 1), xr.grad) < 1e-3, "dx"
    for i, n in enumerate(names):
        assert rel_err(pf[i].grad.cpu(), sd["p." + n].grad) < 1e-3, n
        assert rel_err(pr[i].grad.cpu(), sd["p." + n + "_reverse"].grad) < 1e-3, n + "_reverse"


def test_lstm_cfg3_layer0_full_length(ops):
    """BASELINE configs[2] layer 0 at its real length: T=1600 dependent steps of the bf16x6 forward and BPTT
    kernels at H=1024, shallow-K (D=80) input projection, weight-gradient GEMMs with K = T*B = 51200 over the
    shared dG^T / Y^T panels - forward, dx and every parameter gradient vs ATen lstm on the host
    (src/module.py:131)"""
    _lstm_case(ops, 1600, 32, 80, 1024, True, seed=1600)


def test_lstm_cfg3_layer1_full_length(ops):
    """BASELINE configs[2] layer 1: T=800, Din=4096, H=1024 (the biggest contractions of the step:
    25600x8192x4096 input projection and its two gradients)"""
    _lstm_case(ops, 800, 32, 4096, 1024, True, seed=800)


@pytest.mark.parametrize("share", ["1", "0"])
def test_lstm_cfg3_layer0_full_length_fused_pyramid(ops, monkeypatch, share):
    """the same layer with the 'concat' time reduction fused into the recurrence kernels' stores / loads
    (what the bench runs), with the weight-gradient GEMMs on shared split panels and without"""
    _lstm_case_pyramid(ops, 1600, 32, 80, 1024, 2, "concat", seed=1601, share_panels=share, monkeypatch=monkeypatch)


def test_lstm_fused_pyramid_drop_long(ops):
    """'drop' time reduction (src/module.py:143-144), odd length, H=1024"""
    _lstm_case_pyramid(ops, 401, 32, 256, 1024, 2, "drop", seed=401)


def test_lstm_long_sequence_cfg2(ops):
    """full cfg2 length: T=1000 through the persistent kernels (1000 in-kernel grid syncs)."""
    _lstm_case(ops, 1000, 32, 80, 512, True, seed=77)


def test_lstm_repeatable(ops):
    """grid-sync protocol: two launches on the same inputs must agree bit-for-bit"""
    g = torch.Generator().manual_seed(8)
    T, B, D, H = 200, 32, 64, 512
    x = torch.randn(T, B, D, generator=g).to(DEV)
    ps = [torch.randn(4 * H, D, generator=g) / 8, torch.randn(4 * H, H, generator=g) / 22,
          torch.zeros(4 * H), torch.zeros(4 * H)]
    pf = tuple(p.to(DEV) for p in ps)
    pr = tuple((p * 0.9).to(DEV) for p in ps)
    with torch.no_grad():
        y1 = ops.lstm_layer(x, pf, pr).clone()
        y2 = ops.lstm_layer(x, pf, pr).clone()
    ops.check_errors()
    assert torch.equal(y1, y2)


@pytest.mark.parametrize("T,B,D,H,bidir", [
    (1, 3, 8, 16, True),         # a single step: nothing to exchange, the tail fill covers the only region
    (2, 5, 8, 32, True),         # two steps: every region belongs to the tail fill
    (3, 5, 8, 32, False),        # the first in-kernel re-arm (region 0 at step 2)
    (37, 5, 20, 32, True),       # f32 kernels, ragged tiles
    (64, 32, 80, 512, True),     # bf16x6 kernels, H = 512
    (48, 32, 256, 1024, True),   # bf16x6 kernels, the cfg3 plans (16 units x 16 rows, 256 workgroups)
    (12, 64, 32, 1024, True),    # more groups than CUs: several launches share the buffer -> one full fill behind them
])
def test_exchange_buffers_come_back_armed(ops, T, B, D, H, bidir):
    """ASRK_REC_REARM (include/asrk.h): a recurrence launch hands its exchange buffer back sentinel-filled, so the
    pooled buffer serves the next launch with xchg_prefilled = 1 and no fill pass.  (1) after forward + backward the
    pooled buffers hold 0xFF in EVERY byte; (2) three forward + backward passes over the same pooled buffers give
    bit-identical outputs and gradients, equal to a run that fills per launch (pool off)."""
    g = torch.Generator().manual_seed(T * 31 + H)
    x = torch.randn(T, B, D, generator=g).to(DEV)
    gy = torch.randn(T, B, (2 if bidir else 1) * H, generator=g).to(DEV)
    mk = lambda sc: tuple(p.to(DEV).requires_grad_(True) for p in (
        torch.randn(4 * H, D, generator=g) * sc / D ** 0.5, torch.randn(4 * H, H, generator=g) * sc / H ** 0.5,
        torch.randn(4 * H, generator=g) * 0.1, torch.randn(4 * H, generator=g) * 0.1))
    pf, pr = mk(1.0), (mk(0.9) if bidir else None)

    def run():
        xg = x.clone().requires_grad_(True)
        for p in pf + (pr or ()):
            p.grad = None
        y = ops.lstm_layer(xg, pf, pr)
        y.backward(gy)
        ops.join_deferred()
        return [y.detach().clone(), xg.grad.clone()] + [p.grad.clone() for p in pf + (pr or ())]

    assert ops._XCHG_REARM
    ops.drop_exchange_pool()
    runs = [run() for _ in range(3)]
    ops.check_errors()
    pooled = [e for lst in ops._xchg_pool["free"].values() for e in lst]
    assert len(pooled) == 2                                   # the forward and the backward exchange of this shape
    for buf, armed, _ in pooled:
        assert 0 < armed <= buf.numel()
        assert bool((buf[:armed] == 0xFF).all()), "exchange buffer not fully re-armed"
    def same(a, c):
        # the recurrences are deterministic: outputs bit for bit; gradients pass through GEMMs whose split-K partial
        # sums meet in atomics (order-dependent rounding), hence 1e-5 of the largest element there
        assert torch.equal(a[0], c[0])
        for u, v in zip(a[1:], c[1:]):
            assert float((u - v).abs().max()) <= 1e-5 * float(u.abs().max()) + 1e-12

    for r in runs[1:]:
        same(runs[0], r)
    try:
        ops._XCHG_REARM = False
        ref = run()
        ops.check_errors()
    finally:
        ops._XCHG_REARM = True
        ops.drop_exchange_pool()
    same(runs[0], ref)


def test_exchange_pool_serves_changing_lengths(ops):
    """Round 6 (advisor): the pool is keyed by (device, stream) and hands any armed buffer that is big enough to any
    shape - a training epoch has a different T in almost every batch.  Lengths that go up and down over the same pooled
    buffers give the results of fill-per-launch runs, every pooled buffer stays armed over its recorded extent, and
    after the first pass over the lengths no launch needs a fill pass."""
    B, D, H = 32, 64, 512
    g = torch.Generator().manual_seed(5)
    mk = lambda sc: tuple(p.to(DEV).requires_grad_(True) for p in (
        torch.randn(4 * H, D, generator=g) * sc / D ** 0.5, torch.randn(4 * H, H, generator=g) * sc / H ** 0.5,
        torch.randn(4 * H, generator=g) * 0.1, torch.randn(4 * H, generator=g) * 0.1))
    pf, pr = mk(1.0), mk(0.9)
    lengths = [40, 64, 23, 57, 64, 31, 40]
    xs = {T: torch.randn(T, B, D, generator=g).to(DEV) for T in set(lengths)}
    gys = {T: torch.randn(T, B, 2 * H, generator=g).to(DEV) for T in set(lengths)}

    def run(T):
        xg = xs[T].clone().requires_grad_(True)
        for p in pf + pr:
            p.grad = None
        y = ops.lstm_layer(xg, pf, pr)
        y.backward(gys[T])
        ops.join_deferred()
        return [y.detach().clone(), xg.grad.clone()] + [p.grad.clone() for p in pf + pr]

    def same(a, c):
        assert torch.equal(a[0], c[0])
        for u, v in zip(a[1:], c[1:]):
            assert float((u - v).abs().max()) <= 1e-5 * float(u.abs().max()) + 1e-12

    try:
        ops._XCHG_REARM = False
        ref = {T: run(T) for T in set(lengths)}
        ops.check_errors()
    finally:
        ops._XCHG_REARM = True
    ops.drop_exchange_pool()
    for T in lengths:
        same(run(T), ref[T])
    ops.check_errors()
    miss0 = ops.pool_stats()["exchange"]["miss"]
    for T in lengths:
        same(run(T), ref[T])
    ops.check_errors()
    assert ops.pool_stats()["exchange"]["miss"] == miss0, "a launch of an already seen length ran a fill pass"
    pooled = [e for lst in ops._xchg_pool["free"].values() for e in lst]
    assert 2 <= len(pooled) <= 4                              # forward + backward (+ at most one larger class each)
    for buf, armed, _ in pooled:
        assert bool((buf[:armed] == 0xFF).all()), "pooled exchange buffer not armed over its recorded extent"
    ops.drop_exchange_pool()


@pytest.mark.parametrize("xgrad", [True, False])
@pytest.mark.parametrize("T,B,H,pyr", [(64, 32, 1024, ("concat", 2)), (40, 32, 1024, None), (67, 32, 1024, ("concat", 2)),
                                       (24, 32, 512, ("concat", 2))])
def test_producer_written_panels_equal_split_passes(ops, T, B, H, pyr, xgrad):
    """Round 5: a bf16x6 recurrence launch stores its output (forward) / dG (BPTT) as the row-major split panel the next
    GEMM multiplies (asrk_lstm_rec_{fwd,bwd}_pyr_panel_f32) instead of leaving an f32 tensor for a split pass.  The
    panel holds the same three bf16 planes the split pass would write, so a two-layer stack gives BIT-IDENTICAL outputs
    with the feature on and off (the gradients pass through GEMMs whose split-K sums meet in atomics: 1e-5).  H = 1024:
    the second layer multiplies the first one's panel, its BPTT writes the dG panel, and both layers' BPTT write dG^T
    (the weight gradients' left operand) - xgrad = False makes the first layer the BOTTOM layer, whose two directions'
    weight-gradient GEMMs share that one panel across two streams; H = 512 (no stacked-direction GEMM to take a panel):
    emitted on request, dropped unused, same results."""
    g = torch.Generator().manual_seed(T + H)
    D = 256
    x = torch.randn(T, B, D, generator=g).to(DEV)
    mk = lambda din, sc: tuple(p.to(DEV).requires_grad_(True) for p in (
        torch.randn(4 * H, din, generator=g) * sc / din ** 0.5, torch.randn(4 * H, H, generator=g) * sc / H ** 0.5,
        torch.randn(4 * H, generator=g) * 0.1, torch.randn(4 * H, generator=g) * 0.1))
    d2 = 2 * H * (pyr[1] if pyr else 1)
    layers = [(mk(D, 1.0), mk(D, 0.9)), (mk(d2, 1.0), mk(d2, 0.9))]
    params = [p for lf, lr in layers for p in lf + lr]
    T2 = T // pyr[1] if pyr else T
    gy = torch.randn(T2, B, 2 * H, generator=g).to(DEV)

    def run(on):
        prev = ops._REC_PANELS
        ops._REC_PANELS = on
        try:
            for p in params:
                p.grad = None
            xg = x.clone().requires_grad_(xgrad)
            ops.set_panel_hint(True)                # what Encoder.forward says about a layer followed by another one
            h = ops.lstm_layer(xg, *layers[0], pyramid=(pyr[1], pyr[0]) if pyr else None)
            emitted = ops._panel_state["handover"] is not None
            ops.set_panel_hint(False)
            y = ops.lstm_layer(h, *layers[1])
            assert ops._panel_state["handover"] is None          # consumed (or nothing was emitted)
            y.backward(gy)
            ops.join_deferred()
            ops.check_errors()
            torch.cuda.synchronize()
            return emitted, [y.detach().clone()] + ([xg.grad.clone()] if xgrad else []) + [p.grad.clone() for p in params]
        finally:
            ops._REC_PANELS = prev
            ops.set_panel_hint(False)

    # round 6: a shape that is asked for the first time gets no panel (a new panel costs a zero fill over all of it:
    # only worth paying for shapes that come back); the second request allocates, later ones hit the pool
    ops._panel_pool["seen"].clear()
    ops._panel_pool["free"].clear()
    ops._panel_pool["bytes"] = 0
    em_first, first = run(True)
    assert not em_first
    st0 = dict(ops._panel_state["stats"])
    em1, a = run(True)
    st1 = dict(ops._panel_state["stats"])
    em0, b = run(False)
    assert em1 and not em0                                       # the bf16x6 plans of these widths do emit
    for u, v in zip(first, b):
        assert float((u - v).abs().max()) <= 1e-5 * float(v.abs().max()) + 1e-12
    assert dict(ops._panel_state["stats"]) == st1                # nothing emitted / consumed with the feature off
    if H >= 768:
        assert st1["consumed"] == st0["consumed"] + 1 and st1["dg"] == st0["dg"] + 1
        assert st1.get("dgt", 0) == st0.get("dgt", 0) + 2          # both layers' BPTT wrote their dG^T panel
    if H >= 768:
        assert torch.equal(a[0], b[0])                           # (H = 512: its f32-path GEMMs reduce split-K in atomics)
    for u, v in zip(a, b):
        assert float((u - v).abs().max()) <= 1e-5 * float(v.abs().max()) + 1e-12


# ------------------------------------------------------------------------------ top-k / arg-max
@pytest.mark.parametrize("rows,cols,k", [(5, 5000, 24), (1, 13, 13), (33, 257, 1), (16, 31, 4), (7, 8192, 64),
                                         (3, 5000, 80), (4, 9000, 16), (6, 600, 16)])
def test_topk_matches_stable_sort(ops, rows, cols, k):
    g = torch.Generator().manual_seed(rows * cols)
    x = torch.randn(rows, cols, generator=g)
    x[:, ::7] = x[:, 1:2]                       # plenty of exact ties
    if rows > 2:
        x[1] = -3.25                            # a constant row: the answer is the first k indices
        x[2, ::3] = 0.0
        x[2, 1::3] = -0.0                       # +0 and -0 tie (index order), above every negative value
        x[2, 2::3] = -x[2, 2::3].abs() - 1.0
    if cols == 600:                             # rows with fewer selectable (non-NaN) columns than k, -inf among them
        x[0, 5:] = float("nan")
        x[0, 2] = float("-inf")
        x[3, :] = float("nan")
        x[4, 100:] = float("-inf")
        v, i = ops.topk(t(x), k)
        v, i = v.cpu(), i.cpu()
        assert i[0, :5].tolist() == torch.sort(x[0, :5], descending=True, stable=True).indices.tolist()
        assert (i[0, 5:] == -1).all() and torch.isinf(v[0, 5:]).all() and (i[3] == -1).all()
        assert torch.equal(i[4], torch.sort(x[4], descending=True, stable=True).indices[:k])
        x = torch.nan_to_num(x, nan=-1e30)       # the remaining rows / the generic check below: no NaN
        x[0] = torch.randn(cols, generator=g)
        x[3] = torch.randn(cols, generator=g)
    vals, idx = ops.topk(t(x), k)
    order = torch.sort(x, dim=-1, descending=True, stable=True)      # ties -> smaller index first
    assert torch.equal(idx.cpu(), order.indices[:, :k])
    assert torch.equal(vals.cpu(), order.values[:, :k])
    assert torch.equal(ops.argmax(t(x)).cpu(), order.indices[:, 0])
    assert ops.argmax(t(x).view(rows, 1, cols)).shape == (rows, 1)


# ------------------------------------------------------------------------------ GRU layer / cell
# H % 4 == 0: the persistent kernels in GRU mode (csrc/lstm_rec.hip); H = 6: the per-step host loop.
# (24, 32, 80, 512) / (12, 32, 64, 1024) are the encoder widths of cfg2 / cfg3 (register-resident BPTT plan).
@pytest.mark.parametrize("T,B,D,H,bidir", [(9, 3, 7, 8, True), (17, 5, 12, 20, False), (6, 2, 5, 6, True),
                                           (1, 4, 6, 12, True), (24, 32, 80, 512, True),
                                           (12, 32, 64, 1024, False), (10, 70, 16, 64, True)])
def test_gru_layer_matches_oracle(ops, pkg, T, B, D, H, bidir):
    import importlib
    gru = importlib.import_module(pkg.__name__ + ".gru_ops")
    g = torch.Generator().manual_seed(T * 100 + H)
    names = ['weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0']
    shapes = [(3 * H, D), (3 * H, H), (3 * H,), (3 * H,)]
    sd = {}
    for sfx in ([''] + (['_reverse'] if bidir else [])):
        for n, s in zip(names, shapes):
            # wide layers: keep the recurrence out of the chaotic regime (else fp32 rounding is amplified)
            sd['l.' + n + sfx] = (torch.randn(*s, generator=g) * min(0.4, 1.5 / H ** 0.5)).requires_grad_(True)
    x = torch.randn(B, T, D, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = O.gru_layer(xr, sd, 'l.', bidir)                       # [B,T,ndir*H]
    dy = torch.randn(*yr.shape, generator=g)
    yr.backward(dy)
    dev = {k: v.detach().clone().to(DEV).requires_grad_(True) for k, v in sd.items()}
    xd = x.transpose(0, 1).contiguous().to(DEV).requires_grad_(True)          # time-major
    pf = tuple(dev['l.' + n] for n in names)
    pr = tuple(dev['l.' + n + '_reverse'] for n in names) if bidir else None
    y = gru.gru_layer(xd, pf, pr)
    y.backward(dy.transpose(0, 1).contiguous().to(DEV))
    assert rel_err(y.detach().cpu().transpose(0, 1), yr.detach()) < 1e-4
    assert rel_err(xd.grad.cpu().transpose(0, 1), xr.grad) < 1e-3
    for k in sd:
        assert rel_err(dev[k].grad.cpu(), sd[k].grad) < 1e-3, k


@pytest.mark.parametrize("style,rate,T", [("concat", 2, 11), ("drop", 2, 11), ("concat", 4, 16), ("drop", 3, 7)])
@pytest.mark.parametrize("bias", [True, False])
def test_gru_fused_time_reduction_matches_unfused(ops, pkg, style, rate, T, bias):
    """the reduced layout written by the recurrence kernel == PyramidFn on the plain output, forward and
    gradients (src/module.py:141-153), with and without biases"""
    import importlib
    gru = importlib.import_module(pkg.__name__ + ".gru_ops")
    B, D, H = 5, 9, 16
    g = torch.Generator().manual_seed(T + rate)
    shapes = [(3 * H, D), (3 * H, H), (3 * H,), (3 * H,)]

    def params():
        gg = torch.Generator().manual_seed(7)
        out = []
        for _ in range(2):
            p = [(torch.randn(*s, generator=gg) * 0.4).to(DEV).requires_grad_(True) for s in shapes]
            if not bias:
                p[2] = p[3] = None
            out.append(tuple(p))
        return out
    x = torch.randn(T, B, D, generator=g)
    outs = []
    for fused in (True, False):
        pf, pr = params()
        xd = x.clone().to(DEV).requires_grad_(True)
        if fused:
            y = gru.gru_layer(xd, pf, pr, pyramid=(rate, style))
        else:
            y = ops.pyramid(gru.gru_layer(xd, pf, pr), rate, style)
        gy = torch.Generator().manual_seed(3)
        y.backward(torch.randn(*y.shape, generator=gy).to(DEV))
        outs.append([y.detach().cpu(), xd.grad.cpu()] + [q.grad.cpu() for q in pf + pr if q is not None])
    for a, b in zip(*outs):
        assert a.shape == b.shape
        assert rel_err(a, b) < 1e-5


# ------------------------------------------------------------------------------ bf16x6 split GEMM
@pytest.fixture()
def split_mode(ops):
    """host-layer switch for the per-call ASRK_GEMM_SPLIT_* flag (ops.set_gemm_split); restored afterwards"""
    prev = ops.get_gemm_split()
    yield ops
    ops.set_gemm_split(prev)


@pytest.mark.parametrize("mode,M,N,K", [("NT", 128, 128, 32), ("NT", 300, 200, 70), ("NN", 257, 129, 100),
                                        ("TN", 130, 384, 517), ("NT", 64, 64, 8), ("NT", 1000, 1030, 1024),
                                        ("TN", 1024, 520, 3000), ("NN", 513, 1024, 2048),
                                        # 33 x 16 tiles of 128 x 256 = 2 rounds + 16 tiles on 256 CUs: the wide kernel
                                        # takes 32 row tiles, the last 128 (ragged: 100) rows run as 128 x 128 tiles
                                        ("NT", 4196, 4096, 264), ("NN", 4224, 3900, 512)])
def test_gemm_split_matches_float64(ops, split_mode, mode, M, N, K):
    """bf16x6 operand splitting (csrc/gemm_split.hip) against float64, beside the exact-f32 MFMA kernel:
    the split path must be as accurate as the f32 kernel (both a few 1e-7 of the row scale), with alpha,
    beta, both biases, ragged edges in M, N and K and every operand storage order."""
    g = torch.Generator().manual_seed(M + N + K)
    scale = torch.exp(torch.randn(K, generator=g) * 2.0)           # wide dynamic range along K
    if mode == "NT":
        A, B = torch.randn(M, K, generator=g) * scale, torch.randn(N, K, generator=g)
        a64, b64 = A.double(), B.double().t()
        args = (0, 1, M, N, K, t(A), K, t(B), K)
    elif mode == "NN":
        A, B = torch.randn(M, K, generator=g) * scale, torch.randn(K, N, generator=g)
        a64, b64 = A.double(), B.double()
        args = (0, 0, M, N, K, t(A), K, t(B), N)
    else:
        A, B = torch.randn(K, M, generator=g) * scale[:, None], torch.randn(K, N, generator=g)
        a64, b64 = A.double().t(), B.double()
        args = (1, 0, M, N, K, t(A), M, t(B), N)
    C0 = torch.randn(M, N + 3, generator=g)
    b1, b2 = torch.randn(N, generator=g), torch.randn(N, generator=g)
    ref = 0.75 * (a64 @ b64) + 0.5 * C0[:, :N].double() + b1.double() + b2.double()
    mag = (a64.abs() @ b64.abs()).max().item()                    # scale of the accumulated products
    errs = {}
    for split in (0, 2):
        split_mode.set_gemm_split(split)
        C = t(C0.clone())
        ops.gemm(*args, C, N + 3, alpha=0.75, beta=0.5, bias=t(b1), bias2=t(b2))
        assert torch.equal(C[:, N:].cpu(), C0[:, N:])              # nothing written beyond N
        errs[split] = (C[:, :N].cpu().double() - ref).abs().max().item() / mag
    # random-walk rounding of K f32 accumulations: a few 1e-8 * sqrt(K) of the product scale, either path
    bound = 1e-7 * max(4.0, K ** 0.5)
    assert errs[0] < bound and errs[2] < bound, errs


@pytest.mark.parametrize("env", [{"ASRK_SPLIT_W256": "2"}, {"ASRK_SPLIT_W256": "0"},
                                 {"ASRK_SPLIT_W256": "1", "ASRK_SPLIT_TAIL": "0"}])
def test_gemm_split_parity_with_each_kernel_forced(env):
    """The launch knobs are read once at asrk_init, so the variants that default routing does not pick for the shapes
    above - the 128 x 256 kernel on EVERY launch with N >= 512 (ragged N = 1030: clamped last row block; M = 513: a
    single, mostly empty row tile), the 128 x 128 kernel on the big shapes, the wide kernel with its half-empty last
    round - re-run the float64 parity tests in a process of their own."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_kernels_gpu.py"), "-q", "-x", "-m", "gpu",
                        "-p", "no:cacheprovider", "-k",
                        "test_gemm_split_matches_float64 or test_gemm_panels_ranges or test_gemm_split_propagates_nan"],
                       capture_output=True, text=True, env=dict(os.environ, **env), timeout=900,
                       cwd=os.path.dirname(here))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-500:])
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-500:]


def test_gemm_split_exact_on_bf16_representable_inputs(ops, split_mode):
    """inputs that are sums of <= 3 bf16 pieces with small integer products: the split path is exact"""
    g = torch.Generator().manual_seed(5)
    M, N, K = 256, 256, 64
    A = torch.randint(-300, 300, (M, K), generator=g).float()
    B = torch.randint(-300, 300, (N, K), generator=g).float()
    split_mode.set_gemm_split(2)
    C = t(torch.zeros(M, N))
    ops.gemm(0, 1, M, N, K, t(A), K, t(B), K, C, N)
    assert torch.equal(C.cpu().double(), A.double() @ B.double().t())


def test_gemm_split_propagates_nan(ops, split_mode):
    g = torch.Generator().manual_seed(6)
    A, B = torch.randn(256, 64, generator=g), torch.randn(256, 64, generator=g)
    A[3, 5] = float('nan')
    split_mode.set_gemm_split(2)
    C = t(torch.zeros(256, 256))
    ops.gemm(0, 1, 256, 256, 64, t(A), 64, t(B), 64, C, 256)
    c = C.cpu()
    assert torch.isnan(c[3]).all() and not torch.isnan(c[4]).any()


def test_gemm_panels_ranges_match_float64(ops, split_mode):
    """split once, multiply row / k ranges of the panels (the shared dG^T of the LSTM weight gradients): both
    storage orders, offsets in rows and k, ragged K that runs into one panel's zero padding"""
    g = torch.Generator().manual_seed(11)
    R, K, N = 512, 1000, 384
    A = torch.randn(K, R, generator=g)            # stored [K][rows] -> trans
    Bm = torch.randn(N, K, generator=g)           # stored [rows][K]
    pa = ops.SplitPanel(t(A), R, R, K, True)
    pb = ops.SplitPanel(t(Bm), K, N, K, False)
    a64, b64 = A.double().t(), Bm.double()
    cases = [(512, 384, 1000, 0, 0, 0, 0), (256, 128, 1000, 256, 0, 128, 0), (200, 384, 968, 128, 32, 0, 32),
             (128, 256, 960, 384, 40, 128, 0), (512, 384, 640, 0, 360, 0, 360)]
    for (M, Nn, Kk, ar, ak, br, bk) in cases:
        C = t(torch.zeros(M, Nn))
        ops.gemm_panels(M, Nn, Kk, pa, ar, ak, pb, br, bk, C, Nn)
        ref = a64[ar:ar + M, ak:ak + Kk] @ b64[br:br + Nn, bk:bk + Kk].t()
        err = (C.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 5e-6, (M, Nn, Kk, ar, ak, br, bk, err)
    with pytest.raises(Exception):                # ragged K that ends inside both panels
        ops.gemm_panels(128, 128, 100, pa, 0, 0, pb, 0, 0, t(torch.zeros(128, 128)), 128)
    with pytest.raises(Exception):                # row offset not a multiple of 128
        ops.gemm_panels(128, 128, 1000, pa, 64, 0, pb, 0, 0, t(torch.zeros(128, 128)), 128)


def test_lstm_wide_layer_with_input_gradient_vs_oracle(ops):
    """a wide BiLSTM layer whose input needs a gradient (every layer above the first): M = 2048 tokens, Din = 2048,
    8H = 8192 - every GEMM of the layer on the split path, weight gradients over the shared transposed panels"""
    _lstm_case(ops, 64, 32, 2048, 1024, True, seed=77)


def test_lstm_shared_panels_equal_separate_gemms(ops, monkeypatch):
    """weight gradients through ONE split of dG^T / Y^T / X^T (gemm_panels) == the per-GEMM splits"""
    T, B, D, H = 20, 32, 512, 1024
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(T, B, D, generator=g)
    shapes = ((4 * H, D), (4 * H, H), (4 * H,), (4 * H,))
    p0 = [torch.randn(*s, generator=g) / np.sqrt(s[-1] if len(s) > 1 else 4.0) for s in shapes * 2]
    dy = torch.randn(T, B, 2 * H, generator=g).to(DEV)
    grads = []
    for flag in ("1", "0"):
        monkeypatch.setenv("ASRK_SHARE_PANELS", flag)
        x = x0.clone().to(DEV).requires_grad_(True)
        ps = [q.clone().to(DEV).requires_grad_(True) for q in p0]
        ops.lstm_layer(x, tuple(ps[:4]), tuple(ps[4:])).backward(dy)
        ops.join_deferred()
        grads.append([q.grad.cpu() for q in ps])
    for a, b in zip(*grads):
        assert (a - b).abs().max().item() <= 2e-6 * max(1e-3, b.abs().max().item())
