"""CPU: libasrk.so builds, loads and exports every symbol include/asrk.h declares; host-side
argument validation that needs no GPU."""
import ctypes
import importlib
import os
import re

import pytest

from conftest import PKG_NAME, ROOT


@pytest.fixture(scope="module")
def lib():
    build = importlib.import_module(PKG_NAME + ".build")
    build.build(verbose=False)
    return importlib.import_module(PKG_NAME + "._lib")


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "asrk.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(asrk_[a-z0-9_]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    L = lib.load()
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), "libasrk.so does not export " + s
        assert s in lib.SIGNATURES, "no ctypes signature for " + s
    for s in lib.SIGNATURES:
        assert s in syms, "ctypes binds %s which include/asrk.h does not declare" % s


def test_version_and_strerror(lib):
    L = lib.load()
    assert L.asrk_version() >= 200
    assert lib.strerror(0) == "ok"
    assert "invalid" in lib.strerror(-1)
    assert "shape" in lib.strerror(-2)


def test_argument_errors_without_gpu(lib):
    L = lib.load()
    z = ctypes.c_void_p(0)
    # null pointers / unsupported combos are rejected before any HIP call
    assert L.asrk_gemm_f32(0, 1, 4, 4, 4, 1.0, z, 4, z, 4, 0.0, z, 4, z, z, 0, 0, z, 0, z) == -1
    assert L.asrk_gemm_f32(1, 1, 4, 4, 4, 1.0, z, 4, z, 4, 0.0, z, 4, z, z, 0, 0, z, 0, z) == -1
    assert L.asrk_gemm_f32(0, 1, 0, 4, 4, 1.0, z, 4, z, 4, 0.0, z, 4, z, z, 0, 0, z, 0, z) == 0  # empty
    assert L.asrk_gemm_f32(0, 1, 4, 4, 4, 1.0, z, 4, z, 4, 0.0, z, 4, z, z, 0, -1, z, 0, z) == -1  # bad flags
    # a contraction that takes the split path needs the caller's workspace: ASRK_EWORKSPACE, never a hidden
    # allocation (the library owns no device memory)
    fake = ctypes.c_void_p(4096)
    need = L.asrk_gemm_ws_bytes(1024, 1024, 1024, 2)
    assert need > 0 and L.asrk_gemm_ws_bytes(1024, 1024, 1024, 1) == 0
    assert L.asrk_gemm_f32(0, 1, 1024, 1024, 1024, 1.0, fake, 1024, fake, 1024, 0.0, fake, 1024, z, z, 0, 2,
                           z, 0, z) == -3
    assert L.asrk_gemm_f32(0, 1, 1024, 1024, 1024, 1.0, fake, 1024, fake, 1024, 0.0, fake, 1024, z, z, 0, 2,
                           fake, need - 1, z) == -3
    assert L.asrk_lstm_rec_fwd_f32(z, z, z, z, z, 4, 2, 8, 2, z, 0, z, 0, z) == -1
    assert L.asrk_lstm_rec_fwd_f32(z, z, z, z, z, 4, 2, 8, 3, z, 0, z, 0, z) == -1
    assert L.asrk_lstm_rec_fwd_f32(z, z, z, z, z, 4, 2, 8, 2, z, 0, z, -1, z) == -1
    assert L.asrk_log_softmax_fwd_f32(z, z, 1, 0, 0, z) == -1
    assert L.asrk_ctc_loss_fwd_f32(z, 0, 0, 1, 1, 0, z, 0, 0, z, z, 0, z, z, z, z, z) == -1
    assert L.asrk_lstm_ws_bytes() >= 4096


def test_product_path_fails_loudly_without_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ops = importlib.import_module(PKG_NAME + ".ops")
    with pytest.raises(RuntimeError):
        ops.linear(torch.zeros(2, 4), torch.zeros(3, 4), None)
    with pytest.raises(RuntimeError):
        ops.log_softmax(torch.zeros(2, 4))


def test_split_panel_geometry_and_argument_checks_need_no_gpu():
    """pure host logic of the split-panel API (include/asrk.h): sizes, routing predicate, argument validation
    happen before any device call"""
    import ctypes
    import importlib
    lib = importlib.import_module("end-to-end-asr-pytorch_amd._lib").load()
    b1 = lib.asrk_split_panel_bytes(128, 32, 0)
    # 2 row blocks x (1 k-tile + 1 spare) x 4 chunk columns x 3 planes x 1 KiB (+ channel-spreading pad)
    assert b1 >= 2 * 2 * 4 * 3 * 1024 and b1 % 16 == 0
    assert lib.asrk_split_panel_bytes(129, 32, 0) > b1 and lib.asrk_split_panel_bytes(128, 33, 0) > b1
    assert lib.asrk_split_panel_bytes(0, 32, 0) == 0
    AUTO, OFF, ALWAYS = 0, 1, 2                                      # ASRK_GEMM_SPLIT_* (per-call flags)
    assert lib.asrk_split_panel_bytes(128, 32, 0) > 0 and lib.asrk_split_panel_bytes(128, 32, 4) == 0   # flags: reserved
    assert lib.asrk_gemm_takes_split(25600, 8192, 4096, AUTO) == 1   # cfg3 layer-1 input projection
    assert lib.asrk_gemm_takes_split(32, 4096, 3072, AUTO) == 0      # decoder cell: skinny path
    assert lib.asrk_gemm_takes_split(8192, 80, 51200, AUTO) == 0     # layer-0 weight gradient: N = 80
    assert lib.asrk_gemm_takes_split(25600, 8192, 4096, OFF) == 0
    assert lib.asrk_gemm_takes_split(32, 4096, 3072, ALWAYS) == 1
    assert lib.asrk_gemm_takes_split(25600, 8192, 4096, AUTO | (96 << 8)) == 1   # LDS hint bits do not matter
    # workspace = both operands' panels (10 B per padded element: 3 bf16 planes + row-block padding)
    w = lib.asrk_gemm_ws_bytes(25600, 8192, 4096, AUTO)
    assert w >= (25600 + 8192) * 4096 * 6 and w < (25600 + 8192) * 4096 * 7
    assert lib.asrk_gemm_ws_bytes(25600, 8192, 4096, OFF) == 0
    fake = ctypes.c_void_p(4096)
    ok_args = [256, 256, 64, 1.0, fake, 256, 64, 0, 0, fake, 256, 64, 0, 0, 0.0, fake, 256, None, None, 0, None]
    for pos, bad in ((7, 64), (8, 4), (12, 100), (16, 8), (0, 300)):   # row offset, k offset, b row offset, ldc, M
        a = list(ok_args)
        a[pos] = bad
        assert lib.asrk_gemm_panels_f32(*a) != 0
    a = list(ok_args)
    a[2], a[6], a[11] = 40, 64, 64                                   # ragged K ending inside both panels
    assert lib.asrk_gemm_panels_f32(*a) != 0
