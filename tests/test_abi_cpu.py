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
    assert L.asrk_version() >= 100
    assert lib.strerror(0) == "ok"
    assert "invalid" in lib.strerror(-1)
    assert "shape" in lib.strerror(-2)


def test_argument_errors_without_gpu(lib):
    L = lib.load()
    z = ctypes.c_void_p(0)
    # null pointers / unsupported combos are rejected before any HIP call
    assert L.asrk_gemm_f32(0, 1, 4, 4, 4, 1.0, z, 4, z, 4, 0.0, z, 4, z, z, 0, z) == -1
    assert L.asrk_gemm_f32(1, 1, 4, 4, 4, 1.0, z, 4, z, 4, 0.0, z, 4, z, z, 0, z) == -1
    assert L.asrk_gemm_f32(0, 1, 0, 4, 4, 1.0, z, 4, z, 4, 0.0, z, 4, z, z, 0, z) == 0  # empty
    assert L.asrk_lstm_rec_fwd_f32(z, z, z, z, z, 4, 2, 8, 2, z, 0, z, z) == -1
    assert L.asrk_lstm_rec_fwd_f32(z, z, z, z, z, 4, 2, 8, 3, z, 0, z, z) == -1
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
