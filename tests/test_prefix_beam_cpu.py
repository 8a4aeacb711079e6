"""CPU: the pure-CTC prefix beam search.

 * oracle/ctc_beam_oracle.py (restatement of /root/reference/src/ctc.py:118-352) is PINNED on hypotheses produced
   by running the real reference CTCBeamDecoder on given logits (tests/golden/ctcbeam_big.npz, recipe
   oracle/gen_golden.py --ctc-beam-big): ambiguous decimal-string keys, beam 20 x 30 candidates over V = 5000,
   finished (<eos>) hypotheses, LSTM and tied-GRU language models;
 * the DEVICE ALGORITHM (csrc/prefix_beam.inc - the source the gfx950 kernel is compiled from), built here for
   the host as a single thread (tests/native/prefix_beam_host.cpp), reproduces the same hypotheses, and agrees
   with the oracle on randomised searches designed to collide string keys.
No GPU, no HIP call."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import beam_oracle as BO
from oracle import ctc_beam_oracle as CBO
from oracle.gen_golden import CTC_BEAM_BIG, ctc_beam_big_logits

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "ctcbeam_big.npz")


@pytest.fixture(scope="module")
def pbh(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("pbh") / "libpbh.so")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", so,
                           os.path.join(HERE, "native", "prefix_beam_host.cpp")])
    L = ctypes.CDLL(so)
    L.pbh_new.restype = ctypes.c_void_p
    L.pbh_new.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p]
    L.pbh_frame.restype = ctypes.c_int
    L.pbh_frame.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_int,
                            ctypes.c_int]
    L.pbh_get.argtypes = [ctypes.c_void_p] * 6
    L.pbh_free.argtypes = [ctypes.c_void_p]
    return L


def _lm_fn(sd, cfg):
    """lm_step(token, hidden) -> (log-probs [V] float32 numpy, hidden) through the CPU LM restatement"""
    def step(token, hidden):
        with torch.no_grad():
            logits, hid = BO.lm_step(sd, cfg, torch.tensor([int(token)]), hidden)
            return torch.log_softmax(logits, dim=-1).squeeze().numpy(), hid
    return step


def _case(name):
    V, T, beam, cand, seed, hot, lm_cfg, lm_w = CTC_BEAM_BIG[name]
    g = np.load(GOLD)
    x = torch.log_softmax(ctc_beam_big_logits(name), dim=-1).numpy()          # what src/ctc.py:250 hands the loop
    lm = None
    if lm_cfg is not None:
        pre = name + ".lm."
        sd = {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}
        lm = _lm_fn(sd, lm_cfg)
    want = [g["%s.hyp%d" % (name, i)].tolist() for i in range(int(g[name + ".n"]))]
    return x, V, beam, cand, lm, lm_w, want


def run_host_algorithm(L, x, V, vocab_range, beam, cand, lm_step=None, lm_w=0.0):
    """drive the host build frame by frame exactly as src/ctc.py's device path drives the kernel"""
    T = x.shape[0]
    allowed = np.zeros(V, np.uint8)
    allowed[np.asarray(vocab_range)] = 1
    amax = x.argmax(-1)
    nz = np.nonzero(amax != 0)[0]
    if len(nz) == 0:
        return [[]]
    t_start = int(nz[0])
    h = L.pbh_new(beam, cand, V, T, allowed.ctypes.data)
    assert h
    apply_lm = lm_step is not None and lm_w > 0
    lm_rows = np.zeros((beam, V), np.float32)
    hid = [None] * beam
    if apply_lm:
        lm_rows[0], hid[0] = lm_step(0, None)
    parent, last, gidx = (np.zeros(beam, np.int32) for _ in range(3))
    lens, toks = np.zeros(beam, np.int32), np.zeros((beam, T + 1), np.int32)
    nb = 1
    for t in range(t_start, T):
        xr = np.ascontiguousarray(x[t], np.float32)
        follows = int(apply_lm and t < T - 1)
        nb = L.pbh_frame(h, xr.ctypes.data, lm_rows.ctypes.data if apply_lm else None, np.float32(lm_w),
                         int(t == T - 1), follows)
        if follows:
            L.pbh_get(h, lens.ctypes.data, toks.ctypes.data, parent.ctypes.data, last.ctypes.data, gidx.ctypes.data)
            new_rows, new_hid = lm_rows.copy(), list(hid)
            for r in range(nb):
                if gidx[r] >= beam:
                    new_rows[r], new_hid[r] = lm_step(int(last[r]), hid[parent[r]])
                else:
                    new_rows[r], new_hid[r] = lm_rows[gidx[r]], hid[gidx[r]]
            lm_rows, hid = new_rows, new_hid
    L.pbh_get(h, lens.ctypes.data, toks.ctypes.data, None, None, None)
    out = [toks[r, :lens[r]].tolist() for r in range(nb)]
    L.pbh_free(h)
    return out


@pytest.mark.parametrize("name", sorted(CTC_BEAM_BIG))
def test_oracle_is_pinned_on_the_real_reference(name):
    x, V, beam, cand, lm, lm_w, want = _case(name)
    got = CBO.prefix_beam_search(x, [1] + list(range(3, V)), beam, cand, lm_step=lm, lm_w=lm_w)
    assert got == want


@pytest.mark.parametrize("name", sorted(CTC_BEAM_BIG))
def test_device_algorithm_host_build_equals_reference(pbh, name):
    x, V, beam, cand, lm, lm_w, want = _case(name)
    got = run_host_algorithm(pbh, x, V, [1] + list(range(3, V)), beam, cand, lm_step=lm, lm_w=lm_w)
    assert got == want


@pytest.mark.parametrize("seed", range(48))
def test_device_algorithm_equals_oracle_on_colliding_random_searches(pbh, seed):
    """small vocabularies of symbols whose decimal strings concatenate ambiguously, random beam / candidate
    counts, frequent <eos>, all-blank prefixes; with a synthetic deterministic 'LM' on odd seeds"""
    rng = np.random.RandomState(seed)
    V = int(rng.choice([12, 40, 130, 1300]))
    T = int(rng.randint(1, 30))
    beam = int(rng.randint(1, 9))
    nv = V - 2
    cand = int(rng.randint(1, min(10, nv) + 1))
    logits = rng.randn(T, V).astype(np.float32) * 2.0
    logits[:, 0] += rng.choice([0.0, 2.0, 4.0])
    hot = [k for k in (1, 3, 4, 5, 11, 34, 45, 111, 345, 1111) if k < V]
    for k in hot:
        logits[:, k] += 2.5 * rng.rand(T)
    if rng.rand() < 0.5:
        logits[:int(rng.randint(0, 3)), 0] += 20.0
    if seed % 4 == 3:
        logits = np.round(logits)                           # exact score ties: stability of both sorts matters
    x = torch.log_softmax(torch.from_numpy(logits), dim=-1).numpy()
    lm, lm_w = None, 0.0
    if seed % 2 == 1:
        table = torch.log_softmax(torch.from_numpy(rng.randn(64, V).astype(np.float32) * 1.5), dim=-1).numpy()
        lm_w = float(rng.choice([0.3, 0.6]))

        def lm(token, hidden):                              # hidden = hash of the sequence so far
            hcode = (0 if hidden is None else hidden) * 31 + int(token) + 1
            return table[hcode % 64], hcode
    vr = [1] + list(range(3, V))
    want = CBO.prefix_beam_search(x, vr, beam, cand, lm_step=lm, lm_w=lm_w)
    got = run_host_algorithm(pbh, x, V, vr, beam, cand, lm_step=lm, lm_w=lm_w)
    assert got == want
