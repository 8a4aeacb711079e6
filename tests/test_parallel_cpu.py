"""CPU, world_size 2 over gloo: the data-parallel engine (bucketed, hook-launched all-reduce)
reproduces the single-process global-batch gradients (SURVEY.md §8e exactness conditions)."""
import importlib
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "end-to-end-asr-pytorch_amd"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(12, 40), torch.nn.Tanh(), torch.nn.Linear(40, 40),
                               torch.nn.Tanh(), torch.nn.Linear(40, 7))


def _data():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(8, 12, generator=g)
    y = torch.randint(0, 7, (8,), generator=g)
    y[1] = 0          # ignore_index rows, unevenly spread over the two shards
    y[2] = 0
    y[3] = 0
    return x, y


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    par = importlib.import_module(PKG + ".parallel")
    model = _make_model()
    if rank == 1:  # broadcast_parameters must repair a diverged replica
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    eng = par.DataParallelEngine(model, dist, bucket_bytes=4096)   # several small buckets
    assert len(eng._buckets) >= 3
    x, y = _data()
    shard = slice(rank * 4, (rank + 1) * 4)
    logits = model(x[shard])
    # CrossEntropy(ignore_index=0, mean) normalised by the GLOBAL token count (§8e-2)
    n_tok = (y[shard] != 0).sum()
    loss_sum = torch.nn.functional.cross_entropy(logits, y[shard], ignore_index=0, reduction="sum")
    loss = loss_sum / eng.token_normaliser(n_tok)
    eng.backward(loss)
    grads = [p.grad.clone() for p in model.parameters()]
    # second step re-uses the buckets after zero_grad(set_to_none=True)
    for p in model.parameters():
        p.grad = None
    loss2 = torch.nn.functional.cross_entropy(model(x[shard]), y[shard], ignore_index=0,
                                              reduction="sum") / eng.token_normaliser(n_tok)
    eng.backward(loss2)
    grads2 = [p.grad.clone() for p in model.parameters()]
    if rank == 0:
        torch.save({"g1": grads, "g2": grads2}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_dp_engine_equals_global_batch(tmp_path):
    out = str(tmp_path / "g.pt")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    model = _make_model()
    x, y = _data()
    loss = torch.nn.functional.cross_entropy(model(x), y, ignore_index=0)   # global-batch mean
    loss.backward()
    for g1, g2, p in zip(got["g1"], got["g2"], model.parameters()):
        assert torch.allclose(g1, p.grad, atol=1e-6, rtol=1e-5)
        assert torch.allclose(g2, p.grad, atol=1e-6, rtol=1e-5)


# ---- world 4, ODD global batch (9 = 3 + 2 + 2 + 2 utterances), unequal token counts per rank (§8e cond. 1, 2, 5)
_SHARDS4 = [(0, 3), (3, 5), (5, 7), (7, 9)]


def _data9():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(9, 5, 12, generator=g)               # [utterance, position, feature]
    y = torch.randint(1, 7, (9, 5), generator=g)
    for u, n in enumerate([5, 1, 3, 4, 2, 5, 1, 1, 2]):  # valid tokens per utterance: 9 / 6 / 6 / 3 per rank
        y[u, n:] = 0
    return x, y


def _losses9(model, x, y, utt_norm, tok_norm):
    """per-utterance-mean term (the CTC 'mean' reduction: mean_b(nll_b / len_b), bin/train_asr.py:123) and
    a token-mean cross entropy with ignore_index=0 (bin/train_asr.py:130), each divided by the given normaliser"""
    logits = model(x)                                                        # [b, 5, 7]
    lens = (y != 0).sum(-1).clamp(min=1).float()
    per_utt = (logits.pow(2).sum(-1) * (y != 0)).sum(-1) / lens              # a stand-in nll_b / len_b
    ce_sum = torch.nn.functional.cross_entropy(logits.reshape(-1, 7), y.reshape(-1), ignore_index=0,
                                               reduction="sum")
    return per_utt.sum() / utt_norm + ce_sum / tok_norm


def _worker4(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    par = importlib.import_module(PKG + ".parallel")
    model = _make_model()
    eng = par.DataParallelEngine(model, dist, bucket_bytes=2048)
    x, y = _data9()
    lo, hi = _SHARDS4[rank]
    xs, ys = x[lo:hi], y[lo:hi]
    # global counts / world, so that gradient AVERAGING over ranks gives the global-batch means
    utt_norm = eng.token_normaliser(torch.tensor(float(hi - lo)))
    tok_norm = eng.token_normaliser((ys != 0).sum())
    assert abs(float(utt_norm) - 9 / 4) < 1e-6 and abs(float(tok_norm) - 24 / 4) < 1e-6
    eng.backward(_losses9(model, xs, ys, utt_norm, tok_norm))
    # every rank must hold the same reduced gradients (clip / NaN-skip decisions then agree, solver.py:84-89)
    flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    for g in gathered[1:]:
        assert torch.equal(g, gathered[0])
    if rank == 0:
        torch.save([p.grad.clone() for p in model.parameters()], out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_dp_engine_world4_odd_batch_unequal_tokens_equals_global_batch(tmp_path):
    out = str(tmp_path / "g4.pt")
    mp.spawn(_worker4, args=(4, _free_port(), out), nprocs=4, join=True)
    got = torch.load(out)
    model = _make_model()
    x, y = _data9()
    _losses9(model, x, y, 9.0, float((y != 0).sum())).backward()             # one process, the global batch
    for g, p in zip(got, model.parameters()):
        assert torch.allclose(g, p.grad, atol=1e-6, rtol=1e-5)


def test_single_process_engine_is_plain_backward():
    sys.path.insert(0, ROOT)
    par = importlib.import_module(PKG + ".parallel")
    model = _make_model()
    eng = par.DataParallelEngine(model, None)
    x, y = _data()
    loss = torch.nn.functional.cross_entropy(model(x), y, ignore_index=0)
    eng.backward(loss)
    ref = _make_model()
    torch.nn.functional.cross_entropy(ref(x), y, ignore_index=0).backward()
    for p, q in zip(model.parameters(), ref.parameters()):
        assert torch.allclose(p.grad, q.grad)


def test_forced_collectives_in_a_world_of_one_rank_equal_plain_backward():
    """force_collectives (bench.py ASRK_BENCH_FORCE_DIST, tests/test_parallel_gpu.py real-RCCL test): hooks, buckets,
    broadcast, token-count and gradient all-reduces all run through the process group even with ONE rank - here gloo;
    SUM over one rank is the identity, so the result is the plain backward"""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        par = importlib.import_module(PKG + ".parallel")
        model = _make_model()
        eng = par.DataParallelEngine(model, dist, bucket_bytes=4096, force_collectives=True)
        assert eng._collective and len(eng._buckets) >= 3
        plain = par.DataParallelEngine(_make_model(), dist)
        assert not plain._collective                    # a world of one stays a plain backward unless forced
        plain.remove_hooks()
        x, y = _data()
        n_tok = (y != 0).sum()
        assert float(eng.token_normaliser(n_tok)) == float(n_tok)
        for _ in range(2):                              # steady state reuses the bucket buffers
            for p in model.parameters():
                p.grad = None
            eng.backward(torch.nn.functional.cross_entropy(model(x), y, ignore_index=0))
            assert all(b["work"] is not None for b in eng._buckets)
        ref = _make_model()
        torch.nn.functional.cross_entropy(ref(x), y, ignore_index=0).backward()
        for p, q in zip(model.parameters(), ref.parameters()):
            assert torch.allclose(p.grad, q.grad)
        eng.remove_hooks()
    finally:
        if created:
            dist.destroy_process_group()


def _decode_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    par = importlib.import_module(PKG + ".parallel")
    n = 11                                                   # not a multiple of the world size
    mine = par.shard_indices(n, rank, world)
    local = [("utt%d" % i, [[i, i + 1], [i]], [i] * 3) for i in mine]      # (name, hyps, truth) rows
    merged = par.gather_in_order(local, n, dist, rank, world)
    if rank == 0:
        torch.save(merged, out)
    else:
        assert merged is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_decode_fanout_gathers_rows_in_corpus_order(tmp_path):
    """utterance-level sharding of decoding (reference fan-out: bin/test_asr.py:163-167): every
    utterance is decoded by exactly one rank and rank 0 gets the rows back in corpus order"""
    sys.path.insert(0, ROOT)
    par = importlib.import_module(PKG + ".parallel")
    for world in (1, 2, 3, 8):
        seen = sorted(i for r in range(world) for i in par.shard_indices(11, r, world))
        assert seen == list(range(11))
    out = str(tmp_path / "rows.pt")
    mp.spawn(_decode_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    rows = torch.load(out)
    assert [r[0] for r in rows] == ["utt%d" % i for i in range(11)]
    assert rows[4] == ("utt4", [[4, 5], [4]], [4, 4, 4])
    assert par.gather_in_order([1, 2, 3], 3, None, 0, 1) == [1, 2, 3]


def _solver_worker(rank, world, port, out):
    """drives the REAL BaseSolver.backward (src/solver.py) - engine backward, clip on the reduced
    gradients, NaN guard, optimizer step - on CPU tensors (the solver object is built without its
    GPU-only constructor)."""
    sys.path.insert(0, ROOT)
    import math
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    solver_mod = importlib.import_module(PKG + ".src.solver")
    optim_mod = importlib.import_module(PKG + ".src.optim")
    par = importlib.import_module(PKG + ".parallel")
    util = importlib.import_module(PKG + ".src.util")

    class Paras:
        verbose = False
    s = object.__new__(solver_mod.BaseSolver)
    s.paras, s.rank, s.world, s.dist, s.step = Paras(), rank, world, dist, 0
    s.GRAD_CLIP = 0.05                                    # small: the clip branch is taken
    s.timer = util.Timer()
    s.model = _make_model()
    s.optimizer = optim_mod.Optimizer(s.model.parameters(), "Adadelta", lr=1.0, eps=1e-8, lr_scheduler="fixed")
    s.dp = par.DataParallelEngine(s.model, dist, bucket_bytes=4096)
    x, y = _data()
    shard = slice(rank * 4, (rank + 1) * 4)
    log = {"norm": [], "params": []}
    for step in range(3):
        s.optimizer.pre_step(step)
        logits = s.model(x[shard])
        n_tok = (y[shard] != 0).sum()
        loss = torch.nn.functional.cross_entropy(logits, y[shard], ignore_index=0, reduction="sum") \
            / s.dp.token_normaliser(n_tok)
        if step == 1 and rank == 1:
            loss = loss * float("nan")                    # one rank diverges: EVERY rank must skip the step
        before = [p.detach().clone() for p in s.model.parameters()]
        gn = s.backward(loss)
        log["norm"].append(float(gn))
        log["params"].append([p.detach().clone() for p in s.model.parameters()])
        if step == 1:
            assert math.isnan(float(gn))
            assert all(torch.equal(a, b) for a, b in zip(before, s.model.parameters()))
    torch.save(log, out + ".%d" % rank)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_solver_backward_two_ranks_equals_global_batch_and_skips_nan_together(tmp_path):
    """SURVEY.md §8e conditions 2-3 on the real solver step (reference: src/solver.py:76-91): the
    2-rank run takes the same clipped Adadelta steps as one process on the global batch, and a NaN
    on one rank makes both ranks skip that step (the guard runs on the REDUCED gradients)."""
    import math
    out = str(tmp_path / "log.pt")
    mp.spawn(_solver_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    logs = [torch.load(out + ".%d" % r) for r in range(2)]
    for a, b in zip(logs[0]["params"], logs[1]["params"]):          # replicas stay identical
        assert all(torch.equal(p, q) for p, q in zip(a, b))
    assert logs[0]["norm"][0] == logs[1]["norm"][0] and math.isnan(logs[0]["norm"][1])
    # single process, global batch, torch's own clip + Adadelta; step 1 skipped
    model = _make_model()
    opt = torch.optim.Adadelta(model.parameters(), lr=1.0, eps=1e-8)
    x, y = _data()
    for step in range(3):
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(model(x), y, ignore_index=0)
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 0.05)
        if step != 1:
            opt.step()
        if step == 0:
            assert abs(float(gn) - logs[0]["norm"][0]) < 1e-5 * float(gn)
        for p, q in zip(model.parameters(), logs[0]["params"][step]):
            assert torch.allclose(p, q, atol=1e-6, rtol=1e-5), step
