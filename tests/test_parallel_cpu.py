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
