"""Data-parallel engine: one process per GPU, gradients all-reduced over RCCL/xGMI
(torch.distributed backend "nccl" == RCCL on ROCm) in size-bounded buckets that are launched
from post-accumulate-grad hooks WHILE the rest of the backward pass is still running.

The reference has no distributed code at all; the correctness
contract is therefore "N shards == one process on the global batch" (SURVEY.md §8e):
  * buckets follow reverse parameter-registration order, so the CTC-head / decoder buckets are on
    the wire while the encoder BPTT (the long pole) is still computing;
  * reduction is an AVERAGE over ranks -> with equal shards it equals the global-batch gradient
    for mean-reduced per-utterance losses (CTC 'mean');
  * `token_normaliser()` all-reduces the non-pad token count so CrossEntropy(ignore_index=0,
    mean) can be normalised by the GLOBAL count (naive averaging of per-rank means is wrong when
    counts differ);
  * clip_grad_norm_ / the NaN-skip of src/solver.py:84-89 run AFTER `backward()` returns, i.e. on
    the reduced gradients, so every rank takes the same branch.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): buckets are kept large (default 32 MiB) so
each ring step is bandwidth- not latency-bound; 93 MB (cfg2) / 692 MB (cfg3) of fp32 gradients.
"""
import weakref

import torch

from . import ops


class DataParallelEngine:
    def __init__(self, model, dist, bucket_bytes=32 << 20, broadcast_params=True, force_collectives=False):
        """force_collectives: run hooks, buckets and every collective even in a world of ONE rank (RCCL with a
        single-rank communicator) - how the RCCL-facing code is exercised on a 1-GPU box (tests, ASRK_BENCH_FORCE_DIST)"""
        self.model = model
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None else 1
        self._collective = dist is not None and (self.world > 1 or force_collectives)
        self.params = [p for p in model.parameters() if p.requires_grad]
        self._buckets = []          # list of dicts: params, offsets, numel, flat, pending, work
        self._param_to_bucket = {}
        self._hooks = []
        self._build_buckets(bucket_bytes)
        if broadcast_params and self._collective:
            self.broadcast_parameters()
        for p in self.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad_ready))
        self._active = False
        self._slot_taken = set()
        # weight-gradient GEMMs write straight into the bucket slices (ops.grad_out): leaves are found by the address
        # of their data (the tensors autograd hands back to an op's backward are the parameters themselves)
        self._ptr_to_param = {p.data_ptr(): p for p in self.params if p.is_contiguous() and p.dtype == torch.float32}
        # exposed all-reduce time: hipEvents around the waits at the end of backward (the time the main stream had
        # nothing left to do but wait for the wire), one pair per step while `timing` is on
        self.timing = False
        self.wait_events = []
        # Buckets whose gradients are complete wait here until the next GEMM phase of the backward pass begins
        # (ops.on_gemm_phase: right after a persistent BPTT kernel was enqueued) or until backward() ends.  The
        # bf16x6 recurrence kernels own every CU (one workgroup each, 373-430 VGPRs per lane, ~140 KiB of LDS) and
        # cannot share a CU with a collective kernel; launching the all-reduce BEHIND the recurrence launch puts
        # it beside the dX / dW GEMMs instead (<= 168 MB = ~1.2 ms of ring time per layer inside a
        # GEMM phase of >= 6 ms), and neither kernel ever waits for CUs the other one is spinning on.
        self._ready = []
        # registered through a weak reference: the module-global hook list must not keep an engine (its model, its
        # bucket buffers) alive after its owner dropped it; a dead engine's entry removes itself on its next call
        wself = weakref.ref(self)

        def _phase_hook():
            eng = wself()
            if eng is None:
                ops.remove_gemm_phase_hook(_phase_hook)
            elif eng._active:
                eng._flush_ready()
        self._phase_hook = ops.on_gemm_phase(_phase_hook)

    # ------------------------------------------------------------------ setup
    def _build_buckets(self, bucket_bytes):
        cur, cur_bytes = [], 0
        for p in reversed(self.params):            # reverse registration ~ backward order
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > bucket_bytes:
                self._finish_bucket(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._finish_bucket(cur)

    def _finish_bucket(self, plist):
        offs, n = [], 0
        for p in plist:
            offs.append(n)
            n += p.numel()
        b = dict(params=plist, offsets=offs, numel=n, flat=None, pending=0, work=None,
                 index=len(self._buckets))
        for i, p in enumerate(plist):
            self._param_to_bucket[p] = (b, i)
        self._buckets.append(b)

    def broadcast_parameters(self, src=0):
        """identical model on every rank (same seed already gives this; broadcast makes it certain)"""
        for p in self.model.parameters():
            self.dist.broadcast(p.data, src=src)
        for buf in self.model.buffers():
            self.dist.broadcast(buf.data, src=src)

    # ------------------------------------------------------------------ per-step
    def _flat(self, b, like):
        if b["flat"] is None:
            b["flat"] = torch.empty(b["numel"], dtype=like.dtype, device=like.device)
        return b["flat"]

    def _grad_slot(self, weight, shape):
        """ops.grad_out's question: where does the gradient of `weight` go?  The parameter's slice of its bucket -
        while a backward pass of this engine is running, for a leaf that has no gradient yet (an existing .grad
        means accumulation, which needs a separate summand)."""
        if not self._active:
            return None
        p = self._ptr_to_param.get(weight.data_ptr())
        if p is None or p.grad is not None or tuple(p.shape) != tuple(shape) or p.device != weight.device:
            return None
        # at most ONE producer per parameter per backward: a weight applied several times in one graph (the per-step
        # decoder path: proj_q / merge_head / char_trans once per decode step) has several weight-gradient GEMMs, and
        # autograd sums their results only afterwards - the second and later ones must not land on the first one's
        # memory (p.grad stays None until AccumulateGrad has run, so it cannot tell them apart)
        if p in self._slot_taken:
            return None
        self._slot_taken.add(p)
        b, i = self._param_to_bucket[p]
        off = b["offsets"][i]
        return self._flat(b, p)[off:off + p.numel()].view(shape)

    def _on_grad_ready(self, p):
        if not self._active or not self._collective:
            return
        b, i = self._param_to_bucket[p]
        off = b["offsets"][i]
        dst = self._flat(b, p)[off:off + p.numel()]
        in_place = p.grad.data_ptr() == dst.data_ptr()     # the producing GEMM wrote the bucket slice itself
        # Weight-gradient GEMMs may still be running on ops' side stream (they overlap the next
        # layer's BPTT).  Waiting for them here would serialise exactly that overlap, so once the
        # side stream is in use the bucket copies and the all-reduce launches ride on it instead:
        # ordered after the deferred GEMMs (stream order) and after everything the main stream has
        # produced so far (event), while the main stream goes on with back-propagation.
        side = ops.deferred_stream(p.device) if p.is_cuda else None
        if side is None:
            if not in_place:
                dst.copy_(p.grad.reshape(-1))
                p.grad = dst.view_as(p)              # grad now lives in the bucket
            b["pending"] -= 1
            if b["pending"] == 0:
                self._bucket_ready(b, None)
            return
        main = torch.cuda.current_stream(p.device)
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        g = p.grad
        g.record_stream(side)
        with torch.cuda.stream(side):
            if not in_place:
                dst.copy_(g.reshape(-1))
                p.grad = dst.view_as(p)
            b["pending"] -= 1
            if b["pending"] == 0:
                self._bucket_ready(b, side)

    def _bucket_ready(self, b, stream):
        """all gradients of the bucket are in place (copies enqueued on `stream`, None = the current one)"""
        if b["flat"].is_cuda:
            self._ready.append((b, stream))          # launched at the next GEMM phase / the end of backward
        else:
            self._launch(b)

    def _flush_ready(self):
        """launch every complete bucket now: behind whatever the main stream has been given so far"""
        if not self._ready:
            return
        ready, self._ready = self._ready, []
        for b, stream in ready:
            if stream is None:
                self._launch(b)
            else:
                dev = b["flat"].device
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                stream.wait_event(ev)
                with torch.cuda.stream(stream):
                    self._launch(b)

    def _launch(self, b):
        # async all-reduce on RCCL's own stream; overlaps with the remaining backward kernels
        b["work"] = self.dist.all_reduce(b["flat"], op=self.dist.ReduceOp.SUM, async_op=True)

    def backward(self, loss):
        """loss.backward() with bucketed, overlapped gradient averaging across ranks."""
        if not self._collective:
            loss.backward()
            return
        for b in self._buckets:
            b["pending"] = len(b["params"])
            b["work"] = None
        self._ready = []
        self._slot_taken = set()
        self._active = True
        ops.set_grad_destination(self._grad_slot)
        try:
            # the 1/world of the average rides on the loss (exact for the power-of-two worlds of a node): the ranks'
            # gradients arrive pre-divided and the SUM all-reduce yields the mean - no pass over the buckets afterwards
            (loss * (1.0 / self.world) if self.world > 1 else loss).backward()
        except BaseException:
            # a failed backward leaves half-filled buckets behind: forget them (the next step starts clean) and let
            # the error surface; collectives already on the wire complete on their own stream
            self._ready = []
            for b in self._buckets:
                b["pending"], b["work"] = 0, None
            raise
        finally:
            self._active = False
            ops.set_grad_destination(None)
        # hook copies into the bucket buffers may have been issued on ops' side stream AFTER the event
        # the end-of-backward join waited for; the leftover buckets below are filled and launched from
        # the main stream, so order it behind everything the side stream has been given so far
        for dev, side in list(ops._defer["side"].items()):
            ev = torch.cuda.Event()
            ev.record(side)
            torch.cuda.current_stream(dev).wait_event(ev)
        self._flush_ready()                              # buckets completed after the last GEMM phase began
        for b in self._buckets:
            if b["pending"] > 0:
                # parameters that received no gradient this step (unused branch): zero-fill
                self._flat(b, b["params"][0])
                for i, p in enumerate(b["params"]):
                    off = b["offsets"][i]
                    if p.grad is None or p.grad.data_ptr() != b["flat"][off:].data_ptr():
                        if p.grad is None:
                            b["flat"][off:off + p.numel()].zero_()
                        else:
                            b["flat"][off:off + p.numel()].copy_(p.grad.reshape(-1))
                        p.grad = b["flat"][off:off + p.numel()].view_as(p)
                b["pending"] = 0
                self._launch(b)
        timed = self.timing and self._buckets and self._buckets[0]["flat"].is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for b in self._buckets:
            b["work"].wait()
        if timed:
            e1.record()
            self.wait_events.append((e0, e1))

    def exposed_allreduce_ms(self):
        """mean ms per step the main stream spent waiting for gradient all-reduces after its own work was done
        (call after a device synchronisation); None if nothing was timed"""
        if not self.wait_events:
            return None
        ms = [a.elapsed_time(b) for a, b in self.wait_events]
        self.wait_events = []
        return sum(ms) / len(ms)

    def count_normaliser(self, counts_local):
        """counts [k] of this rank (utterances, non-pad tokens, ...) -> global counts / world, float64 [k], in ONE
        all-reduce.  A per-rank MEAN loss times count_local / result, followed by gradient AVERAGING over ranks,
        is the mean over the global batch (SURVEY §8e cond. 1, 2)."""
        t = counts_local.detach().to(torch.float64).reshape(-1).clone()
        if self._collective:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t / self.world

    def token_normaliser(self, n_tok_local):
        """Global count of non-pad tokens / world (so that rank_loss = sum_CE / result, followed
        by gradient AVERAGING, equals CrossEntropy(mean over the global batch))."""
        return self.count_normaliser(n_tok_local.reshape(1)).to(torch.float32)

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        ops.remove_gemm_phase_hook(self._phase_hook)


# ---------------------------------------------------------------------------------- decode fan-out
# The reference fans utterances out over CPU worker processes (joblib.Parallel over the data loader,
# bin/test_asr.py:163-167).  Here one process drives one GPU: rank r decodes utterances r, r + world,
# r + 2*world, ... and rank 0 gathers the (name, hypotheses, truth) rows back into corpus order.
# Decoding has no data-path collective: utterances are independent ("replicas only").
def shard_indices(n_items, rank, world):
    """indices of the items rank `rank` of `world` processes handles (round-robin: lengths are
    correlated with corpus position, so contiguous blocks would be unbalanced)"""
    return list(range(rank, n_items, world))


def gather_in_order(local_items, n_items, dist, rank, world):
    """rank 0: the full list in original order; other ranks: None.  `local_items[k]` must be the result
    of item shard_indices(n_items, rank, world)[k]."""
    if world == 1 or dist is None:
        return list(local_items)
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(list(local_items), gathered, dst=0)
    if rank != 0:
        return None
    out = [None] * n_items
    for r, part in enumerate(gathered):
        idx = shard_indices(n_items, r, world)
        if len(part) != len(idx):
            raise RuntimeError("rank {} returned {} results for {} utterances".format(r, len(part), len(idx)))
        for i, item in zip(idx, part):
            out[i] = item
    return out
