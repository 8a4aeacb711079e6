"""Operator layer: torch.autograd.Functions over the libasrk C ABI.

PyTorch is used for device memory (caching allocator), the current HIP stream and autograd
bookkeeping only; every tensor that reaches this module is turned into a raw device pointer and
handed to a hand-written gfx950 kernel.  Nothing here falls back to ATen math.
"""
import ctypes

import torch
from torch.autograd import Function

from . import _lib

_ws_cache = {}


def _L():
    return _lib.load()


def _require_gpu(t):
    if not t.is_cuda:
        raise _lib.AsrkError("asrk ops need tensors on the MI355X (got %s); the product path has no "
                             "CPU fallback" % t.device)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """the current torch stream of the current device as a raw hipStream_t.  torch.cuda.current_stream() builds a Stream
    object through several Python layers (~9 us; ~40 launches per decode position: a fifth of a one-utterance position);
    the raw query is the same lookup without the object."""
    if _raw_stream is not None and _raw_device is not None:
        return ctypes.c_void_p(_raw_stream(_raw_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t):
    """contiguous float32 view/copy (copy only when the caller handed something exotic)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def lstm_workspace(device):
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    ws = _ws_cache.get(key)
    if ws is None:
        ws = torch.zeros(int(_L().asrk_lstm_ws_bytes()), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


# ---- per-call arithmetic / launch flags (include/asrk.h).  The library keeps no mode state; the HOST layer
# does, exactly like torch.backends.* switches: the split mode starts from ASRK_GEMM_SPLIT (0 = f32-input
# MFMA everywhere, 1 = exact bf16x6 operand splitting where it pays (default), 2 = wherever the shape
# allows) and bench.py / the tests flip it with set_gemm_split(); ASRK_REC_BF / ASRK_REC_BF_BWD = 0 select the
# f32-MFMA recurrence kernels.
import os as _os
_GEMM_SPLIT_FLAG = {0: 1, 1: 0, 2: 2}          # host mode -> ASRK_GEMM_SPLIT_{OFF, AUTO, ALWAYS}
_gemm_state = {"split": max(0, min(2, int(_os.environ.get("ASRK_GEMM_SPLIT", "1")))), "lds_hint": 0}


def set_gemm_split(mode):
    """0 = never split (v_mfma_f32_32x32x2_f32 everywhere), 1 = where it pays (default), 2 = always"""
    _gemm_state["split"] = max(0, min(2, int(mode)))


def get_gemm_split():
    return _gemm_state["split"]


def gemm_flags():
    return _GEMM_SPLIT_FLAG[_gemm_state["split"]] | ((_gemm_state["lds_hint"] & 0xff) << 8)


def gemm_takes_split(M, N, K):
    return bool(_L().asrk_gemm_takes_split(M, N, K, gemm_flags()))


def rec_flags(backward):
    """ASRK_REC_F32_MFMA when the environment asks for the f32-MFMA recurrence (read per call: the tests and
    bench.py's exact-f32 comparison toggle it)"""
    return 1 if _os.environ.get("ASRK_REC_BF_BWD" if backward else "ASRK_REC_BF", "1") == "0" else 0


ASRK_REC_REARM = 2            # include/asrk.h: the launch hands the exchange buffer back sentinel-filled
_XCHG_REARM = _os.environ.get("ASRK_XCHG_REARM", "1") != "0"
_XCHG_POOL_CAP = int(float(_os.environ.get("ASRK_XCHG_POOL_GB", "24")) * (1 << 30))
# free[(device, stream)] = [[buffer, armed_bytes, last_use_tick], ...]; "stats" counts launches served without / with a
# fill pass (bench.py reports them)
_xchg_pool = {"free": {}, "bytes": 0, "tick": 0, "stats": {"hit": 0, "miss": 0, "evicted": 0}}


def _size_class(n):
    """n rounded up to four classes per octave (<= 25 % slack): utterance lengths differ from batch to batch, and a
    pool keyed by the exact footprint would miss on almost every launch of a real epoch"""
    if n <= 4096:
        return 4096
    q = 1 << (int(n - 1).bit_length() - 3)
    return (n + q - 1) // q * q


def _pool_evict(pool, need):
    """drop least-recently-used free buffers until `need` more bytes fit under the cap"""
    while pool["bytes"] + need > _XCHG_POOL_CAP:
        victim = None
        for key, lst in pool["free"].items():
            for i, e in enumerate(lst):
                if victim is None or e[2] < victim[2][2]:
                    victim = (key, i, e)
        if victim is None:
            return
        key, i, e = victim
        pool["free"][key].pop(i)
        pool["bytes"] -= e[0].numel()
        pool["stats"]["evicted"] += 1


class _Exchange:
    """Scratch for the in-kernel h / dG exchange of ONE recurrence launch.

    The kernels need it pre-filled with a NaN sentinel.  Rounds 1-4 took stream-ordered scratch and let every launch
    fill it first (eight fills per cfg3 step: 1.0 ms of stores in front of latency-bound kernels).  Now the launch is
    asked to hand the buffer back ARMED (ASRK_REC_REARM: the workgroups refill the region of step s - 2 while they
    compute step s) and the buffer goes into a pool.  An armed buffer is all 0xFF bytes whatever layout the launch that
    armed it used, so the pool is keyed by (device, stream) only and any free buffer that is large enough serves any
    shape: buffers come in size classes (four per octave), each remembers how many of its leading bytes have ever been
    armed (a launch that needs more than that runs its fill pass once and extends the extent), and the pool is bounded
    by evicting least-recently-used buffers (ASRK_XCHG_POOL_GB, default 24 of 288).  ASRK_XCHG_REARM=0 restores
    fill-per-launch.  A hand-off timeout leaves buffers dirty: check_errors() drops the pool before it raises.

        x = _Exchange(L, T, B, H, ndir, backward, device)
        check(L.asrk_..._rec_...(..., _p(x.buf), x.prefilled, ..., x.flags, stream));  x.done()"""

    __slots__ = ("buf", "prefilled", "flags", "key", "need", "armed")

    def __init__(self, L, T, B, H, ndir, backward, device):
        base = rec_flags(backward)
        n = int(L.asrk_lstm_xchg_bytes(T, B, H, ndir, backward, base))
        if n == 0:
            raise _lib.AsrkError("LSTM shape T=%d B=%d H=%d ndir=%d is not supported by the persistent "
                                 "gfx950 recurrence kernels" % (T, B, H, ndir))
        self.flags = base | (ASRK_REC_REARM if _XCHG_REARM else 0)
        self.need = n
        self.key = (device.index, torch.cuda.current_stream(device).cuda_stream)
        if not _XCHG_REARM:
            self.buf, self.prefilled, self.armed = torch.empty(n, dtype=torch.uint8, device=device), 0, 0
            return
        free = _xchg_pool["free"].get(self.key, ())
        best = None
        for i, e in enumerate(free):       # the smallest buffer that is big enough; one armed far enough first
            if e[0].numel() >= n:
                rank = (e[1] < n, e[0].numel())
                if best is None or rank < best[0]:
                    best = (rank, i)
        if best is not None:
            e = free.pop(best[1])
            _xchg_pool["bytes"] -= e[0].numel()
            self.buf, self.armed = e[0], e[1]
        else:
            self.buf, self.armed = torch.empty(_size_class(n), dtype=torch.uint8, device=device), 0
        self.prefilled = 1 if self.armed >= n else 0
        _xchg_pool["stats"]["hit" if self.prefilled else "miss"] += 1

    def done(self):
        """the launch is enqueued: the first max(need, armed) bytes are armed for whoever comes next on this stream"""
        if not _XCHG_REARM:
            return
        n = self.buf.numel()
        if n > _XCHG_POOL_CAP:
            return
        _pool_evict(_xchg_pool, n)
        _xchg_pool["tick"] += 1
        _xchg_pool["free"].setdefault(self.key, []).append([self.buf, max(self.armed, self.need), _xchg_pool["tick"]])
        _xchg_pool["bytes"] += n


def drop_exchange_pool():
    _xchg_pool["free"].clear()
    _xchg_pool["bytes"] = 0


def pool_stats():
    """launches served from the exchange pool without / with a fill pass, panels emitted from the pool / skipped"""
    return {"exchange": dict(_xchg_pool["stats"], held_bytes=_xchg_pool["bytes"]),
            "panels": dict(_panel_pool["stats"], held_bytes=_panel_pool["bytes"])}


def check_errors(device=None):
    """Synchronises the current stream and raises if a persistent kernel's grid sync timed out."""
    for key, ws in list(_ws_cache.items()):
        rc = _L().asrk_lstm_check_error(_p(ws), _stream())
        if rc != 0:
            drop_exchange_pool()        # an aborted launch did not re-arm its exchange buffer
        _lib.check(rc, "lstm grid sync")


# --------------------------------------------------------------------------- deferred weight gradients
# Weight-gradient GEMMs (dW = dY^T X) are off the critical path of back-propagation: nothing
# downstream in the backward pass reads them.  They are therefore launched on a second HIP stream
# and run on the CUs the latency-bound kernels on the main stream leave idle (the persistent BPTT
# kernel of a 512-unit layer occupies 128 of 256 CUs; the attention-decoder loop far fewer).  The
# main stream re-joins the side stream at the end of the backward pass (autograd engine callback),
# so every consumer after `loss.backward()` sees finished gradients.  Deferral is only used when the
# gradient's consumer is a plain AccumulateGrad into an empty `.grad` (leaf weight, grad None).
_defer = {"enabled": True, "side": {}, "pending": {}, "cb_armed": False, "release": []}


def set_deferred_weight_grads(flag):
    """Enable/disable launching weight-gradient GEMMs on the side stream (default: enabled)."""
    join_deferred()
    _defer["enabled"] = bool(flag)


def join_deferred():
    """Make the current stream wait for all deferred weight-gradient work (no host sync)."""
    for dev, ev in list(_defer["pending"].items()):
        torch.cuda.current_stream(dev).wait_event(ev)
    _defer["pending"].clear()


def deferred_stream(device):
    """the side stream if weight-gradient work has been deferred onto it in this backward pass (so a
    gradient consumer can order itself after that work without stalling the main stream), else None"""
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device(dev.type, torch.cuda.current_device())
    return _defer["side"].get(dev) if dev in _defer["pending"] else None


def _end_of_backward():
    _defer["cb_armed"] = False
    join_deferred()
    rel, _defer["release"] = _defer["release"], []
    for pnl in rel:                  # pooled panels that side-stream GEMMs were still reading
        pnl.release()


def _can_defer(*weights):
    return _defer["enabled"] and all(w is None or (w.is_leaf and w.grad is None) for w in weights)


# LDS KiB requested by side-stream GEMMs (> 80 = one workgroup per CU: they hand CUs back sooner to the
# main stream's BPTT / dX kernels; 27.5 -> 25.9 ms/step at cfg2).  Only useful when the BPTT kernels
# leave CUs free (H=512 plans use 128 of 256); when they fill the chip (H=1024) the hint just slows
# the background work (cfg3 229 -> 233 ms), so it follows the last BPTT plan seen.  ASRK_SIDE_BG
# forces a value (0 = never).
_BG_FORCED = _os.environ.get("ASRK_SIDE_BG")
_BG_HINT = int(_BG_FORCED) if _BG_FORCED is not None else 0


# ---- phase hooks: callables run right after a persistent BPTT kernel has been enqueued on the main stream, i.e.
# at the start of that layer's GEMM phase (dX / dW).  parallel.DataParallelEngine uses it to launch its ready
# gradient buckets THERE: ordered behind the recurrence kernel (which owns every CU and cannot share one with a
# collective kernel) and beside the GEMMs that follow, instead of racing the next BPTT launch.
_gemm_phase_hooks = []


def on_gemm_phase(fn):
    _gemm_phase_hooks.append(fn)
    return fn


def remove_gemm_phase_hook(fn):
    if fn in _gemm_phase_hooks:
        _gemm_phase_hooks.remove(fn)


def _gemm_phase_begins():
    for fn in list(_gemm_phase_hooks):
        fn()


def _defer_beside_bptt():
    """Weight-gradient GEMMs of a layer may go to the side stream only if the BPTT kernel that follows them on
    the main stream leaves CUs free (H = 512 plans: 128 of 256).  The bf16x6 plans of the wide layers own every
    CU (one workgroup each, 132-143 KiB of LDS): a GEMM launched beside them is not overlapped but PARKED - its
    remaining tiles wait for the whole recurrence launch while the recurrence's workgroups wait for the CUs the
    GEMM still holds (profiles/r02_cfg3_step_timeline.log) - so there the GEMMs run in stream order.
    ASRK_DEFER_WIDE=1 restores the old behaviour for A/B."""
    return _BG_HINT > 0 or _os.environ.get("ASRK_DEFER_WIDE", "0") == "1"


def _note_bptt_plan(L, T, B, H, ndir):
    global _BG_HINT
    if _BG_FORCED is None:
        wgs = int(L.asrk_lstm_plan_workgroups(T, B, H, ndir, 1, rec_flags(1)))
        cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
        _BG_HINT = 96 if 0 < 2 * wgs <= cus else 0


class _SideStream:
    """with _SideStream(device, inputs): kernels launched inside run on the side stream, ordered after
    everything already enqueued on the main stream; `inputs` (main-stream allocations read inside)
    are protected from early reuse by the caching allocator."""

    def __init__(self, device, inputs, background=True):
        self.device = torch.device(device)
        self.inputs = inputs
        self.hint = _BG_HINT if background else 0      # background: something latency-critical follows

    def __enter__(self):
        dev = self.device
        side = _defer["side"].get(dev)
        if side is None:
            side = torch.cuda.Stream(device=dev)
            _defer["side"][dev] = side
        main = torch.cuda.current_stream(dev)
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        for t in self.inputs:
            if t is not None:
                t.record_stream(side)
        self.main, self.side = main, side
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        self.prev_hint = _gemm_state["lds_hint"]
        if self.hint:
            _gemm_state["lds_hint"] = self.hint
        return self

    def keep(self, *outs):
        """outputs allocated on the side stream that the main stream will consume after the join"""
        for t in outs:
            if t is not None:
                t.record_stream(self.main)

    def __exit__(self, *exc):
        _gemm_state["lds_hint"] = self.prev_hint
        self.ctx.__exit__(*exc)
        done = torch.cuda.Event()
        done.record(self.side)
        _defer["pending"][self.device] = done
        if not _defer["cb_armed"]:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
                _defer["cb_armed"] = True
            except RuntimeError:        # not inside a backward pass: join right away
                join_deferred()
        return False


# ---- gradient destinations.  A data-parallel engine reduces gradients in flat bucket buffers (parallel.py); an op
# that produces the gradient of a LEAF weight asks here where to write it, so the weight-gradient GEMM's output IS the
# bucket slice and no copy into the bucket is needed (692 MB per cfg3 step).  Without an engine: plain scratch.
_grad_dest = {"fn": None}


def set_grad_destination(fn):
    """fn(weight, shape) -> f32 tensor of `shape` to write that leaf's gradient into, or None; None unregisters"""
    _grad_dest["fn"] = fn


def grad_out(weight, shape, device):
    fn = _grad_dest["fn"]
    if fn is not None and weight is not None:
        t = fn(weight, tuple(shape))
        if t is not None:
            return t
    return torch.empty(shape, dtype=torch.float32, device=device)


def grad_has_destination(*weights):
    fn = _grad_dest["fn"]
    return fn is not None and all(w is not None and fn(w, tuple(w.shape)) is not None for w in weights)


# --------------------------------------------------------------------------- raw wrappers
def gemm(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, alpha=1.0, beta=0.0, bias=None, bias2=None,
         splitk=0):
    _require_gpu(C)
    # operand extents must cover what the kernel will touch (A: M x K, B: K x N as stored, C: M x N)
    ra, ca = (K, M) if transA else (M, K)
    rb, cb = (N, K) if transB else (K, N)
    for name, t, r, c, ld in (("A", A, ra, ca, lda), ("B", B, rb, cb, ldb), ("C", C, M, N, ldc)):
        avail = t.untyped_storage().nbytes() // t.element_size() - t.storage_offset()
        if r > 0 and c > 0 and (ld < c or avail < (r - 1) * ld + c):
            raise _lib.AsrkError("gemm: operand {} ({} elements behind the pointer) too small for {}x{} "
                                 "with ld {}".format(name, avail, r, c, ld))
    L = _L()
    flags = gemm_flags()
    # split panels of both operands: scratch from torch's caching allocator (stream-ordered, so the block is
    # safe to hand out again as soon as this call's kernels are enqueued)
    nws = int(L.asrk_gemm_ws_bytes(M, N, K, flags)) if splitk <= 0 else 0
    ws = torch.empty((nws,), dtype=torch.uint8, device=C.device) if nws else None
    _lib.check(L.asrk_gemm_f32(int(transA), int(transB), M, N, K, alpha, _p(A), lda, _p(B), ldb,
                               beta, _p(C), ldc, _p(bias), _p(bias2), splitk, flags, _p(ws), nws, _stream()),
               "gemm")


class SplitPanel:
    """bf16x3 split panel of a logical [rows][K] f32 operand (csrc/gemm_split.hip): built once, usable as
    either operand of several gemm_panels calls.  trans: `src` is stored [K][rows] with leading dimension ld."""

    def __init__(self, src, ld, rows, K, trans):
        _require_gpu(src)
        L = _L()
        need = (K - 1) * ld + rows if trans else (rows - 1) * ld + K
        avail = src.untyped_storage().nbytes() // src.element_size() - src.storage_offset()
        if ld < (rows if trans else K) or avail < need:
            raise _lib.AsrkError("split panel: source too small for {}x{} (trans={}) with ld {}".format(
                rows, K, trans, ld))
        self.rows, self.K = rows, K
        self.flags = 0                            # reserved (include/asrk.h)
        self.buf = torch.empty((L.asrk_split_panel_bytes(rows, K, self.flags),), dtype=torch.uint8,
                               device=src.device)
        _lib.check(L.asrk_split_panel_f32(_p(src), ld, rows, K, int(trans), _p(self.buf), self.flags, _stream()),
                   "split_panel")


# ---- panels written by their producers (round 5).  A recurrence launch can store its output (forward) / dG (BPTT) as
# the row-major split panel the GEMM that follows multiplies, so that GEMM needs no split pass over the tensor: 183 M + 367 M
# elements x 10 B per cfg3 step.  The buffers are pooled (zero-initialised ONCE: the kernels overwrite the real extent, the
# padding stays zero) per (device, stream, rows, K).  A forward panel travels from the layer that wrote it to the layer that
# reads it through a one-slot hand-over keyed by the IDENTITY of the output tensor (weak reference + version counter):
# whatever sits between two layers (LayerNorm, dropout, projection) makes a new tensor and the consumer falls back to
# splitting its input itself.  ASRK_REC_PANELS=0 turns both off.
import weakref as _weakref
_REC_PANELS = _os.environ.get("ASRK_REC_PANELS", "1") != "0"
# free[(device, stream, rows, K)] = [[buffer, 0, last_use_tick], ...]; "seen" = shapes asked for before
_panel_pool = {"free": {}, "bytes": 0, "tick": 0, "seen": {}, "stats": {"hit": 0, "miss": 0, "skipped": 0, "evicted": 0}}
_panel_state = {"hint": False, "handover": None, "stats": {"emitted": 0, "consumed": 0, "dg": 0}}


class _BlankPanel:
    """a pooled, zero-initialised panel buffer of a logical [rows][K] operand, to be filled by a recurrence kernel.

    A new buffer costs a zero fill over the whole panel (the kernels write the real extent, the padding must be zero),
    which is what the emission saves downstream - so a panel is only worth emitting for a shape that comes back.
    `_BlankPanel.take` therefore returns None the FIRST time a (rows, K) is asked for (the consumer splits its operand
    itself, as without the feature) and allocates on the second; fixed-shape training pays two steps of split passes,
    an epoch of ever-changing lengths pays no fills at all.  The pool is bounded like the exchange pool (LRU)."""
    __slots__ = ("buf", "rows", "K", "flags", "key")

    @staticmethod
    def take(rows, K, device):
        key = (device.index, torch.cuda.current_stream(device).cuda_stream, rows, K)
        free = _panel_pool["free"].get(key)
        if free:
            e = free.pop()
            _panel_pool["bytes"] -= e[0].numel()
            _panel_pool["stats"]["hit"] += 1
            return _BlankPanel(rows, K, key, e[0])
        seen = _panel_pool["seen"]
        if key not in seen:
            if len(seen) > 4096:
                seen.clear()
            seen[key] = 1
            _panel_pool["stats"]["skipped"] += 1
            return None
        _panel_pool["stats"]["miss"] += 1
        n = int(_L().asrk_split_panel_bytes(rows, K, 0))
        buf = torch.empty((n,), dtype=torch.uint8, device=device)
        _lib.check(_L().asrk_fill_f32(_p(buf), n // 4, 0.0, _stream()), "fill")      # panel sizes are multiples of 16 B
        return _BlankPanel(rows, K, key, buf)

    def __init__(self, rows, K, key, buf):
        self.rows, self.K, self.flags, self.key, self.buf = rows, K, 0, key, buf

    def release(self):
        n = self.buf.numel()
        if n > _XCHG_POOL_CAP:
            return
        _pool_evict(_panel_pool, n)
        _panel_pool["tick"] += 1
        _panel_pool["free"].setdefault(self.key, []).append([self.buf, 0, _panel_pool["tick"]])
        _panel_pool["bytes"] += n


def set_panel_hint(flag):
    """the caller (Encoder.forward) knows that the NEXT consumer of the coming LSTM layer's output is another LSTM layer's
    input projection with nothing in between: worth emitting that output as a panel"""
    _panel_state["hint"] = bool(flag)


def _take_handover(x):
    h, _panel_state["handover"] = _panel_state["handover"], None
    if h is None:
        return None
    ref, version, panel = h
    if ref() is x and x._version == version and panel.rows == x.shape[0] * x.shape[1] and panel.K == x.shape[2]:
        return panel
    panel.release()
    return None


def gemm_panels(M, N, K, A, a_row0, a_k0, B, b_row0, b_k0, C, ldc, alpha=1.0, beta=0.0, bias=None, bias2=None):
    """C[M,N] = alpha * A[a_row0:+M, a_k0:+K] B[b_row0:+N, b_k0:+K]^T + beta C (+ biases), A / B SplitPanels"""
    _require_gpu(C)
    avail = C.untyped_storage().nbytes() // C.element_size() - C.storage_offset()
    if ldc < N or avail < (M - 1) * ldc + N:
        raise _lib.AsrkError("gemm_panels: C too small for {}x{} with ld {}".format(M, N, ldc))
    if A.flags != B.flags:
        raise _lib.AsrkError("gemm_panels: the two panels were built in different layouts")
    _lib.check(_L().asrk_gemm_panels_f32(M, N, K, alpha, _p(A.buf), A.rows, A.K, a_row0, a_k0, _p(B.buf), B.rows,
                                         B.K, b_row0, b_k0, beta, _p(C), ldc, _p(bias), _p(bias2), A.flags,
                                         _stream()), "gemm_panels")


def zeros(shape, device):
    """float32 zeros through asrk_fill_f32 (no ATen fill kernel on the hot path)"""
    t = torch.empty(shape, dtype=torch.float32, device=device)
    if t.numel():
        _lib.check(_L().asrk_fill_f32(_p(t), t.numel(), 0.0, _stream()), "fill")
    return t


def copy_flat(dst, src):
    """dst[:] = src for contiguous f32 tensors of equal size (asrk_copy3d_f32)"""
    n = src.numel()
    if n:
        copy3d(src, dst, 1, 1, n, 0, 0, 0, 0)


def copy3d(src, dst, n0, n1, n2, ss0, ss1, ds0, ds1, accumulate=False):
    _require_gpu(dst)
    _lib.check(_L().asrk_copy3d_f32(_p(src), _p(dst), n0, n1, n2, ss0, ss1, ds0, ds1,
                                    int(accumulate), _stream()), "copy3d")


def colsum(X, M, N, ldx, out, accumulate=False):
    _lib.check(_L().asrk_colsum_f32(_p(X), M, N, ldx, _p(out), int(accumulate), _stream()), "colsum")


# --------------------------------------------------------------------------- Linear (+ tanh)
class LinearFn(Function):
    """y = x W^T + b  (torch.nn.Linear semantics; reference: src/asr.py:29,179,243-248)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _require_gpu(x)
        x2 = _f32c(x).reshape(-1, x.shape[-1])
        w = _f32c(weight)
        M, K = x2.shape
        N = w.shape[0]
        if w.dim() != 2 or w.shape[1] != K or (bias is not None and bias.numel() != N):
            # torch.nn.functional.linear raises the same way; never let the kernel read out of bounds
            raise RuntimeError("linear: input [..., {}] cannot be multiplied with weight {} (bias {})".format(
                K, tuple(w.shape), None if bias is None else tuple(bias.shape)))
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        gemm(0, 1, M, N, K, x2, K, w, K, y, N, bias=bias)
        ctx.save_for_backward(x2, w)
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias
        ctx.in_shape = x.shape
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        M, K = x2.shape
        N = w.shape[0]
        dy2 = _f32c(dy).reshape(M, N)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=torch.float32, device=dy.device)
            gemm(0, 0, M, K, N, dy2, N, w, K, dx, K)
            dx = dx.reshape(ctx.in_shape)
        def param_grads():
            dw_ = db_ = None
            if ctx.needs_input_grad[1]:
                dw_ = grad_out(w, (N, K), dy.device)
                gemm(1, 0, N, K, M, dy2, N, x2, K, dw_, K)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db_ = torch.empty((N,), dtype=torch.float32, device=dy.device)
                colsum(dy2, M, N, N, db_)
            return dw_, db_

        if M >= 1024 and _can_defer(w, ctx.bias_ref) and _defer_beside_bptt():
            with _SideStream(dy.device, (dy2, x2)) as side:
                dw, db = param_grads()
                side.keep(dw, db)
        else:
            dw, db = param_grads()
        return dx, dw, db


def linear(x, weight, bias=None):
    if not torch.is_grad_enabled():              # inference: no autograd node (decode positions are launch-bound)
        _require_gpu(x)
        x2 = _f32c(x).reshape(-1, x.shape[-1])
        w = _f32c(weight)
        M, K = x2.shape
        N = w.shape[0]
        if w.dim() != 2 or w.shape[1] != K or (bias is not None and bias.numel() != N):
            raise RuntimeError("linear: input [..., {}] cannot be multiplied with weight {} (bias {})".format(
                K, tuple(w.shape), None if bias is None else tuple(bias.shape)))
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        gemm(0, 1, M, N, K, x2, K, w, K, y, N, bias=bias)
        return y.reshape(*x.shape[:-1], N)
    return LinearFn.apply(x, weight, bias)


class TanhFn(Function):
    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        xc = _f32c(x)
        y = torch.empty_like(xc)
        _lib.check(_L().asrk_tanh_fwd_f32(_p(xc), _p(y), xc.numel(), _stream()), "tanh")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dyc = _f32c(dy)
        dx = torch.empty_like(y)
        _lib.check(_L().asrk_tanh_bwd_f32(_p(y), _p(dyc), _p(dx), y.numel(), _stream()), "tanh_bwd")
        return dx


def tanh(x):
    return TanhFn.apply(x)


def tanh_(x):
    """in place, no autograd (inference buffers); x contiguous float32"""
    _require_gpu(x)
    assert x.is_contiguous() and x.dtype == torch.float32
    _lib.check(_L().asrk_tanh_fwd_f32(_p(x), _p(x), x.numel(), _stream()), "tanh")
    return x


# --------------------------------------------------------------------------- log-softmax
class LogSoftmaxFn(Function):
    """F.log_softmax(x, dim=-1) (reference: src/asr.py:96, src/decode.py:93,121)."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        xc = _f32c(x)
        V = xc.shape[-1]
        rows = xc.numel() // V
        y = torch.empty_like(xc)
        _lib.check(_L().asrk_log_softmax_fwd_f32(_p(xc), _p(y), rows, V, V, _stream()), "log_softmax")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        V = y.shape[-1]
        rows = y.numel() // V
        dyc = _f32c(dy)
        dx = torch.empty_like(y)
        _lib.check(_L().asrk_log_softmax_bwd_f32(_p(y), _p(dyc), _p(dx), rows, V, V, _stream()),
                   "log_softmax_bwd")
        return dx


def log_softmax(x):
    if not torch.is_grad_enabled():              # inference: the kernel call without the autograd node around it
        _require_gpu(x)
        xc = _f32c(x)
        V = xc.shape[-1]
        y = torch.empty_like(xc)
        _lib.check(_L().asrk_log_softmax_fwd_f32(_p(xc), _p(y), xc.numel() // V, V, V, _stream()), "log_softmax")
        return y
    return LogSoftmaxFn.apply(x)


def topk(x, k):
    """(values, indices) of the k largest entries along the last axis, descending; ties -> smaller
    index.  No gradient (selection only: greedy feedback, beam pruning, hypothesis read-out)."""
    _require_gpu(x)
    xc = _f32c(x.detach())
    V = xc.shape[-1]
    rows = xc.numel() // V
    vals = torch.empty(xc.shape[:-1] + (k,), dtype=torch.float32, device=x.device)
    idx = torch.empty(xc.shape[:-1] + (k,), dtype=torch.int64, device=x.device)
    _lib.check(_L().asrk_topk_f32(_p(xc), rows, V, V, k, _p(vals), _p(idx), _stream()), "topk")
    return vals, idx


def argmax(x):
    """torch.argmax(x, dim=-1) (first maximum)"""
    return topk(x, 1)[1].squeeze(-1)


def token_crop(ids, pad_idx=0, eos_idx=1, ignore_repeat=False):
    """rows of token ids [B, T] int64 on the device -> (compacted ids [B, T], lengths [B] int32): everything
    before the first <eos>, pads dropped, CTC repeats merged on request - the crop of every reference text
    encoder's decode() (src/text.py:61-71), for the whole batch in one launch"""
    _require_gpu(ids)
    x = ids.to(torch.int64)
    if x.dim() != 2:
        raise _lib.AsrkError("token_crop expects [B, T] ids")
    if x.stride(1) != 1:
        x = x.contiguous()
    B, T = x.shape
    out = torch.zeros((B, max(T, 1)), dtype=torch.int64, device=x.device)
    n = torch.zeros((B,), dtype=torch.int32, device=x.device)
    _lib.check(_L().asrk_token_crop_i64(_p(x), x.stride(0) if B else T, B, T, pad_idx, eos_idx, int(ignore_repeat),
                                        _p(out), out.stride(0), _p(n), _stream()), "token_crop")
    return out, n


def edit_distance(a, a_len, b, b_len):
    """Levenshtein distances [B] int32 of B pairs of int64 symbol rows (a [B, La], b [B, Lb], lengths int32) on
    the device (replaces editdistance.eval, src/util.py:126)"""
    _require_gpu(a)
    a, b = a.to(torch.int64).contiguous(), b.to(torch.int64).contiguous()
    a_len, b_len = a_len.to(torch.int32).contiguous(), b_len.to(torch.int32).contiguous()
    B = a.shape[0]
    if b.shape[0] != B or a_len.numel() != B or b_len.numel() != B:
        raise _lib.AsrkError("edit_distance: batch sizes differ")
    if b.shape[1] > 4096 and a.shape[1] <= 4096:      # the kernel keeps rows of the SECOND sequence in LDS; the
        a, a_len, b, b_len = b, b_len, a, a_len       # distance is symmetric, so put the shorter one there
    d = torch.empty((B,), dtype=torch.int32, device=a.device)
    _lib.check(_L().asrk_edit_distance_i64(_p(a), a.shape[1], _p(a_len), _p(b), b.shape[1], _p(b_len), B,
                                           b.shape[1], _p(d), _stream()), "edit_distance")
    return d


# --------------------------------------------------------------------------- layout moves
class SwapBTFn(Function):
    """[A,B,F] -> [B,A,F] (batch-major <-> time-major); its own inverse."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        xc = _f32c(x)
        A, B, F = xc.shape
        y = torch.empty((B, A, F), dtype=torch.float32, device=x.device)
        copy3d(xc, y, A, B, F, B * F, F, F, A * F)
        return y

    @staticmethod
    def backward(ctx, dy):
        return SwapBTFn.apply(dy)


def swap_bt(x):
    return SwapBTFn.apply(x)


class PyramidFn(Function):
    """Time reduction of src/module.py:141-153 on a time-major tensor [T,B,F].

    'concat': drop T % r trailing frames, out[t'] = cat(x[r t'], ..., x[r t' + r - 1]) -> [T//r,B,rF]
    'drop'  : out[t'] = x[r t']                                                   -> [ceil(T/r),B,F]
    """

    @staticmethod
    def forward(ctx, x, rate, style):
        _require_gpu(x)
        xc = _f32c(x)
        T, B, F = xc.shape
        ctx.meta = (T, B, F, rate, style)
        if style == "concat":
            To = T // rate
            y = torch.empty((To, B, rate * F), dtype=torch.float32, device=x.device)
            for j in range(rate):
                copy3d(xc[j:], y[:, :, j * F:], To, B, F, rate * B * F, F, B * rate * F, rate * F)
        else:
            To = (T + rate - 1) // rate
            y = torch.empty((To, B, F), dtype=torch.float32, device=x.device)
            copy3d(xc, y, To, B, F, rate * B * F, F, B * F, F)
        return y

    @staticmethod
    def backward(ctx, dy):
        T, B, F, rate, style = ctx.meta
        dyc = _f32c(dy)
        # 'concat' with T % rate == 0 overwrites every element; otherwise dropped frames get zero
        if style == "concat" and T % rate == 0:
            dx = torch.empty((T, B, F), dtype=torch.float32, device=dy.device)
        else:
            dx = zeros((T, B, F), dy.device)
        if style == "concat":
            To = T // rate
            for j in range(rate):
                copy3d(dyc[:, :, j * F:], dx[j:], To, B, F, B * rate * F, rate * F, rate * B * F, F)
        else:
            To = (T + rate - 1) // rate
            copy3d(dyc, dx, To, B, F, B * F, F, rate * B * F, F)
        return dx, None, None


def pyramid(x_tm, rate, style):
    return PyramidFn.apply(x_tm, rate, style)


# --------------------------------------------------------------------------- LSTM layer
class LSTMLayerFn(Function):
    """One (bi)directional LSTM layer on a time-major sequence [T,B,Din] -> [T,B,ndir*H].

    Same math as nn.LSTM(Din, H, bidirectional, num_layers=1) with zero initial state over the full
    padded length (reference: src/module.py:112-113,131).  Input projection = MFMA GEMM, time loop =
    persistent recurrence kernel; backward = persistent BPTT kernel + 4 GEMMs + column sums.
    """

    @staticmethod
    def forward(ctx, x, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r, pyr_rate=1,
                pyr_style=None):
        """pyr_style 'concat' / 'drop' with pyr_rate > 1: the time reduction that follows the layer
        (src/module.py:141-153) is fused into the recurrence kernel's output store and the returned
        tensor is already the next layer's input [T', B, D']; backward reads the gradient in that
        layout directly."""
        _require_gpu(x)
        L = _L()
        xc = _f32c(x)
        T, B, Din = xc.shape
        H = w_hh_f.shape[1]
        ndir = 2 if w_ih_r is not None else 1
        dev = x.device
        M = T * B
        G = torch.empty((M, ndir * 4 * H), dtype=torch.float32, device=dev)
        w_ih_f, w_hh_f = _f32c(w_ih_f), _f32c(w_hh_f)
        w_stack = None
        if ndir == 2:
            w_ih_r, w_hh_r = _f32c(w_ih_r), _f32c(w_hh_r)
        # stacking pays where the GEMMs own the chip (wide layers: the persistent kernels fill every CU, so
        # nothing co-runs); at H = 512 the weight-gradient GEMMs run throttled beside the BPTT kernels and
        # two smaller launches interleave better (cfg2: 22.4 ms/step unstacked, 23.0 stacked)
        if ndir == 2 and H >= 768 and _os.environ.get("ASRK_STACK_DIRS", "1") != "0":
            # both directions read the same X: ONE GEMM with N = 8H over the stacked weights (a device
            # copy of 2 x 4H x Din floats) instead of two - better tile quantisation, X panels fetched
            # once; the same stack serves dX (one K = 8H contraction) and dW_ih in the backward pass
            if (w_ih_f.untyped_storage().data_ptr() == w_ih_r.untyped_storage().data_ptr()
                    and w_ih_r.storage_offset() == w_ih_f.storage_offset() + 4 * H * Din):
                # the parameter container keeps the two directions adjacent (RNNParams.colocate_directions): a view
                w_stack = w_ih_f.as_strided((8 * H, Din), (Din, 1))
            else:
                w_stack = torch.empty((8 * H, Din), dtype=torch.float32, device=dev)
                copy_flat(w_stack[:4 * H], w_ih_f)
                copy_flat(w_stack[4 * H:], w_ih_r)
            if b_ih_f is not None:
                b1 = torch.empty((2, 8 * H), dtype=torch.float32, device=dev)   # rows: b_ih | b_hh, both directions
                for row, (bf_, br_) in enumerate(((b_ih_f, b_ih_r), (b_hh_f, b_hh_r))):
                    copy_flat(b1[row, :4 * H], _f32c(bf_.detach()))
                    copy_flat(b1[row, 4 * H:], _f32c(br_.detach()))
                b1, b2 = b1[0], b1[1]
            else:
                b1 = b2 = None
            x_panel = _take_handover(x) if x is xc else None
            if x_panel is not None and gemm_takes_split(M, 8 * H, Din):
                # the layer below wrote this input as a split panel already: only the weights are split here
                pW = SplitPanel(w_stack, Din, 8 * H, Din, False)
                gemm_panels(M, 8 * H, Din, x_panel, 0, 0, pW, 0, 0, G, 8 * H, bias=b1, bias2=b2)
                del pW
                _panel_state["stats"]["consumed"] += 1
            else:
                gemm(0, 1, M, 8 * H, Din, xc, Din, w_stack, Din, G, 8 * H, bias=b1, bias2=b2)
            if x_panel is not None:
                x_panel.release()
        else:
            gemm(0, 1, M, 4 * H, Din, xc, Din, w_ih_f, Din, G, ndir * 4 * H, bias=b_ih_f, bias2=b_hh_f)
            if ndir == 2:
                gemm(0, 1, M, 4 * H, Din, xc, Din, w_ih_r, Din, G[:, 4 * H:], ndir * 4 * H, bias=b_ih_r,
                     bias2=b_hh_r)
        Y = torch.empty((M, ndir * H), dtype=torch.float32, device=dev)
        C = torch.empty((M, ndir * H), dtype=torch.float32, device=dev)
        ws = lstm_workspace(dev)
        xc_ = _Exchange(L, T, B, H, ndir, 0, dev)
        mode = {None: 0, 'concat': 1, 'drop': 2}[pyr_style if pyr_rate > 1 else None]
        Y2 = None
        if mode == 1:
            Y2 = torch.empty((T // pyr_rate, B, pyr_rate * ndir * H), dtype=torch.float32, device=dev)
        elif mode == 2:
            Y2 = torch.empty(((T + pyr_rate - 1) // pyr_rate, B, ndir * H), dtype=torch.float32, device=dev)
        rate = max(1, pyr_rate)
        out_panel = None
        if (_REC_PANELS and _panel_state["hint"] and mode in (0, 1) and (mode == 0 or T // rate > 0) and
                (ndir * H) % 8 == 0 and L.asrk_lstm_plan_is_bf(T, B, H, ndir, 0, rec_flags(0))):
            r_ = rate if mode == 1 else 1
            out_panel = _BlankPanel.take((T // r_) * B, r_ * ndir * H, dev)
        if out_panel is not None:
            _panel_state["stats"]["emitted"] += 1
            _lib.check(L.asrk_lstm_rec_fwd_pyr_panel_f32(_p(G), _p(w_hh_f), _p(w_hh_r), _p(Y), _p(C), T, B, H, ndir,
                                                         _p(xc_.buf), xc_.prefilled, _p(ws), _p(Y2), mode, rate,
                                                         _p(out_panel.buf), xc_.flags, _stream()), "lstm_rec_fwd")
        else:
            _lib.check(L.asrk_lstm_rec_fwd_pyr_f32(_p(G), _p(w_hh_f), _p(w_hh_r), _p(Y), _p(C), T, B, H, ndir,
                                                   _p(xc_.buf), xc_.prefilled, _p(ws), _p(Y2), mode, rate,
                                                   xc_.flags, _stream()), "lstm_rec_fwd")
        xc_.done()
        ctx.pyr = (mode, max(1, pyr_rate))
        ctx.dims = (T, B, Din, H, ndir)
        ctx.has_bias = b_ih_f is not None
        ctx.bias_refs = (b_ih_f, b_hh_f, b_ih_r, b_hh_r)
        ctx.save_for_backward(xc, w_ih_f, w_hh_f, w_ih_r, w_hh_r, G, C, Y, w_stack)
        ctx.consumed = False
        out = Y2 if mode else Y.view(T, B, ndir * H)
        if _panel_state["handover"] is not None:         # a panel nobody came for
            _panel_state["handover"][2].release()
            _panel_state["handover"] = None
        if out_panel is not None:
            _panel_state["handover"] = (_weakref.ref(out), out._version, out_panel)
        return out

    @staticmethod
    def backward(ctx, dY):
        L = _L()
        xc, w_ih_f, w_hh_f, w_ih_r, w_hh_r, G, C, Y, w_stack = ctx.saved_tensors
        if ctx.consumed:
            raise RuntimeError("LSTMLayerFn: backward twice (the gate buffer is reused in place)")
        ctx.consumed = True
        T, B, Din, H, ndir = ctx.dims
        dev = dY.device
        M = T * B
        ldg, ldy = ndir * 4 * H, ndir * H
        mode, rate = ctx.pyr
        dYc = _f32c(dY)        # plain: [T,B,ldy]; fused time reduction: the reduced layout, read in place
        ws = lstm_workspace(dev)
        # G (activated gates) -> dG (pre-activation gradients), in place
        _note_bptt_plan(L, T, B, H, ndir)
        xc_ = _Exchange(L, T, B, H, ndir, 1, dev)
        # the bias gradient (column sums of dG) comes out of the BPTT kernel itself
        db_all = torch.empty((ndir, 4 * H), dtype=torch.float32, device=dev) if ctx.has_bias else None
        # In-kernel accumulation adds one partial sum per BATCH GROUP (16 or 32 rows) to each element with a
        # float atomic: with <= 2 groups (B <= 32, every BASELINE workload) the result does not depend on the
        # arrival order (a + b == b + a, the first add lands on an exact 0), i.e. it is bit-reproducible;
        # with more groups the order would matter, so those shapes take the deterministic column-sum pass.
        db_in_kernel = ctx.has_bias and B <= 32
        # the input-gradient GEMM dX = dG [W_ih_f; W_ih_r] multiplies dG as its row-major A panel, the weight-gradient GEMMs
        # multiply dG^T as theirs: the BPTT kernel writes both itself where its plan can (bf16x6) and the GEMMs run in
        # stream order (the wide layers; side-stream consumers would outlive the pooled buffer)
        bf_bwd = bool(_REC_PANELS and L.asrk_lstm_plan_is_bf(T, B, H, ndir, 1, rec_flags(1)))
        pG = pGT = None
        if (bf_bwd and ctx.needs_input_grad[0] and w_stack is not None and ndir * 4 * H == 8 * H and
                gemm_takes_split(M, Din, 8 * H)):
            pG = _BlankPanel.take(M, 8 * H, dev)
            _panel_state["stats"]["dg"] += pG is not None
        share0 = (_os.environ.get("ASRK_SHARE_PANELS", "1") != "0" and T > 1 and H % 128 == 0 and B % 8 == 0 and
                  gemm_takes_split(4 * H, H, (T - 1) * B))
        if bf_bwd and share0 and B % 16 == 0 and not _defer_beside_bptt() and ldg == ndir * 4 * H:
            pGT = _BlankPanel.take(ndir * 4 * H, M, dev)
            _panel_state["stats"]["dgt"] = _panel_state["stats"].get("dgt", 0) + (pGT is not None)
        if pG is not None or pGT is not None:
            _lib.check(L.asrk_lstm_rec_bwd_pyr_panel_f32(_p(G), _p(w_hh_f), _p(w_hh_r), _p(C), _p(dYc), T, B, H,
                                                         ndir, _p(xc_.buf), xc_.prefilled, _p(ws),
                                                         _p(db_all if db_in_kernel else None), mode, rate,
                                                         _p(pG.buf if pG is not None else None),
                                                         _p(pGT.buf if pGT is not None else None), xc_.flags, _stream()),
                       "lstm_rec_bwd")
        else:
            _lib.check(L.asrk_lstm_rec_bwd_pyr_f32(_p(G), _p(w_hh_f), _p(w_hh_r), _p(C), _p(dYc), T, B, H,
                                                   ndir, _p(xc_.buf), xc_.prefilled, _p(ws),
                                                   _p(db_all if db_in_kernel else None), mode, rate,
                                                   xc_.flags, _stream()), "lstm_rec_bwd")
        xc_.done()
        if ctx.has_bias and not db_in_kernel:
            colsum(G, M, ndir * 4 * H, ndir * 4 * H, db_all)
        _gemm_phase_begins()
        dG = G
        f32 = dict(dtype=torch.float32, device=dev)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, Din), **f32)
            if pG is not None:            # dG arrived as a panel: only the (transposed) weight stack is split here
                pWT = SplitPanel(w_stack, Din, Din, 8 * H, True)
                gemm_panels(M, Din, 8 * H, pG, 0, 0, pWT, 0, 0, dx, Din)
                del pWT
                pG.release()
            elif w_stack is not None:     # one contraction over both directions' gate gradients (K = 8H)
                gemm(0, 0, M, Din, 8 * H, dG, ldg, w_stack, Din, dx, Din)
            else:
                gemm(0, 0, M, Din, 4 * H, dG, ldg, w_ih_f, Din, dx, Din)
                if ndir == 2:
                    gemm(0, 0, M, Din, 4 * H, dG[:, 4 * H:], ldg, w_ih_r, Din, dx, Din, beta=1.0)
            dx = dx.view(T, B, Din)
        dw_ih_stack = [None]
        # Weight gradients contract over the tokens with dG^T as the left operand three times (dW_ih, dW_hh of
        # both directions): on the split-GEMM path dG^T (and Y^T, X^T) are split ONCE into bf16 panels and the
        # GEMMs take row / k ranges of them.  Needs every direction's GEMMs on one stream (share[1]).
        share = [share0, False]
        panels = {}
        if pGT is not None:
            panels["dGT"] = pGT               # written by the BPTT kernel: no transposed split pass over dG

        def panel(name, src, ld, rows):
            if name not in panels:
                panels[name] = SplitPanel(src, ld, rows, M, True)
            return panels[name]

        def dg_gemm(Mo, No, Ko, m0, k0, pB, b_row0, b_k0, out, ldo):
            """out[Mo, No] = dG[k0 : k0 + Ko, m0 : m0 + Mo]^T  B-panel rows, through the transposed panel dG^T"""
            gemm_panels(Mo, No, Ko, panel("dGT", dG, ldg, ndir * 4 * H), m0, k0, pB, b_row0, b_k0, out, ldo)

        def param_grads_panels(d):
            pY = panel("YT", Y, ldy, ndir * H)
            Mh = (T - 1) * B
            dw_hh = grad_out(w_hh[d], (4 * H, H), dev)
            # direction 0: dG rows of t >= 1 against Y[t-1]; direction 1: dG rows of t <= T-2 against Y[t+1]
            dg_gemm(4 * H, H, Mh, d * 4 * H, B if d == 0 else 0, pY, d * H, 0 if d == 0 else B, dw_hh, H)
            rows_ih = 8 * H if (w_stack is not None and stack_dw) else 4 * H
            if gemm_takes_split(rows_ih, Din, M) and Din % 4 == 0:
                pX = panel("XT", xc, Din, Din)
                if rows_ih == 8 * H:
                    if dw_ih_stack[0] is None:
                        dw_ih_stack[0] = torch.empty((8 * H, Din), **f32)
                        dg_gemm(8 * H, Din, M, 0, 0, pX, 0, 0, dw_ih_stack[0], Din)
                    dw_ih = dw_ih_stack[0][d * 4 * H:(d + 1) * 4 * H]
                else:
                    dw_ih = grad_out(w_ih[d], (4 * H, Din), dev)
                    dg_gemm(4 * H, Din, M, d * 4 * H, 0, pX, 0, 0, dw_ih, Din)
            elif rows_ih == 8 * H:
                if dw_ih_stack[0] is None:
                    dw_ih_stack[0] = torch.empty((8 * H, Din), **f32)
                    gemm(1, 0, 8 * H, Din, M, dG, ldg, xc, Din, dw_ih_stack[0], Din)
                dw_ih = dw_ih_stack[0][d * 4 * H:(d + 1) * 4 * H]
            else:
                dw_ih = grad_out(w_ih[d], (4 * H, Din), dev)
                gemm(1, 0, 4 * H, Din, M, dG[:, d * 4 * H:], ldg, xc, Din, dw_ih, Din)
            db = db2 = None
            if ctx.has_bias:
                db = db_all[d]
                db2 = db.clone()
            return dw_ih, dw_hh, db, db2

        def param_grads(d):
            if share[0] and share[1]:
                return param_grads_panels(d)
            dGd = dG[:, d * 4 * H:]
            if w_stack is not None and stack_dw:
                if dw_ih_stack[0] is None:    # dW_ih of both directions: one GEMM with M = 8H
                    dw_ih_stack[0] = torch.empty((8 * H, Din), **f32)
                    gemm(1, 0, 8 * H, Din, M, dG, ldg, xc, Din, dw_ih_stack[0], Din)
                dw_ih = dw_ih_stack[0][d * 4 * H:(d + 1) * 4 * H]
            else:
                dw_ih = grad_out(w_ih[d], (4 * H, Din), dev)
                gemm(1, 0, 4 * H, Din, M, dGd, ldg, xc, Din, dw_ih, Din)
            dw_hh = zeros((4 * H, H), dev) if T <= 1 else grad_out(w_hh[d], (4 * H, H), dev)
            if T > 1:
                Mh = (T - 1) * B
                if d == 0:   # h_{t-1} = Y[t-1]
                    gemm(1, 0, 4 * H, H, Mh, dGd[B:], ldg, Y, ldy, dw_hh, H)
                else:        # reverse direction: previous state of t is Y[t+1]
                    gemm(1, 0, 4 * H, H, Mh, dGd, ldg, Y[B:, H:], ldy, dw_hh, H)
            db = db2 = None
            if ctx.has_bias:
                db = db_all[d]
                db2 = db.clone()
            return dw_ih, dw_hh, db, db2

        # both directions' dW_ih share one launch only when they are computed on the same stream - and only into
        # scratch: with a data-parallel engine each direction's GEMM writes its rows straight into that weight's slice
        # of the gradient bucket (two launches of 4H rows: still >= 2 full rounds of tiles on the wide layers)
        w_ih, w_hh = (w_ih_f, w_ih_r), (w_hh_f, w_hh_r)
        stack_dw = ((ctx.needs_input_grad[0] or not _can_defer(w_ih_f, w_hh_f, w_ih_r, w_hh_r, *ctx.bias_refs))
                    and not grad_has_destination(*w_ih[:ndir]))
        beside = _defer_beside_bptt() or not ctx.needs_input_grad[0]
        if _can_defer(w_ih_f, w_hh_f, w_ih_r, w_hh_r, *ctx.bias_refs) and beside:
            # off the critical path: the next layer's BPTT does not need dW / db
            if ctx.needs_input_grad[0] or ndir == 1:
                with _SideStream(dev, (dG, xc, Y, db_all)) as side:
                    share[1] = True
                    grads = [param_grads(d) for d in range(ndir)]
                    side.keep(*[t for g in grads for t in g])
            elif pGT is not None and gemm_takes_split(4 * H, Din, M) and Din % 4 == 0:
                # bottom layer with a WIDE input: dW_ih multiplies panels too (X^T) and its GEMMs fill the chip on their
                # own - stream order, like the layers above
                share[1] = True
                grads = [param_grads(d) for d in range(ndir)]
            elif pGT is not None:
                # bottom layer of a wide stack with the dG^T panel from the kernel: both directions' dW_hh multiply row
                # ranges of that ONE panel (and of Y^T, split here before the streams fork; dW_ih with a narrow input
                # takes no panel); the directions still run side by side, the pooled panel goes back at the end of the
                # backward pass (after the streams re-join)
                pY = panel("YT", Y, ldy, ndir * H)
                share[1] = True
                with _SideStream(dev, (dG, xc, Y, db_all, pGT.buf, pY.buf), background=False) as side:
                    g1 = param_grads(1)
                    side.keep(*g1)
                grads = [param_grads(0), g1]
                _defer["release"].append(pGT)
                pGT = None
            else:
                # bottom layer (no input gradient wanted): no BPTT follows, nothing to hide behind.
                # Its small GEMMs (dW_ih with Din = 80, column sums) leave CUs idle one at a time, so
                # the two directions run side by side: reverse on the side stream, forward here.
                with _SideStream(dev, (dG, xc, Y, db_all), background=False) as side:
                    g1 = param_grads(1)
                    side.keep(*g1)
                grads = [param_grads(0), g1]
        else:
            share[1] = True
            grads = [param_grads(d) for d in range(ndir)]
        if pGT is not None:
            if share[1] and not _defer["pending"]:
                pGT.release()                 # every consumer was enqueued on this stream
            else:
                _defer["release"].append(pGT)
        if ndir == 1:
            grads.append((None, None, None, None))
        return (dx,) + grads[0] + grads[1] + (None, None)


class Copy3dFn(Function):
    """Differentiable strided block copy into a zero-filled tensor of `out_shape` (zero padding /
    slicing without ATen math): dst[i0*ds0 + i1*ds1 + e] = src[i0*ss0 + i1*ss1 + e]."""

    @staticmethod
    def forward(ctx, src, out_shape, n0, n1, n2, ss0, ss1, ds0, ds1):
        _require_gpu(src)
        s = _f32c(src)
        dst = zeros(out_shape, src.device)
        copy3d(s, dst, n0, n1, n2, ss0, ss1, ds0, ds1)
        ctx.meta = (tuple(src.shape), n0, n1, n2, ss0, ss1, ds0, ds1)
        return dst

    @staticmethod
    def backward(ctx, g):
        shape, n0, n1, n2, ss0, ss1, ds0, ds1 = ctx.meta
        gs = zeros(shape, g.device)
        copy3d(_f32c(g), gs, n0, n1, n2, ds0, ds1, ss0, ss1)
        return (gs,) + (None,) * 8


def _pad_lstm_params(params, H, Hp):
    """nn.LSTM-layout (w_ih [4H,Din], w_hh [4H,H], b_ih, b_hh) -> the same with H zero-padded to Hp
    per gate.  Padded units see pre-activation 0 and zero initial state:
    c stays 0 and h = o * tanh(0) = 0 at every step, and their W_hh columns are zero, so the real
    units are untouched — exact, not approximate."""
    w_ih, w_hh, b_ih, b_hh = params
    Din = w_ih.shape[1]
    w_ih_p = Copy3dFn.apply(w_ih, (4 * Hp, Din), 4, H, Din, H * Din, Din, Hp * Din, Din)
    w_hh_p = Copy3dFn.apply(w_hh, (4 * Hp, Hp), 4, H, H, H * H, H, Hp * Hp, Hp)
    b_ih_p = Copy3dFn.apply(b_ih, (4 * Hp,), 1, 4, H, 0, H, 0, Hp)
    b_hh_p = Copy3dFn.apply(b_hh, (4 * Hp,), 1, 4, H, 0, H, 0, Hp)
    return w_ih_p, w_hh_p, b_ih_p, b_hh_p


def lstm_layer(x_tm, params_f, params_r=None, pyramid=None):
    """params_* = (w_ih, w_hh, b_ih, b_hh) in torch nn.LSTM layout.  pyramid = (rate, style): also
    apply the time reduction of src/module.py:141-153 (fused into the kernel when the shape allows)."""
    H = params_f[1].shape[1]
    if pyramid is not None and pyramid[0] > 1:
        if H % 4 != 0 or _os.environ.get("ASRK_FUSE_PYRAMID", "1") == "0":
            return PyramidFn.apply(lstm_layer(x_tm, params_f, params_r), pyramid[0], pyramid[1])
        pr = params_r if params_r is not None else (None, None, None, None)
        return LSTMLayerFn.apply(x_tm, *params_f, *pr, pyramid[0], pyramid[1])
    if H % 4 != 0:
        # the persistent recurrence kernels want H % 4 == 0 (16-B h rows): run on zero-padded units
        Hp = (H + 3) // 4 * 4
        ndir = 2 if params_r is not None else 1
        pf = _pad_lstm_params(params_f, H, Hp)
        pr = _pad_lstm_params(params_r, H, Hp) if params_r is not None else (None,) * 4
        T, B = x_tm.shape[0], x_tm.shape[1]
        yp = LSTMLayerFn.apply(x_tm, *pf, *pr)                               # [T,B,ndir*Hp]
        return Copy3dFn.apply(yp, (T, B, ndir * H), T * B, ndir, H, ndir * Hp, Hp, ndir * H, H)
    pr = params_r if params_r is not None else (None, None, None, None)
    return LSTMLayerFn.apply(x_tm, *params_f, *pr)


def lstm_layer_packed(x_tm, params_f, params_r, lens, pyramid=None):
    """Inference-only (bi)LSTM layer over a zero-padded time-major batch [T,B,Din] in which row b is a sequence of
    lens[b] frames: every row is computed exactly as if it had been run ALONE and unpadded (the reverse direction starts
    at its own last frame) - the reference decodes utterance by utterance (src/decode.py:64,88; bin/test_asr.py:163-167),
    this is what lets several utterances share one encoder pass.  Output frames t >= lens[b] are zero.  pyramid =
    (rate, style) fuses the time reduction ('concat' trims lens[b] % rate frames of every row by itself).
    Returns the layer output [T',B,D']; the caller divides the lengths."""
    _require_gpu(x_tm)
    if torch.is_grad_enabled() and x_tm.requires_grad:
        raise _lib.AsrkError("lstm_layer_packed is inference-only")
    L = _L()
    xc = _f32c(x_tm)
    T, B, Din = xc.shape
    w_ih_f, w_hh_f, b_ih_f, b_hh_f = (None if t is None else _f32c(t.detach()) for t in params_f)
    H = w_hh_f.shape[1]
    if H % 4 != 0:
        raise _lib.AsrkError("lstm_layer_packed: hidden size must be a multiple of 4")
    ndir = 2 if params_r is not None else 1
    dev = xc.device
    M = T * B
    G = torch.empty((M, ndir * 4 * H), dtype=torch.float32, device=dev)
    gemm(0, 1, M, 4 * H, Din, xc, Din, w_ih_f, Din, G, ndir * 4 * H, bias=b_ih_f, bias2=b_hh_f)
    w_hh_r = None
    if ndir == 2:
        w_ih_r, w_hh_r, b_ih_r, b_hh_r = (None if t is None else _f32c(t.detach()) for t in params_r)
        gemm(0, 1, M, 4 * H, Din, xc, Din, w_ih_r, Din, G[:, 4 * H:], ndir * 4 * H, bias=b_ih_r, bias2=b_hh_r)
    lens_d = torch.as_tensor(lens).to(device=dev, dtype=torch.int64).contiguous()
    Y = zeros((M, ndir * H), dev)
    C = torch.empty((M, ndir * H), dtype=torch.float32, device=dev)
    rate, style = pyramid if pyramid is not None else (1, None)
    mode = {None: 0, 'concat': 1, 'drop': 2}[style if rate > 1 else None]
    Y2 = None
    if mode == 1:
        Y2 = zeros((T // rate, B, rate * ndir * H), dev)
    elif mode == 2:
        Y2 = zeros(((T + rate - 1) // rate, B, ndir * H), dev)
    ws = lstm_workspace(dev)
    xc_ = _Exchange(L, T, B, H, ndir, 0, dev)
    _lib.check(L.asrk_lstm_rec_fwd_len_f32(_p(G), _p(w_hh_f), _p(w_hh_r), _p(Y), _p(C), _p(lens_d), T, B, H, ndir,
                                           _p(xc_.buf), xc_.prefilled, _p(ws), _p(Y2), mode, max(1, rate), xc_.flags,
                                           _stream()), "lstm_rec_fwd_len")
    xc_.done()
    return Y2 if mode else Y.view(T, B, ndir * H)


# --------------------------------------------------------------------------- CTC loss
class CTCLossFn(Function):
    """torch.nn.CTCLoss(blank, reduction='mean') (reference: bin/train_asr.py:49,123-124)."""

    @staticmethod
    def forward(ctx, log_probs, targets, input_lengths, target_lengths, blank, reduction):
        _require_gpu(log_probs)
        L = _L()
        if log_probs.dtype != torch.float32:
            log_probs = log_probs.float()
        if log_probs.stride(2) != 1:
            log_probs = log_probs.contiguous()
        T, B, V = log_probs.shape
        dev = log_probs.device
        targets = torch.as_tensor(targets).to(device=dev, dtype=torch.int64)
        if targets.dim() != 2:
            raise _lib.AsrkError("CTCLoss: only padded 2-D targets [B, L] are supported "
                                 "(reference passes txt [B,L], bin/train_asr.py:123)")
        targets = targets.contiguous()
        il = torch.as_tensor(input_lengths).to(device=dev, dtype=torch.int64).contiguous()
        tl = torch.as_tensor(target_lengths).to(device=dev, dtype=torch.int64).contiguous()
        Lmax = targets.shape[1]
        S = 2 * Lmax + 1
        alpha = torch.empty((B, T, S), dtype=torch.float32, device=dev)
        lpg = torch.empty((B, T, S), dtype=torch.float32, device=dev)
        # training: the beta lattice runs concurrently with alpha in the same launch
        beta = torch.empty((B, T, S), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        nll = torch.empty((B,), dtype=torch.float32, device=dev)
        _lib.check(L.asrk_ctc_loss_fwd_f32(_p(log_probs), log_probs.stride(0), log_probs.stride(1), T,
                                           B, V, _p(targets), targets.stride(0), Lmax, _p(il),
                                           _p(tl), blank, _p(alpha), _p(beta), _p(lpg), _p(nll),
                                           _stream()), "ctc_fwd")
        ctx.save_for_backward(log_probs, targets, il, tl, alpha, beta, lpg, nll)
        ctx.meta = (T, B, V, Lmax, blank, reduction)
        if reduction == "mean":
            return (nll / tl.clamp(min=1).to(torch.float32)).mean()
        if reduction == "sum":
            return nll.sum()
        return nll

    @staticmethod
    def backward(ctx, gout):
        L = _L()
        log_probs, targets, il, tl, alpha, beta, lpg, nll = ctx.saved_tensors
        T, B, V, Lmax, blank, reduction = ctx.meta
        dev = log_probs.device
        if reduction == "mean":
            gscale = gout.to(torch.float32) / (tl.clamp(min=1).to(torch.float32) * B)
        elif reduction == "sum":
            gscale = gout.to(torch.float32).expand(B)
        else:
            gscale = gout.to(torch.float32)
        gscale = gscale.contiguous()
        # the gradient takes the memory layout of the input: the solver passes ctc_output.transpose(0, 1)
        # (a [T,B,V] view of a [B,T,V] tensor), so autograd's transpose-back is then a free view and the
        # log-softmax backward reads a contiguous tensor (no 160 MB re-layout copy)
        if log_probs.stride(1) > log_probs.stride(0):
            grad = torch.empty((B, T, V), dtype=torch.float32, device=dev).transpose(0, 1)
        else:
            grad = torch.empty((T, B, V), dtype=torch.float32, device=dev)
        _lib.check(L.asrk_ctc_loss_bwd_f32(_p(log_probs), log_probs.stride(0), log_probs.stride(1), T,
                                           B, V, _p(targets), targets.stride(0), Lmax, _p(il),
                                           _p(tl), blank, _p(alpha), _p(beta), _p(lpg), _p(nll), _p(gscale),
                                           _p(grad), grad.stride(0), grad.stride(1), _stream()),
                   "ctc_bwd")
        return grad, None, None, None, None, None


class CTCLoss(torch.nn.Module):
    """Drop-in for torch.nn.CTCLoss(blank=0, zero_infinity=False) on the MI355X path."""

    def __init__(self, blank=0, reduction="mean", zero_infinity=False):
        super().__init__()
        if zero_infinity:
            raise NotImplementedError("zero_infinity=True is not used by the reference "
                                      "(bin/train_asr.py:49)")
        self.blank = blank
        self.reduction = reduction

    def forward(self, log_probs, targets, input_lengths, target_lengths):
        return CTCLossFn.apply(log_probs, targets, input_lengths, target_lengths, self.blank,
                               self.reduction)


# --------------------------------------------------------------------------- cross entropy
class CrossEntropyFn(Function):
    """CrossEntropyLoss(ignore_index, reduction='mean') on logits [R,V], targets [R]
    (reference: bin/train_asr.py:47,130-131)."""

    @staticmethod
    def forward(ctx, logits, targets, ignore_index):
        _require_gpu(logits)
        x = _f32c(logits)
        R, V = x.shape
        tg = targets.to(device=x.device, dtype=torch.int64).contiguous()
        lse = torch.empty((R,), dtype=torch.float32, device=x.device)
        sums = torch.empty((2,), dtype=torch.float32, device=x.device)
        _lib.check(_L().asrk_cross_entropy_fwd_f32(_p(x), R, V, V, _p(tg), ignore_index, _p(lse),
                                                   _p(sums), _stream()), "cross_entropy")
        ctx.save_for_backward(x, tg, lse, sums)
        ctx.ignore_index = ignore_index
        return sums[0] / sums[1]

    @staticmethod
    def backward(ctx, gout):
        x, tg, lse, sums = ctx.saved_tensors
        R, V = x.shape
        gscale = (gout.to(torch.float32) / sums[1]).reshape(1).contiguous()
        dx = torch.empty_like(x)
        _lib.check(_L().asrk_cross_entropy_bwd_f32(_p(x), R, V, V, _p(tg), ctx.ignore_index, _p(lse),
                                                   _p(gscale), _p(dx), _stream()), "cross_entropy_bwd")
        return dx, None, None


class CrossEntropyLoss(torch.nn.Module):
    """Drop-in for torch.nn.CrossEntropyLoss(ignore_index=0) on the MI355X path."""

    def __init__(self, ignore_index=-100, reduction="mean"):
        super().__init__()
        assert reduction == "mean"
        self.ignore_index = ignore_index

    def forward(self, logits, targets):
        return CrossEntropyFn.apply(logits, targets, self.ignore_index)


# --------------------------------------------------------------------------- LayerNorm / dropout
class LayerNormFn(Function):
    """torch.nn.LayerNorm(D) over the last axis (reference: src/module.py:116-117,135-136)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        _require_gpu(x)
        xc = _f32c(x)
        D = xc.shape[-1]
        if weight.numel() != D or bias.numel() != D:
            raise RuntimeError("layer_norm: normalized_shape {} does not match input [..., {}]".format(
                tuple(weight.shape), D))
        rows = xc.numel() // D
        w, b = _f32c(weight), _f32c(bias)
        y = torch.empty_like(xc)
        mean = torch.empty((rows,), dtype=torch.float32, device=x.device)
        rstd = torch.empty((rows,), dtype=torch.float32, device=x.device)
        _lib.check(_L().asrk_layer_norm_fwd_f32(_p(xc), _p(w), _p(b), _p(y), _p(mean), _p(rstd), rows, D,
                                                float(eps), _stream()), "layer_norm")
        ctx.save_for_backward(xc, w, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, w, mean, rstd = ctx.saved_tensors
        D = xc.shape[-1]
        rows = xc.numel() // D
        dyc = _f32c(dy)
        dx = torch.empty_like(xc) if ctx.needs_input_grad[0] else None
        want_p = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dw = torch.empty((D,), dtype=torch.float32, device=dy.device) if want_p else None
        db = torch.empty((D,), dtype=torch.float32, device=dy.device) if want_p else None
        _lib.check(_L().asrk_layer_norm_bwd_f32(_p(xc), _p(w), _p(dyc), _p(mean), _p(rstd), _p(dx), _p(dw),
                                                _p(db), rows, D, _stream()), "layer_norm_bwd")
        return dx, dw, db, None


def layer_norm(x, weight, bias, eps):
    return LayerNormFn.apply(x, weight, bias, eps)


class DropoutFn(Function):
    """Inverted dropout whose mask is a pure function of (seed, element index): nothing but the
    64-bit seed is kept for the backward (reference: nn.Dropout at src/module.py:118-119,137-138,
    src/asr.py:36,162; the RNG stream necessarily differs from torch's)."""

    @staticmethod
    def forward(ctx, x, p, seed):
        _require_gpu(x)
        xc = _f32c(x)
        y = torch.empty_like(xc)
        _lib.check(_L().asrk_dropout_f32(_p(xc), _p(y), xc.numel(), float(p), seed, 0, _stream()), "dropout")
        ctx.p, ctx.seed = float(p), seed
        return y

    @staticmethod
    def backward(ctx, dy):
        dyc = _f32c(dy)
        dx = torch.empty_like(dyc)
        _lib.check(_L().asrk_dropout_f32(_p(dyc), _p(dx), dyc.numel(), ctx.p, ctx.seed, 0, _stream()),
                   "dropout_bwd")
        return dx, None, None


def dropout(x, p, training, seed=None):
    """nn.Dropout(p)(x).  `seed` (optional, 64-bit) pins the mask; by default it is drawn from torch's
    CPU generator, so torch.manual_seed() makes runs reproducible without a device sync."""
    if not training or p == 0:
        return x
    if p >= 1:
        raise ValueError("dropout probability has to be in [0, 1), got {}".format(p))
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    return DropoutFn.apply(x, p, seed)
