#!/usr/bin/env python
"""bench.py — training throughput of the LAS/CTC hot path on MI355X (BASELINE.json metric:
audio frames/sec training).

One "step" = one full optimiser step of the reference's training loop (bin/train_asr.py:95-137,
src/solver.py:76-91): forward (encoder -> CTC head [-> attention decoder]) + losses + backward +
[gradient all-reduce] + clip_grad_norm_(5.0) + Adadelta step, on a synthetic LibriSpeech-shaped
batch that is already resident in HBM.  Default workload = BASELINE.json configs[2] ("cfg3", the
configuration the metric is quoted on: full LAS, 4 x pBLSTM-1024 [2,2,2,1] concat + location-aware
attention + LSTM-1024 decoder + CTC hybrid lambda 0.5, B=32 x T=1600 x 80-mel, V=5000, L=64);
`--workload cfg2` = configs[1] (2 x pBLSTM-512, CTC-only, T=1000).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...          # no WORLD_SIZE in the environment: bench.py starts the N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   # the driver's form

One process per GPU (rank r on GPU r), RCCL through torch.distributed ("nccl").  The N-rank line carries RCCL's own
view of the job (`rccl`: world size, the all-reduced sum of the ranks, exposed all-reduce ms per step).

Rank 0 prints ONE JSON line (its last line of stdout).  `roofline` is for the dominant kernel by time (the bf16x6
split GEMM), measured with hipEvents on the launch streams INSIDE the timed region; `roofline_recurrence` (the
persistent LSTM recurrence kernels: latency-bound dependent chain), `roofline_hbm` (the HBM-bound kernels: split
passes, optimiser, CTC gradient, feature front end, against 8 TB/s) and `kernel_families` come from hipEvents over a
few extra untimed steps right after the region (an event pair costs ~3 us of stream time: all families together were
1 ms per cfg3 step); `cpu_baseline` times the CPU oracle (a port of the reference's --cpu arithmetic) on the host cores.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "end-to-end-asr-pytorch_amd"

import torch  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[1]
    "cfg2": dict(B=32, T=1000, D=80, V=5000, L=64,
                 model=dict(ctc_weight=1.0,
                            encoder=dict(prenet='', module='LSTM', bidirection=True, dim=[512, 512],
                                         dropout=[0, 0], layer_norm=[False, False],
                                         proj=[False, False], sample_rate=[2, 2],
                                         sample_style='concat'),
                            attention={}, decoder={})),
    # BASELINE.json configs[2]
    "cfg3": dict(B=32, T=1600, D=80, V=5000, L=64,
                 model=dict(ctc_weight=0.5,
                            encoder=dict(prenet='', module='LSTM', bidirection=True,
                                         dim=[1024] * 4, dropout=[0] * 4, layer_norm=[False] * 4,
                                         proj=[False] * 4, sample_rate=[2, 2, 2, 1],
                                         sample_style='concat'),
                            attention=dict(mode='loc', dim=300, num_head=1, v_proj=False,
                                           temperature=0.5, loc_kernel_size=100, loc_kernel_num=10),
                            decoder=dict(module='LSTM', dim=1024, layer=1, dropout=0))),
    # BASELINE.json configs[0]: the architecture the reference ships (config/libri/asr_example.yaml:34-59) at its own
    # batch size - VGG prenet on 40 fbank x (static, delta, delta-delta), 5 x BLSTM-512 + tanh(Linear), location-aware
    # attention, LSTM-512 decoder, attention only, subword-16k vocabulary; T = 800 frames (8 s), L = 40
    "shipped": dict(B=16, T=800, D=120, V=16000, L=40,
                    model=dict(ctc_weight=0.0,
                               encoder=dict(prenet='vgg', module='LSTM', bidirection=True, dim=[512] * 5,
                                            dropout=[0] * 5, layer_norm=[False] * 5, proj=[True] * 5,
                                            sample_rate=[1] * 5, sample_style='drop'),
                               attention=dict(mode='loc', dim=300, num_head=1, v_proj=False,
                                              temperature=0.5, loc_kernel_size=100, loc_kernel_num=10),
                               decoder=dict(module='LSTM', dim=512, layer=1, dropout=0))),
}
# the shipped architecture behind the reference's other prenet (src/module.py:68-90, `prenet: 'cnn'`)
WORKLOADS["cnn"] = dict(WORKLOADS["shipped"], model=dict(
    WORKLOADS["shipped"]["model"], encoder=dict(WORKLOADS["shipped"]["model"]["encoder"], prenet='cnn')))

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32-input MFMA peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak
# f32 contractions on the bf16 matrix cores by exact 3-way operand splitting (csrc/gemm_split.hip): six bf16
# MFMA products per f32 product -> the roofline of that kernel in f32-equivalent flops
SPLIT_GEMM_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0
HBM_PEAK_GBS = 8000.0          # spec; 6.3 TB/s achievable


def encoder_algorithmic_work(w):
    """SURVEY.md §8(d) formulas: per-layer compulsory bytes / ih flops / hh flops (forward)."""
    B, T, d_in = w["B"], w["T"], w["D"]
    enc = w["model"]["encoder"]
    tot = dict(bytes=0, flops_ih=0, flops_hh=0, steps=0, flops_prenet=0, flops_proj=0)
    if enc.get("prenet") == "vgg":
        # VGGExtractor (src/module.py:7-66): C x F input planes -> 64, 64, pool, 128, 128, pool; 3x3 convs, pad 1
        C, Fq = (d_in // 13, 13) if d_in % 13 == 0 else (d_in // 40, 40)
        tot["bytes"] += 4 * B * T * d_in
        t, f = T - T % 4, Fq
        for ci, co, pool in ((C, 64, False), (64, 64, True), (64, 128, False), (128, 128, True)):
            tot["flops_prenet"] += 2 * B * t * f * co * ci * 9
            tot["bytes"] += 4 * (co * ci * 9 + co)
            if pool:
                t, f = t // 2, f // 2
        T, d_in = t, f * 128
    elif enc.get("prenet") == "cnn":
        # CNNExtractor (src/module.py:68-90): Conv1d(D -> dim[0], 4, stride 2, pad 1) twice
        O = enc["dim"][0]
        tot["bytes"] += 4 * (B * T * d_in + O * d_in * 4 + O * O * 4 + 2 * O)
        tot["flops_prenet"] += 2 * B * (T // 2) * O * d_in * 4 + 2 * B * (T // 4) * O * O * 4
        T, d_in = T // 4, O
    for l, H in enumerate(enc["dim"]):
        tot["bytes"] += 4 * (B * T * d_in + B * T * 2 * H + 2 * (4 * H * d_in + 4 * H * H + 8 * H))
        tot["flops_ih"] += 2 * B * T * d_in * 8 * H
        tot["flops_hh"] += 2 * B * T * H * 8 * H
        tot["steps"] += T
        r = enc["sample_rate"][l]
        d_in = 2 * H * (r if enc["sample_style"] == "concat" else 1)
        T = T // r
        if enc["proj"][l]:                          # tanh(Linear(d, d)) after the time reduction (src/module.py:154-156)
            tot["flops_proj"] += 2 * B * T * d_in * d_in
            tot["bytes"] += 4 * (d_in * d_in + d_in + 2 * B * T * d_in)
    return tot


def synth_batch(B, T, D, V, L, seed):
    """SURVEY.md §8d throughput batch: feat ~ N(0,1) [B,T,D], every utterance full length; txt tokens
    uniform in [3,V) ending with <eos>=1, lengths in [L/2, L], 0-padded (product-side generator: the
    GPU leg imports nothing from oracle/)."""
    g = torch.Generator().manual_seed(seed)
    feat = torch.randn(B, T, D, generator=g)
    lens = torch.full((B,), T, dtype=torch.long)
    txt = torch.zeros(B, L, dtype=torch.long)
    tl = torch.randint(max(2, L // 2), L + 1, (B,), generator=g)
    tl[0] = L
    for b in range(B):
        n = int(tl[b])
        txt[b, :n - 1] = torch.randint(3, V, (n - 1,), generator=g)
        txt[b, n - 1] = 1
    return feat, lens, txt


def synth(w, seed, device):
    feat, feat_len, txt = synth_batch(w["B"], w["T"], w["D"], w["V"], w["L"], seed=seed)
    return feat.to(device), feat_len.to(device), txt.to(device)


def build_model(w, device):
    asr = importlib.import_module(PKG + ".src.asr")
    torch.manual_seed(0)
    m = w["model"]
    model = asr.ASR(w["D"], w["V"], True, m["ctc_weight"], m["encoder"], m["attention"], m["decoder"])
    return model.to(device).train()


def _cpu_oracle_step_fn(workload):
    """(fwd_bwd(nb), full optimiser step(), workload dict): the oracle's arithmetic on the GPU run's tensors"""
    from oracle import asr_oracle as O
    w = dict(WORKLOADS[workload])
    m = w["model"]
    sd = O.make_state_dict(m, w["D"], w["V"], seed=0)
    params = [v.requires_grad_(True) for v in sd.values()]
    opt = torch.optim.Adadelta(params, lr=1.0, eps=1e-8)
    feat, feat_len, txt = synth_batch(w["B"], w["T"], w["D"], w["V"], w["L"], seed=0)
    L = int((txt != 0).sum(-1).max())

    def fwd_bwd(nb):
        opt.zero_grad()
        c, l, a, _, _ = O.asr_forward(sd, m, feat[:nb], feat_len[:nb], L, teacher=txt[:nb], lstm_impl="aten")
        total, _, _ = O.asr_losses(m, c, l, a, txt[:nb])
        total.backward()

    def step():
        fwd_bwd(w["B"])
        torch.nn.utils.clip_grad_norm_(params, 5.0)
        opt.step()

    return fwd_bwd, step, w


def _cpu_probe_worker(workload, threads):
    """one forward + backward on a quarter of the batch (full T, L) at `threads` threads, after one untimed pass;
    prints the seconds.  Run by cpu_baseline() under a timeout: a 256-thread oneDNN LSTM can crawl for minutes."""
    torch.set_num_threads(threads)
    fwd_bwd, _, w = _cpu_oracle_step_fn(workload)
    nb = max(1, w["B"] // 4)
    fwd_bwd(nb)
    t0 = time.time()
    fwd_bwd(nb)
    print(json.dumps({"probe_s": time.time() - t0, "threads": threads}), flush=True)


def _cpu_baseline_worker(workload, steps, threads, probe):
    """Oracle (port of the reference --cpu arithmetic incl. ATen lstm / ctc_loss, clip_grad_norm_, Adadelta) on
    the host cores, BASELINE.md §3 protocol: the FULL batch of the workload (identical synthetic tensors to the
    GPU run), 1 warm-up + `steps` (>= 3) timed optimiser steps at the thread count cpu_baseline() found fastest.
    Prints one JSON line after every timed step so that a budget timeout still leaves the best estimate so far."""
    ncpu = os.cpu_count() or 1
    torch.set_num_threads(threads)
    _, step, w = _cpu_oracle_step_fn(workload)
    step()  # warm-up
    t0 = time.time()
    for i in range(steps):
        step()
        dt = (time.time() - t0) / (i + 1)
        print(json.dumps({"value": w["B"] * w["T"] / dt, "unit": "frames/s", "cores": threads,
                          "kind": "port", "timed_steps": i + 1, "s_per_step": dt, "host_cores": ncpu,
                          "thread_probe_s": probe,
                          "sample": "%d full optimiser steps (fwd + CTC/CE losses + bwd + clip_grad_norm_ + Adadelta) on "
                                    "the FULL batch (B=%d, T=%d, L=%d, the GPU run's tensors) after 1 warm-up; "
                                    "kind=port because /root/reference does not exist on the GPU box: the CPU "
                                    "oracle restates the reference --cpu path on the same ATen lstm / ctc_loss "
                                    "(torch %s); %d of %d host threads = the fastest of the candidates %s on a "
                                    "quarter-batch forward+backward probe (null = the probe did not finish within "
                                    "its time limit); %.2f s/step" % (
                                        i + 1, w["B"], w["T"], w["L"], torch.__version__, threads, ncpu,
                                        json.dumps(probe), dt)}), flush=True)


def cpu_baseline(workload, budget_s=480):
    """full batch, >= 1 warm-up + 3 timed steps, thread count = the fastest of {32, 64, all host
    cores}, stated.  Every candidate is probed in its own subprocess under a timeout (ATen's CPU LSTM can take
    minutes at hundreds of threads); then the timed run, bounded by `budget_s`.  Returns the last JSON line the
    worker printed."""
    import subprocess
    ncpu = os.cpu_count() or 1
    me = [sys.executable, os.path.abspath(__file__), "--workload", workload]
    probe, limit = {}, 90.0
    for t in sorted({min(32, ncpu), min(64, ncpu), ncpu}):
        try:
            r = subprocess.run(me + ["--cpu-probe", str(t)], capture_output=True, text=True, timeout=limit)
            line = [x for x in r.stdout.splitlines() if x.startswith("{")]
            probe[str(t)] = round(json.loads(line[-1])["probe_s"], 3) if line else None
        except subprocess.TimeoutExpired:
            probe[str(t)] = None
        done = [v for v in probe.values() if v]
        if done:
            limit = 3.0 * min(done) + 30.0          # a candidate slower than 3x the best so far is abandoned
    ok = {int(k): v for k, v in probe.items() if v}
    threads = min(ok, key=ok.get) if ok else min(32, ncpu)
    out, err = "", ""
    try:
        r = subprocess.run(me + ["--cpu-baseline-only", "--cpu-threads", str(threads), "--cpu-probe-json",
                                 json.dumps(probe)], capture_output=True, text=True, timeout=budget_s)
        out, err = r.stdout, r.stderr
    except subprocess.TimeoutExpired as e:
        out = (e.stdout or b"").decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        err = "budget of %d s exhausted" % budget_s
    for line in reversed(out.strip().splitlines()):
        if line.startswith("{"):
            res = json.loads(line)
            if res.get("timed_steps", 0) < 3:
                res["sample"] += " [only %d timed step(s) fit the %d s budget]" % (res.get("timed_steps", 0), budget_s)
            return res
    return {"value": None, "unit": "frames/s", "cores": threads, "kind": "port", "thread_probe_s": probe,
            "sample": "cpu oracle produced no timed step: " + err[-300:]}


def kernel_source_digest():
    """sha256 over the kernel sources whose HBM traffic profiles/rNN_hbm_traffic_*.json describes; the PMC
    passes (tools/pmc_hbm.sh) stamp it into the summary, and a summary whose stamp differs from the sources
    this run was built from is STALE: its traffic is then reported as null, not silently reused."""
    import hashlib
    h = hashlib.sha256()
    for f in ("gemm_split.hip", "gemm.hip", "lstm_rec.hip", "lstm_rec_common.h", "lstm_rec_fwd_f32.hip",
              "lstm_rec_fwd_bf.hip", "lstm_rec_bwd_f32.hip", "lstm_rec_bwd_bf.hip"):
        with open(os.path.join(ROOT, PKG, "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def conv_source_digest():
    """the same stamp for the implicit-GEMM convolution kernels (`roofline_conv.traffic` of the VGG-fronted workloads)"""
    import hashlib
    with open(os.path.join(ROOT, PKG, "csrc", "conv3x3.hip"), "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()[:16]


def newest_traffic_summary(workload, digest):
    """(path, parsed json) of the newest profiles/rNN_hbm_traffic_<workload>.json, or (path, None) when its stamp is not
    `digest` (collected on other kernel sources: stale), or ("", None) when there is none"""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic_%s.json" % workload)), reverse=True)
    cands = [c for c in cands if "f32mfma" not in os.path.basename(c)]
    if not cands:
        return "", None
    tj = json.load(open(cands[0]))
    return cands[0], (tj if tj.get("kernel_source_digest") == digest else None)


def isolated_gemm_rate(ops, w, device, reps=5):
    """the largest contraction of the workload (encoder layer-1 input projection, one direction:
    [T/2*B, 4H(=2*2H)] x [4H, 4H]^T) launched alone: what the in-situ rate (co-scheduled with the
    recurrence kernels and the side stream) is to be compared with."""
    enc = w["model"]["encoder"]
    H = enc["dim"][1] if len(enc["dim"]) > 1 else enc["dim"][0]
    r = enc["sample_rate"][0]
    t_enc = w["T"] // (4 if enc.get("prenet") else 1)
    M, N, K = (t_enc // r) * w["B"], 4 * H, 2 * enc["dim"][0] * r
    A = torch.randn(M, K, device=device)
    Bm = torch.randn(N, K, device=device)
    C = torch.empty(M, N, device=device)
    ops.gemm(0, 1, M, N, K, A, K, Bm, K, C, N)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(0, 1, M, N, K, A, K, Bm, K, C, N)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    peak = SPLIT_GEMM_PEAK_TFLOPS if ops.get_gemm_split() > 0 else F32_MFMA_PEAK_TFLOPS
    return {"shape_MNK": [M, N, K], "ms": ms, "achieved": tf, "frac": tf / peak,
            "frac_of_f32_mfma_peak": tf / F32_MFMA_PEAK_TFLOPS}


def frontend_rate(lib, w, device, reps=5):
    """the whole-batch feature front end (src/audio.py:BatchFeatureTransform, csrc/audio.hip: since round 6 TWO launches -
    fbank_logmel_batch_kernel: framing, 512-point FFT in LDS, power, mel weights, log, one wave per frame; then
    delta_cmvn_batch_kernel) on a synthetic int16 PCM batch of the workload's shape, against the HBM roofline on its
    ALGORITHMIC bytes (2 B per sample in, 4 B per output feature out).  Rounds 3-5 ran the DFT as a dense [frames x 512] x
    [512 x 514] GEMM with materialised operands (1.11 ms, 0.4 % of the roofline)."""
    import ctypes
    import numpy as np
    audio = importlib.import_module(PKG + ".src.audio")
    B, T, D = w["B"], w["T"], w["D"]
    delta = 0 if D in (40, 80) else 2
    tf, feat_dim = audio.create_transform(dict(feat_type="fbank", feat_dim=D // (delta + 1), frame_length=25,
                                               frame_shift=10, dither=0, apply_cmvn=True, delta_order=delta,
                                               delta_window_size=2), device=str(device))
    bt = getattr(tf, "batch", None)
    if bt is None:
        return {"kernel": "feature front end", "note": "no whole-batch transform for this configuration"}
    n = 400 + 160 * (T - 1)
    rng = np.random.default_rng(0)
    pcm = [(rng.standard_normal(n) * 3000).astype(np.int16) for _ in range(B)]
    with torch.no_grad():
        bt(pcm, 16000)                                       # warm-up (tables, staging buffer)
        torch.cuda.synchronize()
        lib.asrk_profile_reset()
        lib.asrk_profile_enable(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            feat, flen = bt(pcm, 16000)
        e1.record()
        torch.cuda.synchronize()
        lib.asrk_profile_enable(0)
    ms_k, nl = ctypes.c_double(0), ctypes.c_int64(0)
    lib.asrk_profile_get(7, ctypes.byref(ms_k), ctypes.byref(nl))
    ms_g, ng = ctypes.c_double(0), ctypes.c_int64(0)
    lib.asrk_profile_get(0, ctypes.byref(ms_g), ctypes.byref(ng))
    by = 2.0 * B * n + 4.0 * float(feat.numel())
    ms = (ms_k.value + ms_g.value) / reps
    gbs = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    frames = int(flen.sum())
    return {"kernel": "feature front end: fbank_logmel_batch (framing + FFT in LDS + power + mel + log) + delta_cmvn_batch "
                      "(device kernels of one %d x %d-frame batch; wall incl. host padding and the PCM upload: %.2f ms)" % (
                          B, T, e0.elapsed_time(e1) / reps),
            "bytes_per_step": by, "ms_per_step": ms, "ms_streaming_kernels": ms_k.value / reps,
            "ms_gemms": ms_g.value / reps, "launches": (nl.value + ng.value) / reps, "achieved": gbs, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "frames_per_s": frames / (ms * 1e-3) if ms > 0 else 0.0,
            "bound": "two launches of ~5 us floor each plus, per frame, a 256-point complex FFT in LDS (~4 k LDS "
                     "accesses per wave) and the three sweeps of the CMVN over L2-resident features: latency / LDS "
                     "bound on %.0f MB of algorithmic bytes, not HBM" % (by / 1e6)}


def build_step(workload, device, dist=None, rank=0, force_collectives=False):
    """model + one full optimiser step (forward, CTC/CE losses, backward, clip, Adadelta) on a
    resident synthetic batch of `workload`; returns (model, step) with step() -> (loss, grad_norm)"""
    ops = importlib.import_module(PKG + ".ops")
    w = WORKLOADS[workload]
    model = build_model(w, device)
    world = dist.get_world_size() if dist is not None else 1
    engine = (importlib.import_module(PKG + ".parallel").DataParallelEngine(model, dist, force_collectives=force_collectives)
              if world > 1 or (dist is not None and force_collectives) else None)
    params = list(model.parameters())
    # config/libri/asr_example.yaml:28-30 (Adadelta lr 1.0 eps 1e-8); fused streaming kernel, same state
    opt = importlib.import_module(PKG + ".fused_optim").FusedAdadelta(params, lr=1.0, eps=1e-8)
    ctc_loss_fn = ops.CTCLoss(blank=0)
    ce_loss_fn = ops.CrossEntropyLoss(ignore_index=0) if model.enable_att else None

    feat, feat_len, txt = synth(w, seed=rank, device=device)    # per-rank data, same model seed
    txt_len = torch.sum(txt != 0, dim=-1)
    L = int(txt_len.max())
    n_tok_local = txt_len.sum().to(torch.float32)

    def step():
        opt.zero_grad(set_to_none=True)
        ctc_out, enc_len, att_out, _, _ = model(feat, feat_len, L, tf_rate=1.0, teacher=txt)
        total = 0
        if ctc_out is not None:
            total = total + ctc_loss_fn(ctc_out.transpose(0, 1), txt, enc_len, txt_len) * model.ctc_weight
        if att_out is not None:
            b, t, _ = att_out.shape
            ce = ce_loss_fn(att_out.view(b * t, -1), txt.view(-1))
            if engine is not None:
                # CrossEntropy(mean) over the GLOBAL batch: local mean * n_local / (n_global / world),
                # then gradient averaging (SURVEY.md §8e condition 2; bin/train_asr.py:130-131)
                ce = ce * (n_tok_local / engine.token_normaliser(n_tok_local)).squeeze(0)
            total = total + ce * (1 - model.ctc_weight)
        if engine is not None:
            engine.backward(total)
        else:
            total.backward()
        gn = opt.clip_and_step(5.0)          # clip_grad_norm_(params, 5.0) + step, one pass over the gradients
        return total, gn

    def probe():
        """forward + backward on the current weights WITHOUT an update: (loss, gradient norm) in float64"""
        opt.zero_grad(set_to_none=True)
        ctc_out, enc_len, att_out, _, _ = model(feat, feat_len, L, tf_rate=1.0, teacher=txt)
        total = 0
        if ctc_out is not None:
            total = total + ctc_loss_fn(ctc_out.transpose(0, 1), txt, enc_len, txt_len) * model.ctc_weight
        if att_out is not None:
            b, t, _ = att_out.shape
            total = total + ce_loss_fn(att_out.view(b * t, -1), txt.view(-1)) * (1 - model.ctc_weight)
        total.backward()
        ops.join_deferred()
        sq = sum(float(p.grad.double().pow(2).sum()) for p in params if p.grad is not None)
        return float(total.detach()), sq ** 0.5

    step.probe = probe
    step.engine = engine
    return model, step


# ------------------------------------------------------------------------------------------------ cfg5 (decode)
# BASELINE.json configs[4] / SURVEY.md §8(d) cfg5: joint CTC-attention beam search (beam 16, ctc_weight 0.5 -> 24
# candidates, src/decode.py:64-173) + RNN-LM shallow fusion (lm_weight 0.5, 2 x LSTM-1024 over the same 5000 tokens),
# max_len_ratio 0.07 / min_len_ratio 0.01 (config/libri/decode_example.yaml), cfg3 acoustic model, seeded random-init
# weights.  A "step" = one device batch of U utterances of 8 s through BeamDecoder.forward_batch (the reference fans
# utterances out over CPU worker processes, bin/test_asr.py:163-167; here their beams are rows of one device batch).
# Multi-GPU: utterance replicas, no data-path collective (SURVEY.md §8e "replicas only").
CFG5_LM = dict(emb_tying=False, emb_dim=1024, module='LSTM', dim=1024, n_layers=2, dropout=0.0)
CFG5_DECODE = dict(beam_size=16, min_len_ratio=0.01, max_len_ratio=0.07, ctc_weight=0.5, lm_weight=0.5)
CFG5 = dict(U=32, T=800, D=80)


def cfg5_utterance(T, D=80, seed=5):
    g = torch.Generator().manual_seed(seed + T)
    return torch.randn(1, T, D, generator=g), torch.tensor([T])


def cfg5_bytes_per_decode_step(w, U, Te):
    """Algorithmic HBM bytes of ONE decode step of a batch of U utterances (f32): every weight matrix the step
    multiplies is read once (decoder cell incl. the embedding columns, query projection, character projection, the
    LM's two cells and its projection), every utterance's attention key / value memory once (its 16 beams share it),
    and its CTC log-probabilities for the 24 candidates' prefix scores once.  The beams' own state (16 U rows of a few
    KB) is not counted."""
    m = w["model"]
    H, A, V = m["decoder"]["dim"], m["attention"]["dim"], w["V"]
    Dv = 2 * m["encoder"]["dim"][-1]
    E = H                                             # pre_embed: vocab -> dec_dim (src/asr.py:32)
    dec = 4 * H * (E + Dv) + 4 * H * H + A * H + V * H
    lm = 2 * (4 * 1024 * 1024 + 4 * 1024 * 1024) + V * 1024
    per_utt = Te * (A + Dv) + Te * V
    return 4.0 * (dec + lm + U * per_utt)


def decode_main(args, rank, world, device, dist):
    import tempfile
    import yaml
    asr_decode = importlib.import_module(PKG + ".src.decode")
    lm_mod = importlib.import_module(PKG + ".src.lm")
    ops = importlib.import_module(PKG + ".ops")
    lib = importlib.import_module(PKG + "._lib").load()
    w = WORKLOADS["cfg3"]
    model = build_model(w, device).eval()
    torch.manual_seed(1)
    lm_sd = lm_mod.RNNLM(w["V"], **CFG5_LM).state_dict()
    tmp = tempfile.mkdtemp()
    torch.save({'model': lm_sd}, os.path.join(tmp, 'lm.pth'))
    yaml.safe_dump({'model': CFG5_LM}, open(os.path.join(tmp, 'lm.yaml'), 'w'))
    dec = asr_decode.BeamDecoder(model, None, **dict(CFG5_DECODE, lm_path=os.path.join(tmp, 'lm.pth'),
                                                     lm_config=os.path.join(tmp, 'lm.yaml'))).to(device)
    U, T = CFG5["U"], CFG5["T"]
    feat = torch.cat([cfg5_utterance(T, seed=5 + 1000 * rank + u)[0] for u in range(U)]).to(device)   # resident in HBM
    flen = torch.full((U,), T, dtype=torch.int64, device=device)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            hyps = dec.forward_batch(feat, flen)
        ops.check_errors()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            hyps = dec.forward_batch(feat, flen)
        fence()
        dt = time.perf_counter() - t0
        ops.check_errors()
        if dist is not None:
            tt = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        n_dec = max(len(h[0].outIndex) for h in hyps)             # decode steps of the batch = its longest hypothesis
        # launches per decode step: the library's own count over one more batch (every family's hooks on)
        import ctypes
        lib.asrk_profile_reset()
        lib.asrk_profile_families(0xffffffff)
        lib.asrk_profile_enable(1)
        dec.forward_batch(feat, flen)
        torch.cuda.synchronize()
        lib.asrk_profile_enable(0)
        launches = 0
        for idx in range(13):
            ms, n = ctypes.c_double(0), ctypes.c_int64(0)
            lib.asrk_profile_get(idx, ctypes.byref(ms), ctypes.byref(n))
            launches += n.value
        # one utterance at a time (the reference's batch = 1 decoding, src/decode.py:64): reported beside the headline
        f1, l1 = feat[:1], flen[:1]
        dec(f1, l1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            h1 = dec(f1, l1)
        torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t1) / 3
    if rank != 0:
        return
    Te = T // 8                                                    # cfg3 encoder: time reduced by 2 * 2 * 2 * 1
    ms_step = dt / args.steps * 1e3
    ms_dec = ms_step / n_dec
    bytes_dec = cfg5_bytes_per_decode_step(w, U, Te)
    gbs = bytes_dec / (ms_dec * 1e-3) / 1e9
    out = {
        "metric": "joint CTC-attention beam-search decode (beam 16) + RNN-LM shallow fusion",
        "value": world * U * args.steps / dt, "unit": "utt/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "rtf": dt / (args.steps * U * T * 0.01),
        "config": {"workload": "cfg5: %d utterances of %d frames (8 s) per device batch, %s, cfg3 acoustic model + "
                               "2xLSTM-1024 LM, V=5000" % (U, T, json.dumps(CFG5_DECODE)),
                   "parallelism": "replicas x%d (utterances sharded over ranks, no data-path collective)" % world,
                   "decode_steps_per_batch": n_dec, "ms_per_decode_step": ms_dec,
                   "launches_per_decode_step": launches / n_dec,
                   "one_utterance_at_a_time": {"s_per_utt": dt1, "utt_per_s": 1.0 / dt1, "rtf": dt1 / (T * 0.01),
                                               "ms_per_decode_step": dt1 * 1e3 / len(h1[0].outIndex)}},
        "roofline": {"kernel": "one decode position of the batch (decoder / LM cells and the vocabulary projections - "
                               "bf16x6 panel GEMMs against weight panels split once at >= 128 live rows, weight-streaming "
                               "skinny kernels below; attention over the utterances' key / value memory; prefix scores; "
                               "radix top-k; the beam bookkeeping on the device): every operand is read once per position",
                     "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                     "traffic": None, "bytes_per_decode_step": bytes_dec,
                     "note": "a decode position is a chain of ~%d dependent library launches (plus the gathers between "
                             "them) with no read-back: latency- and launch-bound, not bandwidth-bound; the host looks at "
                             "the number of unfinished utterances every 8th position" % round(launches / n_dec)},
    }
    if not args.no_cpu_baseline and world == 1:
        from oracle import beam_oracle as BO          # checker-side code: CPU baseline leg only
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        torch.set_num_threads(min(args.cpu_threads, os.cpu_count() or 1))
        f, fl = cfg5_utterance(T, seed=5)
        t0 = time.perf_counter()
        BO.beam_search(sd, w["model"], f, fl, lm_sd={k: v.cpu() for k, v in lm_sd.items()}, lm_cfg=CFG5_LM,
                       lstm_impl="aten", **CFG5_DECODE)
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 1.0 / cdt, "unit": "utt/s", "rtf": cdt / (T * 0.01),
                               "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "one utterance of %d frames (8 s), joint CTC-attention + LM, beam 16, through "
                                         "oracle/beam_oracle.py (the restatement of src/decode.py:64-173 pinned "
                                         "hypothesis for hypothesis on the reference): %.1f s" % (T, cdt)}
    print(json.dumps(out), flush=True)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` with no launcher around it: re-run this file as N ranks of ONE node under
    torch.distributed.run (static rendezvous on 127.0.0.1 - the container hostname may not resolve), one process per
    GPU.  Rank 0's JSON line goes to this process' stdout unchanged; returns the launcher's exit code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "8")
    env["ASRK_BENCH_SPAWNED"] = "1"
    return subprocess.call(cmd, env=env)


def comm_selfcheck(dist, rank, world, device):
    """the communicator's own account of the job: its world size and an all-reduce of the rank numbers (sum must be
    N(N-1)/2: every rank took part exactly once)"""
    t = torch.tensor([float(rank)], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rank_sum_allreduce": float(t.item()),
            "rank_sum_expected": world * (world - 1) / 2.0}


def dry_spawn_main(args):
    """--dry-spawn: the launch path without a GPU (CI here): every rank joins a gloo group, takes part in the
    self-check all-reduce and a barrier; rank 0 prints the line skeleton."""
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    chk = comm_selfcheck(dist, rank, world, torch.device("cpu"))
    pids = [None] * world
    dist.all_gather_object(pids, (rank, int(os.environ.get("LOCAL_RANK", "0")), os.getpid()))
    dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_spawn": True, "n_gpus": world, "gpus_flag": args.gpus,
                          "config": {"parallelism": "dp%d" % world}, "rccl": chk,
                          "ranks": [{"rank": r, "local_rank": lr, "pid": pid} for r, lr, pid in pids]}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS) + ["cfg5"],
                    help="cfg2 / cfg3 / shipped / cnn: training step (frames/s); cfg5: beam-search decode (utt/s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-check", action="store_true",
                    help="skip the exact-f32-MFMA cross-check (loss / gradient norm / step time without operand splitting)")
    ap.add_argument("--cpu-baseline-only", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--cpu-probe", type=int, default=0)
    ap.add_argument("--cpu-probe-json", default="{}")
    ap.add_argument("--print-kernel-digest", action="store_true")
    ap.add_argument("--dry-spawn", action="store_true",
                    help="exercise the N-rank launch path without GPUs (gloo): ranks rendezvous, all-reduce their "
                         "rank numbers, rank 0 prints the line skeleton")
    args = ap.parse_args()
    if args.print_kernel_digest:
        print("conv " + conv_source_digest())
        print(kernel_source_digest())             # last line: what tools/pmc_hbm.sh stamps as kernel_source_digest
        return
    if args.cpu_probe:
        _cpu_probe_worker(args.workload, args.cpu_probe)
        return
    if args.cpu_baseline_only:
        _cpu_baseline_worker(args.workload, max(3, args.cpu_steps), args.cpu_threads, json.loads(args.cpu_probe_json))
        return

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started plainly (`python bench.py --gpus N`): this process becomes the launcher of N ranks
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, args.gpus):
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE)" % (args.gpus, world))
    if args.dry_spawn:
        dry_spawn_main(args)
        return
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    assert local_rank < torch.cuda.device_count(), "rank %d has no GPU %d on this node" % (rank, local_rank)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    # ASRK_BENCH_FORCE_DIST=1: take the data-parallel path (RCCL communicator, gradient buckets, collectives from
    # the backward hooks) even with ONE rank - the way to run the multi-GPU code on a 1-GPU box
    force_dist = os.environ.get("ASRK_BENCH_FORCE_DIST", "0") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    if args.workload == "cfg5":
        decode_main(args, rank, world, device, dist)
        if dist is not None:
            dist.destroy_process_group()
        return
    ops = importlib.import_module(PKG + ".ops")
    lib = importlib.import_module(PKG + "._lib").load()
    w = WORKLOADS[args.workload]
    model, step = build_step(args.workload, device, dist=dist, rank=rank, force_collectives=force_dist)

    comm = comm_selfcheck(dist, rank, world, device) if dist is not None else None
    for _ in range(args.warmup):
        step()
    ops.check_errors()
    if step.engine is not None:
        step.engine.timing = True

    # hipEvents around Encoder.forward (the north-star names the encoder forward)
    enc_events = []

    def _enc_pre(mod, inp):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        enc_events.append([ev, None])

    def _enc_post(mod, inp, out):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        enc_events[-1][1] = ev

    hooks = [model.encoder.register_forward_pre_hook(_enc_pre), model.encoder.register_forward_hook(_enc_post)]

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # hipEvents inside the TIMED region: the dominant kernel's family (the GEMMs, roofline.achieved) only.  An event
    # pair costs ~3 us of stream time (its marker packet waits for the kernel in front of it): all ~180 pairs of a cfg3
    # step were measured at 1.0 ms per step (103.0 vs 104.0 ms), so the other families are timed in PROF_STEPS extra,
    # untimed steps right after the region (same process, same tensors).
    import ctypes
    GEMM_FAMILIES = (1 << 0) | (1 << 8)
    lib.asrk_profile_reset()
    lib.asrk_profile_families(GEMM_FAMILIES)
    lib.asrk_profile_enable(1)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, gn = step()
    fence()
    dt = time.perf_counter() - t0
    lib.asrk_profile_enable(0)
    ops.check_errors()
    for h in hooks:
        h.remove()
    enc_ms = sum(a.elapsed_time(b) for a, b in enc_events) / max(1, len(enc_events))
    if dist is not None:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    def read_families(ids, n_steps, with_work=False):
        res = {}
        for name, idx in ids:
            ms, n, wk = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_double(0)
            lib.asrk_profile_get(idx, ctypes.byref(ms), ctypes.byref(n))
            lib.asrk_profile_get_work(idx, ctypes.byref(wk))
            res[name] = {"ms_per_step": ms.value / n_steps, "launches_per_step": n.value / n_steps,
                         "_work_per_step": wk.value / n_steps}
        return res

    fam = read_families((("gemm", 0), ("gemm_bg", 8)), args.steps)
    PROF_STEPS = max(1, min(args.steps, 5))
    lib.asrk_profile_reset()
    lib.asrk_profile_families(0xffffffff & ~GEMM_FAMILIES)
    lib.asrk_profile_enable(1)
    for _ in range(PROF_STEPS):
        step()
    fence()
    lib.asrk_profile_enable(0)
    lib.asrk_profile_families(0xffffffff)
    fam.update(read_families((("lstm_fwd", 1), ("lstm_bwd", 2), ("ctc", 3), ("rowops", 4), ("attn", 5), ("cell", 6),
                              ("speller", 9), ("conv", 10), ("split", 11), ("optim", 12), ("conv_mfma", 13)), PROF_STEPS))

    if rank == 0:
        work_of = {k: v.pop("_work_per_step") for k, v in fam.items()}
        work = encoder_algorithmic_work(w)
        ms_step = dt / args.steps * 1e3
        frames = w["B"] * w["T"] * world
        # Dominant kernel by time: the f32-MFMA GEMM (one kernel template, ~45 % of the kernel time of a
        # step; rocprofv3 summary in profiles/).  achieved = algorithmic flops (2*M*N*K of every call,
        # counted by the library while the hipEvent hooks are on) / hipEvent-measured kernel time of the
        # family on the streams it was launched on.  With the weight-gradient GEMMs overlapping the BPTT
        # kernels that time includes the contention, i.e. this is the in-situ rate, not a microbenchmark.
        def gemm_rate(idx, name):
            fl, ms = work_of[name], fam[name]["ms_per_step"]
            return fl, ms, (fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0)
        # foreground = launches at full occupancy on the critical path; background = the weight-gradient
        # GEMMs deliberately launched at one workgroup per CU on the side stream (they yield the chip to
        # the BPTT kernels they overlap with, so their own duration is long by design)
        fg_fl, fg_ms, fg_tf = gemm_rate(0, "gemm")
        bg_fl, bg_ms, bg_tf = gemm_rate(8, "gemm_bg")
        gemm_fl, gemm_ms = fg_fl + bg_fl, fg_ms + bg_ms
        gemm_tf = gemm_fl / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        # second family: the persistent LSTM recurrence (latency-bound dependent chain; each launch of a
        # layer does flops_hh of that layer, so per step fwd and bwd families do flops_hh each)
        rec_ms = fam["lstm_fwd"]["ms_per_step"] + fam["lstm_bwd"]["ms_per_step"]
        rec_flops = 2 * work["flops_hh"]
        achieved = rec_flops / (rec_ms * 1e-3) / 1e12 if rec_ms > 0 else 0.0
        fwd_ms = fam["lstm_fwd"]["ms_per_step"]
        # the recurrence kernels multiply on the bf16 matrix cores (exact 3-way split) at H = 512 / 1024
        enc_dims = w["model"]["encoder"]["dim"]
        rec_bf = (os.environ.get("ASRK_REC_BF", "1") != "0" and w["model"]["encoder"]["module"] in ("LSTM", "GRU")
                  and all(d in (512, 1024) for d in enc_dims))
        rec_peak = SPLIT_GEMM_PEAK_TFLOPS if rec_bf else F32_MFMA_PEAK_TFLOPS
        # HBM bytes per launch from the PMC passes of tools/pmc_hbm.sh ((2*FETCH_SIZE + WRITE_SIZE)*1024,
        # gfx950 correction of MI355X_MICROARCH.md §HBM).  Hardware counters cannot be read from inside this
        # process (rocprofv3 owns them), so the committed summary of the same command is reported - but only
        # if it was collected on THESE kernel sources (kernel_source_digest stamp); a stale summary gives null.
        traffic = rec_traffic = None
        traffic_note = "no HBM-traffic summary under profiles/ for this workload"
        split_on = ops.get_gemm_split() > 0
        gemm_peak = SPLIT_GEMM_PEAK_TFLOPS if split_on else F32_MFMA_PEAK_TFLOPS
        tpath, tj = newest_traffic_summary(args.workload, kernel_source_digest())
        if tpath and tj is None:
            traffic_note = "%s was collected on other kernel sources (now %s): stale, dropped" % (
                os.path.basename(tpath), kernel_source_digest())
        if tj:
            ks = tj["kernels"]
            traffic_note = os.path.basename(tpath)

            def per_launch(prefix):
                sel = [v for k, v in ks.items() if k.split("::")[-1].startswith(prefix)]
                n = sum(v["launches"] for v in sel)
                return sum(v["hbm_bytes_per_launch"] * v["launches"] for v in sel) / n if n else None
            traffic, rec_traffic = per_launch("gemm_bf16x6" if split_on else "gemm_f32"), per_launch("lstm_rec_")
        out = {
            "metric": "audio frames/sec training (LAS+CTC, LibriSpeech 80-mel)",
            "value": frames / (dt / args.steps), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "dtype_note": ("f32 values everywhere (operands, accumulators, state). Large GEMMs form each f32 product "
                           "as six exact bf16-MFMA partial products of the exactly split operands (error vs float64 "
                           "equal to the f32-MFMA kernel's: tests/test_kernels_gpu.py::test_gemm_split_*); "
                           "ASRK_GEMM_SPLIT=0 runs them on v_mfma_f32_32x32x2_f32") if split_on else "exact f32 MFMA",
            "gemm_arithmetic": "bf16x6 (exact split)" if split_on else "f32 MFMA",
            "data": "synthetic", "loss": float(loss.detach()), "grad_norm": float(gn),
            "config": {"workload": "%s: %s" % (args.workload, json.dumps(
                {k: w[k] for k in ("B", "T", "D", "V", "L")})), "global_batch": w["B"] * world,
                "parallelism": "dp%d" % world},
            "roofline": {"kernel": ("gemm_bf16x6 (f32 operands split exactly into 3 bf16 planes, 6 bf16-MFMA products "
                                    "per f32 product, f32 accumulate; small / skinny shapes stay on gemm_f32)")
                         if split_on else "gemm_f32 (128x128x32 f32-MFMA tiles + skinny-M streaming variants)",
                         "bound": "mfma", "achieved": gemm_tf, "peak": gemm_peak,
                         "peak_note": ("bf16 dense MFMA peak 2500 TF/s / 6 products; f32-equivalent flops. "
                                       "f32-input MFMA peak would be 157.3") if split_on else "f32-input MFMA peak",
                         "unit": "TFLOP/s", "frac": gemm_tf / gemm_peak,
                         "frac_of_f32_mfma_peak": gemm_tf / F32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "flops_per_step": gemm_fl,
                         "launches_per_step": fam["gemm"]["launches_per_step"] + fam["gemm_bg"]["launches_per_step"],
                         "foreground": {"achieved": fg_tf, "frac": fg_tf / gemm_peak,
                                        "ms_per_step": fg_ms, "flops_per_step": fg_fl},
                         "background": {"achieved": bg_tf, "frac": bg_tf / gemm_peak,
                                        "ms_per_step": bg_ms, "flops_per_step": bg_fl}},
            "roofline_recurrence": {"kernel": ("lstm_rec_fwd_bf+lstm_rec_bwd_bf (persistent; bf16x6 operand split; bound by "
                                               "the inter-workgroup hand-off and the CU's 64 B/clk fragment path)")
                                    if rec_bf else "lstm_rec_fwd+lstm_rec_bwd (persistent, latency-bound)",
                                    "bound": "mfma", "achieved": achieved, "peak": rec_peak,
                                    "unit": "TFLOP/s", "frac": achieved / rec_peak,
                                    "frac_of_f32_mfma_peak": achieved / F32_MFMA_PEAK_TFLOPS,
                                    "traffic": rec_traffic,
                                    "us_per_recurrent_step_fwd": fwd_ms * 1e3 / work["steps"],
                                    "us_per_recurrent_step_bwd":
                                        fam["lstm_bwd"]["ms_per_step"] * 1e3 / work["steps"]},
            # the north-star's named target.  With exact-f32 arithmetic the encoder forward is bounded by
            # the f32-MFMA rate and the dependent recurrence chain, not by HBM: the compulsory bytes need
            # 0.26-0.33 ms at cfg3 while the flops alone need 30 ms at peak, so hbm_fraction cannot
            # exceed ~1 %; both fractions are reported as asked.
            "encoder_fwd": {"ms": enc_ms, "compulsory_bytes": work["bytes"], "flops_ih": work["flops_ih"],
                            "flops_hh": work["flops_hh"], "flops_prenet": work["flops_prenet"],
                            "flops_proj": work["flops_proj"], "dependent_steps": work["steps"],
                            "hbm_fraction_of_8.0TBs": work["bytes"] / (enc_ms * 1e-3) / 8.0e12,
                            "hbm_fraction_of_6.3TBs": work["bytes"] / (enc_ms * 1e-3) / 6.3e12,
                            "mfma_fraction": (work["flops_ih"] + work["flops_hh"] + work["flops_prenet"]
                                              + work["flops_proj"]) / (enc_ms * 1e-3)
                            / (F32_MFMA_PEAK_TFLOPS * 1e12)},
            "kernel_families": fam,
            "kernel_families_note": ("gemm / gemm_bg: hipEvents inside the timed region; every other family: hipEvents over "
                                     "%d extra untimed steps right after it (an event pair costs ~3 us of stream time; all "
                                     "families together were 1.0 ms per cfg3 step).  `split` (the f32 -> bf16-plane passes) is "
                                     "nested inside `gemm`: its ms are part of gemm's" % PROF_STEPS),
            "launches_per_step": sum(v["launches_per_step"] for k, v in fam.items() if k != "split"),
            # recurrence launches served without (hit) / with (miss) a sentinel fill pass, panels taken from the pool
            # (hit), zero-filled anew (miss) or not emitted (skipped: first sight of a shape) over the whole run
            "pools": ops.pool_stats(),
        }
        if comm is not None:
            eng = step.engine
            # RCCL's own account of the job (torch.distributed "nccl" == RCCL): world size, every rank present once,
            # and the part of the gradient all-reduce that was NOT hidden behind the backward pass
            comm.update({"exposed_allreduce_ms_per_step": eng.exposed_allreduce_ms() if eng is not None else None,
                         "gradient_bytes": sum(b["numel"] for b in eng._buckets) * 4 if eng is not None else 0,
                         "buckets": len(eng._buckets) if eng is not None else 0,
                         "launcher": "bench.py (self-spawned)" if os.environ.get("ASRK_BENCH_SPAWNED") == "1"
                         else "external (torch.distributed.run)"})
            out["rccl"] = comm
        if fam["conv_mfma"]["ms_per_step"] > 0:
            # the VGG prenet's implicit-GEMM convolutions (csrc/conv3x3.hip) run on the f32-input matrix cores: their own
            # family, priced against THAT peak (they are not part of `roofline`, whose peak is the bf16x6 one)
            cfl, cms = work_of["conv_mfma"], fam["conv_mfma"]["ms_per_step"]
            ctf = cfl / (cms * 1e-3) / 1e12
            ctraffic = None
            if tj is not None and tj.get("conv_source_digest") == conv_source_digest():
                csel = [v for k, v in tj["kernels"].items() if k.split("::")[-1].startswith(("conv3x3_kernel", "conv3x3_wgrad_kernel"))]
                cn = sum(v["launches"] for v in csel)
                ctraffic = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in csel) / cn if cn else None
            out["roofline_conv"] = {"kernel": "conv3x3_kernel / conv3x3_wgrad_kernel: 3x3 convolutions as implicit GEMMs on "
                                              "v_mfma_f32_32x32x2_f32 (forward, data gradient, weight gradient), in situ",
                                    "bound": "mfma", "achieved": ctf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": ctf / F32_MFMA_PEAK_TFLOPS, "flops_per_step": cfl, "ms_per_step": cms,
                                    "launches_per_step": fam["conv_mfma"]["launches_per_step"], "traffic": ctraffic,
                                    "traffic_note": "HBM bytes per launch of the conv3x3 kernels (PMC summary stamped with "
                                                    "conv3x3.hip's digest), launch-weighted mean" if ctraffic else
                                                    "no PMC summary stamped with this conv3x3.hip"}
        # The HBM-bound kernels of the step against the 8 TB/s roofline: algorithmic bytes the library
        # counted for the family / hipEvent time of its launches INSIDE the timed region; plus the feature front end
        # (off the resident-batch step), run here once on a synthetic PCM batch of the same shape.
        def hbm_row(name, idx, what):
            by, ms = work_of[name], fam[name]["ms_per_step"]
            gbs = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {"kernel": what, "bytes_per_step": by, "ms_per_step": ms,
                    "launches_per_step": fam[name]["launches_per_step"], "achieved": gbs, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
        out["roofline_hbm"] = [
            hbm_row("split", 11, "split_panel_kernel / split_panel_t_kernel: f32 operand -> three bf16 planes (4 B read + "
                    "6 B written per element), every GEMM operand of the step; in situ"),
            hbm_row("optim", 12, "sqnorm_partial + multi_step_kernel<Adadelta>: gradient norm (4 B per element) + clipped "
                    "update (read p, g, 2 states; write p, 2 states: 28 B per parameter); in situ"),
            hbm_row("ctc", 3, "ctc_gather + ctc_lattice (alpha / beta) + ctc_grad_dense + ctc_grad_fix: bytes = the dense "
                    "gradient (read every log-prob, write every gradient element); the lattice phase is a T'-step "
                    "dependent chain, not bandwidth; in situ"),
            {"kernel": "sentinel_fill_kernel", "ms_per_step": 0.0, "note": "gone from the step since round 5: the "
             "recurrence launches hand their exchange buffers back armed (ASRK_REC_REARM), pooled buffers are filled "
             "once at first use"},
            frontend_rate(lib, w, device)]
        out["roofline"]["isolated"] = isolated_gemm_rate(ops, w, device)
        if world == 1 and split_on and not args.no_exact_check:
            # the same step with every contraction on the f32-input MFMA (no operand splitting anywhere):
            # loss / gradient norm of one forward + backward on the SAME weights under both arithmetics, and
            # the step time of the exact-f32 path, so the line carries both numbers
            la, ga = step.probe()
            ops.set_gemm_split(0)
            saved = {k: os.environ.get(k) for k in ("ASRK_REC_BF", "ASRK_REC_BF_BWD")}
            os.environ["ASRK_REC_BF"] = "0"
            os.environ["ASRK_REC_BF_BWD"] = "0"
            lb, gb = step.probe()
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n_exact = 5
            for _ in range(n_exact):
                step()
            torch.cuda.synchronize()
            dt_exact = (time.perf_counter() - t1) / n_exact
            ops.set_gemm_split(1)
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            out["exact_f32_mfma"] = {
                "what": ("same workload with ASRK_GEMM_SPLIT_OFF / ASRK_REC_F32_MFMA on every call: every "
                         "product formed by v_mfma_f32_* (no bf16 planes)"),
                "ms_per_step": dt_exact * 1e3, "value": frames / dt_exact, "steps": n_exact,
                "loss_same_weights": {"default": la, "exact_f32_mfma": lb, "rel_diff": abs(la - lb) / max(abs(lb), 1e-30)},
                "grad_norm_same_weights": {"default": ga, "exact_f32_mfma": gb,
                                           "rel_diff": abs(ga - gb) / max(abs(gb), 1e-30)}}
        out["roofline"]["traffic_source"] = traffic_note
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes a version banner through C stdio, which (redirected) is flushed at exit, i.e. AFTER anything
        # Python printed: push it out first so that the JSON line is the last line of this rank's stdout
        try:
            import ctypes as _ct
            _ct.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
