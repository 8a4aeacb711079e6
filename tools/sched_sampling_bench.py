#!/usr/bin/env python
"""Scheduled sampling (0 < tf_rate < 1, src/asr.py:119-135) at cfg3: forward + losses + backward per step through the
two-pass fused loop (ASR._scheduled_sampling_inputs + the teacher-forced loop on the mixed tokens) against the per-step
autograd path (ASRK_SPELLER=0), and full teacher forcing for scale; with `--gru` the same architecture with a GRU-1024
decoder under full teacher forcing, one-node loop (asrk_speller_t::cell = 1) against the per-step GRU kernels; with
`--layers N` a stacked N-layer LSTM-1024 decoder (asrk_speller_t::nlayer, round 6) likewise.
python tools/sched_sampling_bench.py [tf_rate] [--gru | --layers N] [--dot]"""
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import copy
import torch

bench = importlib.import_module("bench")
ops = importlib.import_module(bench.PKG + ".ops")
nums = [a for a in sys.argv[1:] if not a.startswith("--")]
GRU = "--gru" in sys.argv
LAYERS = int(sys.argv[sys.argv.index("--layers") + 1]) if "--layers" in sys.argv else 1
if "--layers" in sys.argv:
    nums = [a for a in nums if a != sys.argv[sys.argv.index("--layers") + 1]]
tf = float(nums[0]) if nums else 0.5
dev = torch.device("cuda", 0)
w = copy.deepcopy(bench.WORKLOADS["cfg3"])
if GRU:
    w["model"]["decoder"]["module"] = "GRU"
if LAYERS > 1:
    w["model"]["decoder"]["layer"] = LAYERS
DOT = "--dot" in sys.argv            # the verdict's variant: dot-product attention, 4 heads of 256, value projection
if DOT:
    w["model"]["attention"] = dict(mode='dot', dim=256, num_head=4, v_proj=True, temperature=1.0, loc_kernel_size=3,
                                   loc_kernel_num=4)
model = bench.build_model(w, dev)
feat, feat_len, txt = bench.synth(w, seed=0, device=dev)
txt_len = torch.sum(txt != 0, dim=-1)
L = int(txt_len.max())
ctc_fn, ce_fn = ops.CTCLoss(blank=0), ops.CrossEntropyLoss(ignore_index=0)


def run(tf_rate, n):
    for i in range(n + 2):
        if i == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        for p in model.parameters():
            p.grad = None
        ctc_out, enc_len, att_out, _, _ = model(feat, feat_len, L, tf_rate=tf_rate, teacher=txt)
        b, t, _ = att_out.shape
        total = ctc_fn(ctc_out.transpose(0, 1), txt, enc_len, txt_len) * 0.5 + \
            ce_fn(att_out.view(b * t, -1), txt.view(-1)) * 0.5
        total.backward()
        ops.join_deferred()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


if GRU or LAYERS > 1 or DOT:
    out = {"workload": "cfg3 with %sa %s decoder, forward + losses + backward (no update), tf_rate 1" % (
        "4-head dot-product attention (dim 256, value projection) and " if DOT else "",
        "GRU-1024" if GRU else "%d-layer LSTM-1024" % LAYERS)}
    out["one_node_loop_ms"] = run(1.0, 5)
    os.environ["ASRK_SPELLER"] = "0"
    out["per_step_kernels_ms"] = run(1.0, 3)
else:
    out = {"workload": "cfg3 forward + losses + backward (no update)", "tf_rate": tf}
    out["teacher_forcing_ms"] = run(1.0, 5)
    out["scheduled_two_pass_fused_ms"] = run(tf, 5)
    os.environ["ASRK_SPELLER"] = "0"
    out["scheduled_per_step_kernels_ms"] = run(tf, 3)
print(json.dumps(out))
