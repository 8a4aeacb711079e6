"""How far is each arithmetic of the cfg3 training step from the TRUTH?  One forward + backward of BASELINE
configs[2] (B=16, T=1600, L=64 by default) under
  * the oracle in float64 on the host (oracle/asr_oracle.py on ATen CPU kernels)          = the truth,
  * the oracle in float32 on the host                                                     = the reference's own arithmetic,
  * the HIP path with every product on v_mfma_f32_* (ASRK_GEMM_SPLIT_OFF, ASRK_REC_F32_MFMA),
  * the HIP path as shipped (bf16x6 exact operand split in the large GEMMs and the recurrences),
and for each: loss, relative L2 error of the whole gradient vector and the worst per-tensor relative L2 error
against float64.  Checker-side tool (uses oracle/): GPU box only.
    python tools/arith_vs_float64.py [--B 16 --T 1600 --L 64] [--out profiles/r03_arith_vs_float64.json]"""
import argparse
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import asr_oracle as O
from oracle.gen_golden import synth_batch, CFG3_MODEL

PKG = "end-to-end-asr-pytorch_amd"
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=16)
ap.add_argument("--T", type=int, default=1600)
ap.add_argument("--L", type=int, default=64)
ap.add_argument("--threads", type=int, default=32)
ap.add_argument("--out", default="")
args = ap.parse_args()
torch.set_num_threads(args.threads)
D, V = 80, 5000
feat, feat_len, txt = synth_batch(args.B, args.T, D, V, args.L, seed=23)
sd = O.make_state_dict(CFG3_MODEL, D, V, seed=5)


def host(dtype):
    sdr = {k: (v.to(dtype) if v.is_floating_point() else v).clone().requires_grad_(v.is_floating_point())
           for k, v in sd.items()}
    t0 = time.time()
    c, l, a, s, _ = O.asr_forward(sdr, CFG3_MODEL, feat.to(dtype), feat_len, args.L, teacher=txt, lstm_impl="aten")
    tot, _, _ = O.asr_losses(CFG3_MODEL, c, l, a, txt)
    tot.backward()
    return float(tot.detach()), {k: v.grad.double() for k, v in sdr.items() if v.grad is not None}, time.time() - t0


def device(split, rec_bf):
    ops = importlib.import_module(PKG + ".ops")
    asr = importlib.import_module(PKG + ".src.asr")
    ops.set_gemm_split(split)
    os.environ["ASRK_REC_BF"] = os.environ["ASRK_REC_BF_BWD"] = "1" if rec_bf else "0"
    m = asr.ASR(D, V, True, CFG3_MODEL["ctc_weight"], CFG3_MODEL["encoder"], CFG3_MODEL["attention"],
                CFG3_MODEL["decoder"])
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda").train()
    t = txt.to("cuda")
    ctc_out, enc_len, att_out, _, _ = m(feat.to("cuda"), feat_len.to("cuda"), args.L, tf_rate=1.0, teacher=t)
    txt_len = torch.sum(t != 0, dim=-1)
    ctc = ops.CTCLoss(blank=0)(ctc_out.transpose(0, 1), t, enc_len, txt_len)
    b, tt, _ = att_out.shape
    att = ops.CrossEntropyLoss(ignore_index=0)(att_out.view(b * tt, -1), t.view(-1))
    tot = ctc * m.ctc_weight + att * (1 - m.ctc_weight)
    tot.backward()
    ops.join_deferred()
    ops.check_errors()
    torch.cuda.synchronize()
    ops.set_gemm_split(1)
    return float(tot.detach()), {n: p.grad.double().cpu() for n, p in m.named_parameters() if p.grad is not None}


l64, g64, t64 = host(torch.float64)
rows = {}


def score(name, loss, grads, extra=None):
    num = sum(float((grads[k] - g64[k]).pow(2).sum()) for k in g64)
    den = sum(float(g64[k].pow(2).sum()) for k in g64)
    per = {k: (float((grads[k] - g64[k]).norm()) / max(float(g64[k].norm()), 1e-300)) for k in g64
           if float(g64[k].norm()) > 1e-12}
    worst = max(per, key=per.get)
    rows[name] = {"loss": loss, "loss_rel_err": abs(loss - l64) / abs(l64), "grad_rel_l2_err": (num / den) ** 0.5,
                  "worst_tensor": worst, "worst_tensor_rel_l2_err": per[worst]}
    if extra:
        rows[name].update(extra)
    print("%-28s loss %.9f (rel %.2e)  grad rel-L2 %.3e  worst tensor %.3e (%s)" % (
        name, loss, rows[name]["loss_rel_err"], rows[name]["grad_rel_l2_err"], per[worst], worst), flush=True)


print("float64 oracle: loss %.12f, %d gradient tensors, %.1f s on %d host threads" % (l64, len(g64), t64, args.threads),
      flush=True)
l32, g32, t32 = host(torch.float32)
score("host float32 (ATen CPU)", l32, g32, {"seconds": t32})
score("hip f32-MFMA everywhere", *device(0, False))
score("hip bf16x6 (default)", *device(1, True))
out = {"workload": {"B": args.B, "T": args.T, "L": args.L, "model": "cfg3"}, "truth": {"loss": l64, "seconds": t64},
       "rows": rows}
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
