"""VGG prenet alone, product (GPU) against the oracle's vgg_forward (ATen on the host), at growing sizes: output,
input gradient and every parameter gradient (max-abs error over max-abs reference).  Diagnosis helper."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "end-to-end-asr-pytorch_amd"
from oracle import asr_oracle as O   # noqa: E402  (checker)

mod = importlib.import_module(PKG + ".src.module")
ops = importlib.import_module(PKG + ".ops")


def run(B, T, D=120, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, D, generator=g)
    xl = torch.full((B,), T, dtype=torch.long)
    vgg = mod.VGGExtractor(D)
    sd = {"e." + k: v.detach().clone() for k, v in vgg.state_dict().items()}
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    yr, _ = O.vgg_forward(sdr, xr, xl, "e.extractor.")
    gy = torch.randn(yr.shape, generator=g)
    (yr * gy).sum().backward()
    vgg = vgg.cuda()
    xd = x.clone().cuda().requires_grad_(True)
    yd, _ = vgg(xd, xl.cuda())
    (yd * gy.cuda()).sum().backward()
    ops.join_deferred()
    torch.cuda.synchronize()

    def err(a, b):
        return float((a.cpu() - b).abs().max() / b.abs().max())
    out = {"y": err(yd.detach(), yr.detach()), "dx": err(xd.grad, xr.grad)}
    for n, p in vgg.named_parameters():
        out[n] = err(p.grad, sdr["e." + n].grad)
    print("B=%d T=%d: " % (B, T) + "  ".join("%s %.1e" % kv for kv in out.items()), flush=True)


for B, T in ((2, 24), (2, 800), (16, 100), (16, 400), (16, 800)):
    run(B, T)
