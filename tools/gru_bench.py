"""Time one bidirectional GRU encoder layer (forward + backward) through the persistent kernels and
through the per-step host loop.  usage: python tools/gru_bench.py [T B Din H]"""
import importlib
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("end-to-end-asr-pytorch_amd")
gru = importlib.import_module("end-to-end-asr-pytorch_amd.gru_ops")
ops = importlib.import_module("end-to-end-asr-pytorch_amd.ops")

T, B, Din, H = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (400, 32, 1024, 512)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
shapes = [(3 * H, Din), (3 * H, H), (3 * H,), (3 * H,)]
pf = tuple((torch.randn(*s, generator=g) * 0.05).to(dev).requires_grad_(True) for s in shapes)
pr = tuple((torch.randn(*s, generator=g) * 0.05).to(dev).requires_grad_(True) for s in shapes)
x = torch.randn(T, B, Din, generator=g).to(dev).requires_grad_(True)
dy = torch.randn(T, B, 2 * H, generator=g).to(dev)


def run(n):
    for _ in range(n):
        for q in pf + pr + (x,):
            q.grad = None
        y = gru.gru_layer(x, pf, pr)
        y.backward(dy)
    ops.join_deferred()
    torch.cuda.synchronize()


out = {"T": T, "B": B, "Din": Din, "H": H}
for mode in ("1", "0"):
    os.environ["ASRK_GRU_PERSISTENT"] = mode
    run(2)
    t0 = time.perf_counter()
    n = 5 if mode == "1" else 2
    run(n)
    out["persistent" if mode == "1" else "host_loop"] = round((time.perf_counter() - t0) / n * 1e3, 3)
print(json.dumps(out))
