"""The convolutional prenets alone at the shipped size (B=16, T=800, D=120): forward + backward of
VGGExtractor / CNNExtractor (src/module.py:7-90) in ms, hipEvent-timed.

    python tools/prenet_bench.py [vgg|cnn] [--steps N]
    rocprofv3 --kernel-trace --stats -d gpurun_out/prenet -- python tools/prenet_bench.py vgg
"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

PKG = "end-to-end-asr-pytorch_amd"


def main():
    which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "vgg"
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 20
    B = int(os.environ.get("PRENET_B", 16)); T = int(os.environ.get("PRENET_T", 800)); D = 120
    mod = importlib.import_module(PKG + ".src.module")
    ops = importlib.import_module(PKG + ".ops")
    dev = torch.device("cuda")
    torch.manual_seed(0)
    net = (mod.VGGExtractor(D) if which == "vgg" else mod.CNNExtractor(D, 1024)).to(dev)
    x = torch.randn(B, T, D, device=dev)
    xl = torch.full((B,), T, device=dev)
    out, _ = net.forward_bm2tm(x, xl)
    dy = torch.randn_like(out)

    def step():
        for p in net.parameters():
            p.grad = None
        o, _ = net.forward_bm2tm(x, xl)
        o.backward(dy)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    fwd = 0.0
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    with torch.no_grad():
        for _ in range(steps):
            net.forward_bm2tm(x, xl)
    e2.record()
    torch.cuda.synchronize()
    ops.check_errors()
    print(json.dumps({"prenet": which, "B": B, "T": T, "D": D, "steps": steps,
                      "fwd_bwd_ms": e0.elapsed_time(e1) / steps, "fwd_ms": e1.elapsed_time(e2) / steps}))


if __name__ == "__main__":
    main()
