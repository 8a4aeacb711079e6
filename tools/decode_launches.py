"""Kernel launches of ONE utterance's beam search at the cfg5 widths, per decode position (run under
rocprofv3 --kernel-trace --stats; the stats CSV divided by the printed position count gives launches per position).

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/decl -o d -- python tools/decode_launches.py [U]
"""
import importlib, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, yaml
PKG = "end-to-end-asr-pytorch_amd"
import bench
from tools.decode_bench import CFG5_LM, CFG5_DECODE, cfg5_utterance

U = int(sys.argv[1]) if len(sys.argv) > 1 else 1
REPS = 3
D = importlib.import_module(PKG + ".src.decode")
lm_mod = importlib.import_module(PKG + ".src.lm")
dev = torch.device("cuda")
w = bench.WORKLOADS["cfg3"]
model = bench.build_model(w, dev).eval()
torch.manual_seed(1)
tmp = tempfile.mkdtemp()
torch.save({'model': lm_mod.RNNLM(w["V"], **CFG5_LM).state_dict()}, os.path.join(tmp, 'lm.pth'))
yaml.safe_dump({'model': CFG5_LM}, open(os.path.join(tmp, 'lm.yaml'), 'w'))
dec = D.BeamDecoder(model, None, **dict(CFG5_DECODE, lm_path=os.path.join(tmp, 'lm.pth'),
                                        lm_config=os.path.join(tmp, 'lm.yaml'))).to(dev)
feat = torch.stack([cfg5_utterance(800, seed=5 + u)[0][0] for u in range(U)]).to(dev)
flen = torch.tensor([800] * U).to(dev)
with torch.no_grad():
    dec.forward_batch(feat, flen)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        hyps = dec.forward_batch(feat, flen)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / REPS
steps = max(len(h[0].outIndex) for h in hyps)
print(json.dumps({"utterances": U, "decodes": REPS + 1, "positions_per_decode": steps, "ms_per_position": dt * 1e3 / steps}))
