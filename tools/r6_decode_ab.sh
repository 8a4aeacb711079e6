#!/bin/bash
# one utterance at a time (BeamDecoder.forward, cfg5 widths, T = 800): the device-resident loop against the host record loop
# (ASRK_DECODE_HOST_BEAM=1: one read-back + _expand_beam per decode position, the round-5 path)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for H in 0 1; do
ASRK_DECODE_HOST_BEAM=$H python - <<'PY'
import importlib, json, os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import torch, yaml
import bench
from tools.decode_bench import CFG5_LM, CFG5_DECODE, cfg5_utterance
PKG = "end-to-end-asr-pytorch_amd"
D = importlib.import_module(PKG + ".src.decode"); lm_mod = importlib.import_module(PKG + ".src.lm")
dev = torch.device("cuda"); w = bench.WORKLOADS["cfg3"]; model = bench.build_model(w, dev).eval()
torch.manual_seed(1); tmp = tempfile.mkdtemp()
torch.save({'model': lm_mod.RNNLM(w["V"], **CFG5_LM).state_dict()}, os.path.join(tmp, 'lm.pth'))
yaml.safe_dump({'model': CFG5_LM}, open(os.path.join(tmp, 'lm.yaml'), 'w'))
dec = D.BeamDecoder(model, None, **dict(CFG5_DECODE, lm_path=os.path.join(tmp, 'lm.pth'), lm_config=os.path.join(tmp, 'lm.yaml'))).to(dev)
feat, flen = cfg5_utterance(800); feat, flen = feat.to(dev), flen.to(dev)
with torch.no_grad():
    h = dec(feat, flen); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): h = dec(feat, flen)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(json.dumps({"ASRK_DECODE_HOST_BEAM": os.environ.get("ASRK_DECODE_HOST_BEAM"), "s_per_utt": dt, "utt_per_s": 1 / dt,
                  "positions": len(h[0].outIndex), "ms_per_position": dt * 1e3 / len(h[0].outIndex),
                  "best": h[0].outIndex[:12]}))
PY
done
