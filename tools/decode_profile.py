"""Host-side profile of one cfg5 beam-search decode (cProfile, top cumulative) - where the per-step
time goes when the device kernels are short.   python tools/decode_profile.py [T]"""
import cProfile, importlib, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, yaml
import bench

PKG = "end-to-end-asr-pytorch_amd"
asr_decode = importlib.import_module(PKG + ".src.decode")
lm_mod = importlib.import_module(PKG + ".src.lm")
dev = torch.device("cuda")
torch.manual_seed(0)
w = bench.WORKLOADS["cfg3"]
model = bench.build_model(w, dev).eval()
lm_cfg = dict(emb_tying=False, emb_dim=1024, module='LSTM', dim=1024, n_layers=2, dropout=0.0)
tmp = tempfile.mkdtemp()
torch.save({'model': lm_mod.RNNLM(w["V"], **lm_cfg).state_dict()}, os.path.join(tmp, 'lm.pth'))
yaml.safe_dump({'model': lm_cfg}, open(os.path.join(tmp, 'lm.yaml'), 'w'))
dec = asr_decode.BeamDecoder(model, None, beam_size=16, min_len_ratio=0.01, max_len_ratio=0.07, ctc_weight=0.5,
                             lm_weight=0.5, lm_path=os.path.join(tmp, 'lm.pth'),
                             lm_config=os.path.join(tmp, 'lm.yaml')).to(dev)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 800
feat, flen = torch.randn(1, T, w["D"], device=dev), torch.tensor([T], device=dev)
with torch.no_grad():
    dec(feat, flen)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    dec(feat, flen)
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
