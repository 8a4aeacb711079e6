"""Decode benchmark: joint CTC-attention + RNN-LM beam search, batch = 1
utterances of T in {800, 1200, 1600} frames, beam 16, ctc_weight 0.5, lm_weight 0.5, max_len_ratio
0.07 / min_len_ratio 0.01 (config/libri/decode_example.yaml), cfg3 acoustic model + 2xLSTM-1024 LM
over the same 5000-token vocabulary, random-init weights.  Reports wall time per utterance, the
real-time factor (10 ms frames) and utterances/s.   python tools/decode_bench.py"""
import importlib, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, yaml
import bench

PKG = "end-to-end-asr-pytorch_amd"
asr_decode = importlib.import_module(PKG + ".src.decode")
lm_mod = importlib.import_module(PKG + ".src.lm")
ops = importlib.import_module(PKG + ".ops")

dev = torch.device("cuda")
torch.manual_seed(0)
w = bench.WORKLOADS["cfg3"]
model = bench.build_model(w, dev).eval()
V = w["V"]
lm_cfg = dict(emb_tying=False, emb_dim=1024, module='LSTM', dim=1024, n_layers=2, dropout=0.0)
tmp = tempfile.mkdtemp()
lm = lm_mod.RNNLM(V, **lm_cfg)
torch.save({'model': lm.state_dict()}, os.path.join(tmp, 'lm.pth'))
yaml.safe_dump({'model': lm_cfg}, open(os.path.join(tmp, 'lm.yaml'), 'w'))
for tag, kw in (("joint CTC-att + LM", dict(ctc_weight=0.5, lm_weight=0.5, lm_path=os.path.join(tmp, 'lm.pth'),
                                            lm_config=os.path.join(tmp, 'lm.yaml'))),
                ("attention only", dict(ctc_weight=0.0, lm_weight=0.0))):
    dec = asr_decode.BeamDecoder(model, None, beam_size=16, min_len_ratio=0.01, max_len_ratio=0.07, **kw).to(dev)
    for T in (800, 1200, 1600):
        feat = torch.randn(1, T, w["D"], device=dev)
        flen = torch.tensor([T], device=dev)
        with torch.no_grad():
            dec(feat, flen)                                  # warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 3
            for _ in range(n):
                hyps = dec(feat, flen)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("%-20s T=%4d (%.1f s audio): %.3f s/utt  RTF %.4f  %.2f utt/s  best hyp len %d" % (
            tag, T, T * 0.01, dt, dt / (T * 0.01), 1.0 / dt, len(hyps[0].outIndex)))
ops.check_errors()
