"""Decode benchmark (BASELINE.json configs[4] / SURVEY.md §8(d) cfg5): joint CTC-attention + RNN-LM beam
search, batch = 1 utterances of T in {800, 1200, 1600} frames, beam 16, ctc_weight 0.5 (24 candidates),
lm_weight 0.5, max_len_ratio 0.07 / min_len_ratio 0.01 (config/libri/decode_example.yaml), cfg3 acoustic
model + 2 x LSTM-1024 LM over the same 5000-token vocabulary, seeded random-init weights (parity of the
hypotheses at these widths is the job of tests/test_decode_gpu.py, against tests/golden/decode_cfg5.npz).

    python tools/decode_bench.py [--cpu-baseline]   -> one JSON line (RTF = decode time / audio time at
                                                       10 ms frames; utt/s; ms per decode step)

Multi-GPU decoding is utterance-sharded replicas (bin/test_asr.py, parallel.shard_indices): N GPUs decode
N utterances at once with no exchange, so the N-GPU rate is N x the single-GPU rate measured here.
`cpu_baseline` (kind "port"): oracle/beam_oracle.py - the restatement of the reference's BeamDecoder that
tests/test_beam_oracle_cpu.py pins hypothesis-for-hypothesis on the real reference - on the host cores.
"""
import importlib, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, yaml

PKG = "end-to-end-asr-pytorch_amd"


CFG5_LM = dict(emb_tying=False, emb_dim=1024, module='LSTM', dim=1024, n_layers=2, dropout=0.0)
CFG5_DECODE = dict(beam_size=16, min_len_ratio=0.01, max_len_ratio=0.07, ctc_weight=0.5, lm_weight=0.5)


def cfg5_utterance(T, D=80, seed=5):
    g = torch.Generator().manual_seed(seed + T)
    return torch.randn(1, T, D, generator=g), torch.tensor([T])


def main():
    import bench
    asr_decode = importlib.import_module(PKG + ".src.decode")
    lm_mod = importlib.import_module(PKG + ".src.lm")
    ops = importlib.import_module(PKG + ".ops")
    dev = torch.device("cuda")
    w = bench.WORKLOADS["cfg3"]
    CFG3_MODEL = w["model"]
    model = bench.build_model(w, dev).eval()
    torch.manual_seed(1)
    lm_sd = lm_mod.RNNLM(w["V"], **CFG5_LM).state_dict()
    tmp = tempfile.mkdtemp()
    torch.save({'model': lm_sd}, os.path.join(tmp, 'lm.pth'))
    yaml.safe_dump({'model': CFG5_LM}, open(os.path.join(tmp, 'lm.yaml'), 'w'))
    out = {"metric": "joint CTC-attention beam-search decode (beam 16) + RNN-LM shallow fusion", "unit": "utt/s",
           "n_gpus": 1, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "cfg5: %s + 2xLSTM-1024 LM, V=5000" % json.dumps(CFG5_DECODE)},
           "results": []}
    for tag, kw in (("joint_ctc_att_lm", dict(CFG5_DECODE, lm_path=os.path.join(tmp, 'lm.pth'),
                                              lm_config=os.path.join(tmp, 'lm.yaml'))),
                    ("joint_ctc_att", dict(CFG5_DECODE, lm_weight=0.0)),
                    ("attention_only", dict(CFG5_DECODE, lm_weight=0.0, ctc_weight=0.0))):
        dec = asr_decode.BeamDecoder(model, None, **kw).to(dev)
        for T in (800, 1200, 1600):
            feat, flen = cfg5_utterance(T)
            feat, flen = feat.to(dev), flen.to(dev)
            with torch.no_grad():
                dec(feat, flen)                                  # warm-up
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 5
                for _ in range(n):
                    hyps = dec(feat, flen)
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            steps = len(hyps[0].outIndex)
            out["results"].append({"mode": tag, "T": T, "audio_s": T * 0.01, "s_per_utt": dt,
                                   "rtf": dt / (T * 0.01), "utt_per_s": 1.0 / dt, "decode_steps": steps,
                                   "ms_per_decode_step": dt * 1e3 / steps})
    ops.check_errors()
    head = [r for r in out["results"] if r["mode"] == "joint_ctc_att_lm"]
    out["value_one_utterance_at_a_time"] = sum(r["utt_per_s"] for r in head) / len(head)
    out["rtf_one_utterance_at_a_time"] = sum(r["rtf"] for r in head) / len(head)
    # ---- several utterances per device step (BeamDecoder.forward_batch): the reference fans utterances out over CPU
    # worker processes (bin/test_asr.py:163-167); here U utterances' beams are rows of one device batch
    dec = asr_decode.BeamDecoder(model, None, **dict(CFG5_DECODE, lm_path=os.path.join(tmp, 'lm.pth'),
                                                     lm_config=os.path.join(tmp, 'lm.yaml'))).to(dev)
    out["batched"] = []
    for U, Ts in ((8, [800] * 8), (16, [800] * 16), (32, [800] * 32),
                  (16, [800, 1200, 1600, 1000] * 4), (32, [800, 1200, 1600, 1000, 600, 1400, 900, 1100] * 4)):
        feat = torch.zeros(U, max(Ts), w["D"])
        for u, T in enumerate(Ts):
            feat[u, :T] = cfg5_utterance(T, seed=5 + u)[0][0]
        feat, flen = feat.to(dev), torch.tensor(Ts).to(dev)
        with torch.no_grad():
            dec.forward_batch(feat, flen)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                hyps = dec.forward_batch(feat, flen)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        audio = sum(Ts) * 0.01
        out["batched"].append({"mode": "joint_ctc_att_lm", "utterances_per_batch": U,
                               "T": Ts[0] if len(set(Ts)) == 1 else "mixed %d..%d" % (min(Ts), max(Ts)),
                               "audio_s": audio, "s_per_batch": dt, "rtf": dt / audio, "utt_per_s": U / dt,
                               "decode_steps": max(len(h[0].outIndex) for h in hyps)})
    ops.check_errors()
    best = max(out["batched"], key=lambda r: r["utt_per_s"] if r["T"] == 800 else 0.0)
    out["value"] = best["utt_per_s"]
    out["rtf"] = best["rtf"]
    out["value_note"] = "utt/s of %d utterances of 8 s per device batch (forward_batch)" % best["utterances_per_batch"]
    if "--cpu-baseline" in sys.argv:
        from oracle import beam_oracle as BO          # checker-side code: CPU baseline leg only
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        feat, flen = cfg5_utterance(800)
        t0 = time.perf_counter()
        BO.beam_search(sd, CFG3_MODEL, feat, flen, lm_sd=lm_sd, lm_cfg=CFG5_LM, lstm_impl="aten", **CFG5_DECODE)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 1.0 / dt, "unit": "utt/s", "rtf": dt / 8.0, "cores": torch.get_num_threads(),
                               "kind": "port", "sample": "one T=800 (8 s) utterance, joint CTC-att + LM, beam 16: "
                               "oracle/beam_oracle.py (per-hypothesis loop like src/decode.py:103-167), %.2f s" % dt}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
