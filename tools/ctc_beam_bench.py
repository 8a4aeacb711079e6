"""Pure-CTC prefix beam search: time per utterance of the device search (csrc/prefix_beam.hip)
against the host-bookkeeping path it replaces and the CPU oracle (the reference's loop: oracle/ctc_beam_oracle.py),
at the reference's decode configuration (config/libri/ctc_decode_example.yaml: beam 20, 30 candidates) over
V = 5000 symbols and T' = 200 encoder frames (a 16-s utterance at 8x time reduction).

    python tools/ctc_beam_bench.py [--out gpurun_out/ctc_beam.json]
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "end-to-end-asr-pytorch_amd"


class Stub:
    enable_ctc = True

    def __init__(self, V):
        self.vocab_size = V


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--V", type=int, default=5000)
    ap.add_argument("--T", type=int, default=200)
    ap.add_argument("--beam", type=int, default=20)
    ap.add_argument("--cand", type=int, default=30)
    ap.add_argument("--timeline", action="store_true")
    args = ap.parse_args()
    ctc = importlib.import_module(PKG + ".src.ctc")
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(args.T, args.V, generator=g) * 1.5
    logits[:, 0] += 9.0                                         # blank dominates most frames, as in a trained model
    spikes = torch.randperm(args.T, generator=g)[:args.T // 4]  # ~50 emitting frames
    logits[spikes, 0] -= 9.0
    x = torch.log_softmax(logits, -1)
    xd = x.cuda().contiguous()
    vr = [1] + list(range(3, args.V))
    dec = ctc.CTCBeamDecoder(Stub(args.V), vr, args.beam, args.cand)

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, out

    t_dev, h_dev = timed(lambda: dec.search_device(xd), 10)
    if args.timeline:
        import ctypes
        lib = importlib.import_module(PKG + "._lib").load()
        lib.asrk_ctc_prefix_beam_set_debug_.argtypes = [ctypes.c_void_p]
        buf = torch.zeros(64 * 16, dtype=torch.int64, device="cuda")
        lib.asrk_ctc_prefix_beam_set_debug_(ctypes.c_void_p(buf.data_ptr()))
        dec.search_device(xd)
        torch.cuda.synchronize()
        lib.asrk_ctc_prefix_beam_set_debug_(None)
        a = buf.cpu().numpy().reshape(64, 16)
        names = ["rank(0-15)", "P0 tables(15-1)", "P1 rows(1-2)", "P2 expand(2-3)", "P3 string rank(3-4)",
                 "P4a bounds(4-5)", "P4b winners(5-6)", "P5 score rank(6-7)", "P6 rebuild+pairs(7-8)", "tails(8-9)",
                 "publish(9-10)"]
        order = [0, 15, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10]
        d = np.diff(a[20:60][:, order].astype(np.float64), axis=1).mean(0)
        print("phase timeline (wall_clock64 ticks @100 MHz -> us = ticks / 100), mean over frames 20-59:")
        for n, v in zip(names, d):
            print("  %-24s %8.1f us" % (n, v / 100.0))
        print("  frame total              %8.1f us" % ((a[21:60, 0] - a[20:59, 0]).mean() / 100.0))
    # several utterances side by side: one launch each, on its own stream (CTCBeamDecoder.forward_batch does exactly
    # this behind one packed encoder pass): the one-workgroup kernel of every utterance gets its own CU
    def many(U):
        xs = [(xd + 0.01 * u).contiguous() for u in range(U)]          # distinct inputs, same shape
        streams = [torch.cuda.Stream() for _ in range(min(U, 16))]
        main = torch.cuda.current_stream()

        def run():
            fetch = []
            for u in range(U):
                st = streams[u % len(streams)]
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    fetch.append((st, dec.search_device(xs[u], t_start=0, defer=True)))
            out = []
            for st, f in fetch:
                with torch.cuda.stream(st):
                    out.append(f())
            return out
        return timed(run, 3)[0]
    t_many = {U: many(U) for U in (8, 32, 128)}
    t_host, h_host = timed(lambda: dec._search_host(xd), 1)
    # with RNN-LM shallow fusion (the reference's default ctc_decode_example.yaml: lm_weight 0.5, lm_example.yaml:
    # 2 x LSTM-1024 over the same vocabulary): one launch per frame + a batched LM step, nothing read back
    import tempfile
    import yaml
    lm_mod = importlib.import_module(PKG + ".src.lm")
    lm_cfg = dict(emb_tying=False, emb_dim=1024, module="LSTM", dim=1024, n_layers=2, dropout=0.0)
    tmp = tempfile.mkdtemp()
    torch.manual_seed(3)
    lm = lm_mod.RNNLM(args.V, **lm_cfg)
    yaml.safe_dump({"model": lm_cfg}, open(os.path.join(tmp, "lm.yaml"), "w"))
    torch.save({"model": lm.state_dict()}, os.path.join(tmp, "lm.pth"))
    dec_lm = ctc.CTCBeamDecoder(Stub(args.V), vr, args.beam, args.cand, lm_path=os.path.join(tmp, "lm.pth"),
                                lm_config=os.path.join(tmp, "lm.yaml"), lm_weight=0.5, device="cuda")
    with torch.no_grad():
        t_dev_lm, h_dev_lm = timed(lambda: dec_lm.search_device(xd), 3)
        t_host_lm, h_host_lm = timed(lambda: dec_lm._search_host(xd), 1)
    from oracle import ctc_beam_oracle as CBO
    t0 = time.perf_counter()
    h_ref = CBO.prefix_beam_search(x.numpy(), vr, args.beam, args.cand)
    t_ref = time.perf_counter() - t0
    audio_s = args.T * 8 * 0.01
    res = {"config": vars(args), "audio_seconds": audio_s, "hyp_len": len(h_dev[0]),
           "device_search": {"s_per_utt": t_dev, "rtf": t_dev / audio_s, "ms_per_frame": t_dev / args.T * 1e3},
           "device_search_side_by_side": {str(U): {"s_per_batch": t, "utt_per_s": U / t, "rtf": t / (U * audio_s)}
                                          for U, t in t_many.items()},
           "host_bookkeeping_path": {"s_per_utt": t_host, "rtf": t_host / audio_s},
           "cpu_oracle_reference_loop": {"s_per_utt": t_ref, "rtf": t_ref / audio_s, "cores": 1, "kind": "port"},
           "with_lm_2xLSTM1024": {"device_search_s_per_utt": t_dev_lm, "device_rtf": t_dev_lm / audio_s,
                                  "device_ms_per_frame": t_dev_lm / args.T * 1e3,
                                  "host_bookkeeping_s_per_utt": t_host_lm, "host_rtf": t_host_lm / audio_s,
                                  "hypotheses_equal": [list(h) for h in h_dev_lm] == [list(h) for h in h_host_lm]},
           "hypotheses_equal": {"device_vs_host_path": [list(h) for h in h_dev] == [list(h) for h in h_host],
                                "device_vs_oracle": [list(h) for h in h_dev] == [list(h) for h in h_ref]}}
    print(json.dumps(res, indent=1))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
