"""Input-pipeline throughput: frames/s of the whole-batch device front end
(src/audio.py:BatchFeatureTransform: padded int16 PCM -> fbank -> delta -> CMVN -> [B, T, D] zero padded) on a
cfg3-shaped batch (32 utterances x 1600 frames x 80 mel), beside the per-file module chain the collate used
before and the CPU oracle (Kaldi-compliant fbank restatement + the reference's Delta / CMVN / Postprocess) on
one host core, which is how the reference runs it (one file per DataLoader worker, src/data.py:148).
The figure to compare with: the model consumes 4.5e5 frames/s (bench.py).

    python tools/input_pipeline_bench.py [--out gpurun_out/input_pipeline.json]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--T", type=int, default=1600)
    ap.add_argument("--mel", type=int, default=80)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    audio = importlib.import_module(PKG + ".src.audio")
    cfg = dict(feat_type="fbank", feat_dim=args.mel, frame_length=25, frame_shift=10, dither=0, apply_cmvn=True,
               delta_order=0)
    tr, dim = audio.create_transform(dict(cfg))
    n = 400 + 160 * (args.T - 1)
    rng = np.random.RandomState(0)
    pcm = [np.clip(np.round(rng.randn(n) * 3000), -32768, 32767).astype(np.int16) for _ in range(args.B)]
    frames = args.B * args.T

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    res = {"workload": {"B": args.B, "T": args.T, "mel": args.mel, "samples_per_utt": n, "sample_rate": 16000}}
    # whole-batch front end, PCM on the host (includes padding, pinning, the int16 H2D copy)
    t_batch = timed(lambda: tr.batch(pcm, 16000), args.reps)
    res["batch_front_end"] = {"ms_per_batch": t_batch * 1e3, "frames_per_s": frames / t_batch,
                              "includes": "host padding + pinned int16 H2D (%.1f MB) + 7 launches" % (
                                  args.B * n * 2 / 1e6)}
    # device part alone: PCM already resident
    lib = importlib.import_module(PKG + "._lib").load()
    bt = tr.batch
    host = np.stack(pcm)
    wave = torch.from_numpy(host).cuda()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import ctypes
    lib.asrk_profile_reset(); lib.asrk_profile_enable(1)
    tr.batch(pcm, 16000)
    torch.cuda.synchronize()
    lib.asrk_profile_enable(0)
    ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
    fam = {}
    for name, idx in (("gemm", 0), ("fbank_kernels", 7)):
        lib.asrk_profile_get(idx, ctypes.byref(ms), ctypes.byref(cnt))
        fam[name] = {"ms": ms.value, "launches": cnt.value}
    res["batch_front_end"]["kernel_ms"] = fam
    # per-file module chain (what collate did before): one launch chain per utterance
    waves_f = [(torch.from_numpy(x.astype(np.float32) / 32768.0).unsqueeze(0), 16000) for x in pcm]
    t_file = timed(lambda: [tr(w) for w in waves_f], max(2, args.reps // 4))
    res["per_file_chain"] = {"ms_per_batch": t_file * 1e3, "frames_per_s": frames / t_file}
    # CPU oracle, one core, one utterance (the reference extracts per file in DataLoader workers)
    from oracle import fbank_oracle as FO
    torch.set_num_threads(1)
    x64 = pcm[0].astype(np.float64) / 32768.0
    FO.audio_transform(x64, 16000, args.mel, delta_order=0)
    t0 = time.perf_counter()
    for _ in range(3):
        FO.audio_transform(x64, 16000, args.mel, delta_order=0)
    t_cpu = (time.perf_counter() - t0) / 3
    res["cpu_oracle_one_core"] = {"ms_per_utterance": t_cpu * 1e3, "frames_per_s": args.T / t_cpu,
                                  "kind": "port (numpy float64 restatement of kaldi fbank + reference Delta/CMVN)",
                                  "cores": 1, "host_cores": os.cpu_count()}
    print(json.dumps(res, indent=1))
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
