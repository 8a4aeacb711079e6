"""Generate tests/golden/*.npz by running the REAL reference (read-only import from
/root/reference) on seeded inputs.  Runs only in the build container (the GPU box has no
/root/reference); the resulting small fixtures are committed.

    python oracle/gen_golden.py            # writes tests/golden/

Harness-side stubs: `editdistance`, `tensorboard`, `torchaudio`, `matplotlib` are absent here and
only needed by code paths outside the hot path; no reference file is modified.
"""
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def import_reference():
    sys.dont_write_bytecode = True
    for name in ('editdistance', 'torchaudio', 'matplotlib', 'matplotlib.pyplot'):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                m = types.ModuleType(name)
                if name == 'matplotlib':
                    m.use = lambda *a, **k: None
                sys.modules[name] = m
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import src.asr as ref_asr  # noqa
    import src.ctc as ref_ctc  # noqa
    return ref_asr, ref_ctc


def synth_batch(B, T, D, V, L, seed, ragged=True):
    """Synthetic batch per SURVEY.md §8d: feat ~ N(0,1), lengths sorted descending, zero-padded
    tails; txt tokens in [3,V) ending with <eos>=1, 0-padded."""
    g = torch.Generator().manual_seed(seed)
    feat = torch.randn(B, T, D, generator=g)
    if ragged:
        lens = torch.randint(int(0.6 * T), T + 1, (B,), generator=g).sort(descending=True)[0]
        lens[0] = T
    else:
        lens = torch.full((B,), T, dtype=torch.long)
    for b in range(B):
        feat[b, lens[b]:] = 0
    txt = torch.zeros(B, L, dtype=torch.long)
    tl = torch.randint(max(2, L // 2), L + 1, (B,), generator=g)
    tl[0] = L
    for b in range(B):
        n = int(tl[b])
        txt[b, :n - 1] = torch.randint(3, V, (n - 1,), generator=g)
        txt[b, n - 1] = 1
    return feat, lens, txt


CASES = {
    # name: (model cfg, D, V, B, T, L, init_adadelta)
    'enc_ctc_concat': (dict(ctc_weight=1.0,
                            encoder=dict(prenet='', module='LSTM', bidirection=True, dim=[16, 16],
                                         dropout=[0, 0], layer_norm=[False, False],
                                         proj=[False, False], sample_rate=[2, 2],
                                         sample_style='concat'),
                            attention=None, decoder=None), 12, 11, 3, 37, 5, True),
    'enc_ctc_drop_proj': (dict(ctc_weight=1.0,
                               encoder=dict(prenet='', module='LSTM', bidirection=True, dim=[16, 24],
                                            dropout=[0, 0], layer_norm=[False, False],
                                            proj=[True, True], sample_rate=[2, 1],
                                            sample_style='drop'),
                               attention=None, decoder=None), 10, 9, 4, 21, 4, False),
    'las_hybrid_loc': (dict(ctc_weight=0.5,
                            encoder=dict(prenet='', module='LSTM', bidirection=True, dim=[16, 16],
                                         dropout=[0, 0], layer_norm=[False, False],
                                         proj=[False, False], sample_rate=[2, 1],
                                         sample_style='concat'),
                            attention=dict(mode='loc', dim=12, num_head=1, v_proj=False,
                                           temperature=0.5, loc_kernel_size=3, loc_kernel_num=4),
                            decoder=dict(module='LSTM', dim=20, layer=1, dropout=0)),
                       8, 13, 3, 26, 6, True),
    'las_att_dot_mh': (dict(ctc_weight=0.0,
                            encoder=dict(prenet='', module='LSTM', bidirection=False, dim=[16],
                                         dropout=[0], layer_norm=[False], proj=[False],
                                         sample_rate=[2], sample_style='drop'),
                            attention=dict(mode='dot', dim=8, num_head=2, v_proj=True,
                                           temperature=1.0, loc_kernel_size=3, loc_kernel_num=4),
                            decoder=dict(module='LSTM', dim=12, layer=2, dropout=0)),
                       8, 10, 2, 15, 5, False),
    # location-aware attention with TWO heads (prev_att [B,2,T] through Conv1d(2 -> 5, k=2*2+1),
    # src/module.py:229-258) + value projection + merged heads, 2-layer decoder, unidirectional encoder
    'las_loc_mh': (dict(ctc_weight=0.2,
                        encoder=dict(prenet='', module='LSTM', bidirection=False, dim=[20, 16],
                                     dropout=[0, 0], layer_norm=[False, False],
                                     proj=[True, False], sample_rate=[1, 2],
                                     sample_style='drop'),
                        attention=dict(mode='loc', dim=8, num_head=2, v_proj=True,
                                       temperature=2.0, loc_kernel_size=2, loc_kernel_num=5),
                        decoder=dict(module='LSTM', dim=16, layer=2, dropout=0)),
                   7, 12, 3, 22, 5, False),
    # LayerNorm after each recurrent layer (src/module.py:116-117,135-136), odd feature widths
    'enc_ctc_ln': (dict(ctc_weight=1.0,
                        encoder=dict(prenet='', module='LSTM', bidirection=True, dim=[18, 16],
                                     dropout=[0, 0], layer_norm=[True, True],
                                     proj=[True, False], sample_rate=[1, 2],
                                     sample_style='concat'),
                        attention=None, decoder=None), 9, 10, 3, 23, 4, False),
    # VGG prenet (src/module.py:7-66): D=26 -> 2 channels x 13 MFCC bins, T=23 crops to 20 -> 5 frames
    'enc_vgg_ctc': (dict(ctc_weight=1.0,
                         encoder=dict(prenet='vgg', module='LSTM', bidirection=True, dim=[16],
                                      dropout=[0], layer_norm=[False], proj=[False], sample_rate=[1],
                                      sample_style='drop'),
                         attention=None, decoder=None), 26, 8, 2, 23, 2, False),
    # CNN prenet (src/module.py:68-90): two strided Conv1d, no activation
    'enc_cnn_ctc': (dict(ctc_weight=1.0,
                         encoder=dict(prenet='cnn', module='LSTM', bidirection=True, dim=[12, 16],
                                      dropout=[0, 0], layer_norm=[False, False], proj=[False, False],
                                      sample_rate=[1, 1], sample_style='drop'),
                         attention=None, decoder=None), 10, 9, 3, 30, 3, False),
    # unidirectional GRU encoder, LayerNorm + projection, 'drop' pyramid with odd T, CTC only
    'enc_gru_uni_ctc': (dict(ctc_weight=1.0,
                             encoder=dict(prenet='', module='GRU', bidirection=False, dim=[14, 12],
                                          dropout=[0, 0], layer_norm=[True, False],
                                          proj=[True, True], sample_rate=[2, 2],
                                          sample_style='drop'),
                             attention=None, decoder=None), 6, 9, 4, 27, 3, False),
    # GRU everywhere (module: 'GRU'): bidirectional GRU encoder with pyramid, 2-layer GRU decoder
    'las_gru': (dict(ctc_weight=0.3,
                     encoder=dict(prenet='', module='GRU', bidirection=True, dim=[12, 16],
                                  dropout=[0, 0], layer_norm=[False, False], proj=[False, True],
                                  sample_rate=[2, 1], sample_style='concat'),
                     attention=dict(mode='loc', dim=10, num_head=1, v_proj=False,
                                    temperature=0.5, loc_kernel_size=3, loc_kernel_num=4),
                     decoder=dict(module='GRU', dim=16, layer=2, dropout=0)),
                9, 11, 3, 19, 5, False),
}


def run_case(ref_asr, name, spec):
    cfg, D, V, B, T, L, adadelta = spec
    torch.manual_seed(1234 + len(name))
    model = ref_asr.ASR(D, V, adadelta, cfg['ctc_weight'], cfg['encoder'],
                        cfg['attention'] or {}, cfg['decoder'] or {})
    if not adadelta:  # make biases non-trivial
        with torch.no_grad():
            for n, p in model.named_parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn_like(p))
    model.train()
    feat, feat_len, txt = synth_batch(B, T, D, V, L, seed=7 + len(name))
    feat.requires_grad_(True)
    txt_len = torch.sum(txt != 0, dim=-1)
    ctc_out, enc_len, att_out, att_seq, _ = model(feat, feat_len, int(txt_len.max()), tf_rate=1.0,
                                                  teacher=txt)
    total = 0
    out = {}
    if ctc_out is not None:
        ctc_loss = torch.nn.CTCLoss(blank=0, zero_infinity=False)(
            ctc_out.transpose(0, 1), txt, enc_len, txt_len)
        total = total + ctc_loss * model.ctc_weight
        out['ctc_output'] = ctc_out.detach().numpy()
        out['ctc_loss'] = ctc_loss.detach().numpy()
    if att_out is not None:
        b, t, _ = att_out.shape
        att_loss = torch.nn.CrossEntropyLoss(ignore_index=0)(att_out.view(b * t, -1), txt.view(-1))
        total = total + att_loss * (1 - model.ctc_weight)
        out['att_output'] = att_out.detach().numpy()
        out['att_seq'] = att_seq.detach().numpy()
        out['att_loss'] = att_loss.detach().numpy()
    total.backward()
    out['total_loss'] = total.detach().numpy()
    out['encode_len'] = enc_len.numpy()
    out['feat'] = feat.detach().numpy()
    out['feat_len'] = feat_len.numpy()
    out['txt'] = txt.numpy()
    out['grad_feat'] = feat.grad.numpy()
    for n, p in model.named_parameters():
        out['param.' + n] = p.detach().numpy()
        out['grad.' + n] = p.grad.numpy() if p.grad is not None else np.zeros_like(p.detach().numpy())
    # greedy inference (no teacher) for the attention cases: argmax feedback (src/asr.py:136-142)
    if att_out is not None:
        model.eval()
        with torch.no_grad():
            _, _, g_att, _, _ = model(feat.detach(), feat_len, L + 2)
        out['greedy_att_output'] = g_att.numpy()
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print('wrote', name, {k: v.shape for k, v in out.items() if not k.startswith(('param', 'grad.'))})


def inverse_cdf_sample(self, sample_shape=torch.Size()):
    """Deterministic stand-in for Categorical.sample used by BOTH sides of the scheduled-sampling parity test:
    the uniforms come from the default CPU generator (the same one src/asr.py:122 draws its teacher-forcing
    decisions from), so the reference on the host and the HIP path on the GPU take the same draws in the same order."""
    p = self.probs
    u = torch.rand(p.shape[0])
    cdf = p.detach().double().cpu().cumsum(-1)
    idx = torch.searchsorted(cdf, u.double().unsqueeze(-1)).squeeze(-1).clamp(max=p.shape[-1] - 1)
    return idx.to(p.device)


SCHED_CASES = {'sched_las_hybrid_loc': ('las_hybrid_loc', 0.5, 99), 'sched_las_gru': ('las_gru', 0.3, 7)}


def sched_sampling_case(ref_asr, name):
    """src/asr.py:119-135 with 0 < tf_rate < 1: per step a torch.rand(1) decision between the teacher's
    character and a character SAMPLED from the model's own distribution (under no_grad)."""
    from torch.distributions.categorical import Categorical
    base, tf_rate, seed = SCHED_CASES[name]
    cfg, D, V, B, T, L, adadelta = CASES[base]
    torch.manual_seed(1234 + len(base))
    model = ref_asr.ASR(D, V, adadelta, cfg['ctc_weight'], cfg['encoder'], cfg['attention'] or {},
                        cfg['decoder'] or {})
    if not adadelta:
        with torch.no_grad():
            for n, p in model.named_parameters():
                if p.dim() == 1:
                    p.add_(0.1 * torch.randn_like(p))
    model.train()
    feat, feat_len, txt = synth_batch(B, T, D, V, L, seed=7 + len(base))
    feat.requires_grad_(True)
    txt_len = torch.sum(txt != 0, dim=-1)
    orig = Categorical.sample
    Categorical.sample = inverse_cdf_sample
    try:
        torch.manual_seed(seed)
        ctc_out, enc_len, att_out, att_seq, _ = model(feat, feat_len, int(txt_len.max()), tf_rate=tf_rate,
                                                      teacher=txt)
    finally:
        Categorical.sample = orig
    b, t, _ = att_out.shape
    att_loss = torch.nn.CrossEntropyLoss(ignore_index=0)(att_out.view(b * t, -1), txt.view(-1))
    total = att_loss * (1 - model.ctc_weight)
    if ctc_out is not None:
        total = total + torch.nn.CTCLoss(blank=0)(ctc_out.transpose(0, 1), txt, enc_len, txt_len) * model.ctc_weight
    total.backward()
    # how many steps left the teacher: replay the decisions of the same generator
    torch.manual_seed(seed)
    out = {'att_output': att_out.detach().numpy(), 'att_seq': att_seq.detach().numpy(),
           'total_loss': total.detach().numpy(), 'feat': feat.detach().numpy(), 'feat_len': feat_len.numpy(),
           'txt': txt.numpy(), 'grad_feat': feat.grad.numpy(), 'tf_rate': np.float64(tf_rate), 'seed': np.int64(seed)}
    for n, p in model.named_parameters():
        out['param.' + n] = p.detach().numpy()
        out['grad.' + n] = p.grad.numpy() if p.grad is not None else np.zeros_like(p.detach().numpy())
    # the same call under full teacher forcing, to show the golden really left the teacher's path
    with torch.no_grad():
        _, _, att_tf, _, _ = model(feat.detach(), feat_len, int(txt_len.max()), tf_rate=1.0, teacher=txt)
    out['differs_from_teacher_forcing'] = np.float64((att_tf - att_out.detach()).abs().max().item())
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print('wrote', name, 'max |sampled - teacher-forced| =', out['differs_from_teacher_forcing'])


if __name__ == '__main__' and '--sched-sampling' in sys.argv:
    os.makedirs(OUT, exist_ok=True)
    _ref_asr, _ = import_reference()
    for _n in SCHED_CASES:
        sched_sampling_case(_ref_asr, _n)
    sys.exit(0)


def ctc_cases():
    """Standalone torch.nn.CTCLoss vectors incl. repeats, ragged lengths, T == minimal length."""
    g = torch.Generator().manual_seed(99)
    T, B, V, L = 14, 5, 7, 5
    lp = torch.randn(T, B, V, generator=g).log_softmax(-1).requires_grad_(True)
    targets = torch.tensor([[3, 3, 4, 1, 0],      # repeat
                            [2, 5, 1, 0, 0],
                            [6, 6, 6, 6, 1],      # many repeats: needs T >= 5 + 3
                            [1, 0, 0, 0, 0],
                            [4, 2, 4, 2, 1]])
    in_len = torch.tensor([14, 9, 14, 3, 5])      # last: T == L exactly (no repeats)
    tg_len = torch.sum(targets != 0, dim=-1)
    loss = torch.nn.CTCLoss(blank=0, zero_infinity=False)(lp, targets, in_len, tg_len)
    loss.backward()
    nll = torch.nn.functional.ctc_loss(lp, targets, in_len, tg_len, blank=0, reduction='none')
    np.savez_compressed(os.path.join(OUT, 'ctc_loss.npz'), log_probs=lp.detach().numpy(),
                        targets=targets.numpy(), input_lengths=in_len.numpy(),
                        target_lengths=tg_len.numpy(), loss=loss.detach().numpy(),
                        nll=nll.detach().numpy(), grad=lp.grad.numpy())
    print('wrote ctc_loss', float(loss))


def audio_cases():
    """Delta / CMVN / Postprocess of the reference's own src/audio.py classes (torchaudio stubbed:
    only kaldi.fbank needs it) applied to the oracle's fbank of tests/sample_data/*.wav."""
    import src.audio as ref_audio
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import fbank_oracle as FO
    wav = os.path.join(REF, 'tests', 'sample_data', '3830-12529-0005.wav')
    x, sr = FO.read_wav(wav)
    fb = FO.kaldi_fbank(x[0], sr, num_mel_bins=40).astype(np.float32)      # [T, 40]
    inp = torch.from_numpy(fb.T.copy()).unsqueeze(0)                        # [1, D, T]
    out = {'wave_i16': np.round(x[0] * 32768.0).astype(np.int16), 'sample_rate': np.int64(sr),
           'fbank': fb}
    for order in (0, 1, 2):
        mods = ([ref_audio.Delta(order, 2)] if order >= 1 else []) + [ref_audio.CMVN(),
                                                                      ref_audio.Postprocess()]
        y = inp
        for m in mods:
            y = m(y)
        out['post_order%d' % order] = y.numpy()
    y = ref_audio.Postprocess()(ref_audio.Delta(2, 2)(inp))                 # delta without CMVN
    out['delta2_nocmvn'] = y.numpy()
    np.savez_compressed(os.path.join(OUT, 'audio_post.npz'), **out)
    print('wrote audio_post', {k: getattr(v, 'shape', v) for k, v in out.items()})


def main():
    os.makedirs(OUT, exist_ok=True)
    ref_asr, ref_ctc = import_reference()
    if '--case' in sys.argv:          # regenerate a single model case
        name = sys.argv[sys.argv.index('--case') + 1]
        run_case(ref_asr, name, CASES[name])
        return
    if '--audio-only' not in sys.argv:
        for name, spec in CASES.items():
            run_case(ref_asr, name, spec)
        ctc_cases()
        decode_cases()
    audio_cases()




def decode_cases():
    """Reference BeamDecoder (joint CTC-attention [+RNN-LM]) / CTCBeamDecoder / CTCPrefixScore on the
    golden models' weights: hypotheses + scores (src/decode.py:64-173, src/ctc.py:76-116,241-352)."""
    import tempfile
    import yaml
    import src.asr as ref_asr
    import src.ctc as ref_ctc
    import src.decode as ref_decode
    import src.lm as ref_lm
    out = {}
    # ---- prefix scorer vectors
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 17, 9, generator=g).log_softmax(-1)
    ps = ref_ctc.CTCPrefixScore(x)
    r0 = ps.init_state()
    psi1, r1 = ps.cheap_compute([], r0, [3, 1, 5, 8])
    psi2, r2 = ps.cheap_compute([3], r1[0], [3, 4, 1, 2])       # last char in candidates
    psi3, r3 = ps.cheap_compute([3, 3], r2[0], [1, 7, 3])
    out.update(ps_x=x.numpy(), ps_r0=r0, ps_psi1=psi1, ps_r1=r1, ps_psi2=psi2, ps_r2=r2,
               ps_psi3=psi3, ps_r3=r3)
    # ---- joint beam search on the las_hybrid_loc golden model
    name = 'las_hybrid_loc'
    cfg, D, V, B, T, L, adadelta = CASES[name]
    gold = np.load(os.path.join(OUT, name + '.npz'))
    model = ref_asr.ASR(D, V, adadelta, cfg['ctc_weight'], cfg['encoder'], cfg['attention'], cfg['decoder'])
    model.load_state_dict({k[6:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith('param.')})
    model.eval()
    feat = torch.from_numpy(gold['feat'])[:1]
    flen = torch.from_numpy(gold['feat_len'])[:1]
    lm_cfg = dict(emb_tying=False, emb_dim=10, module='LSTM', dim=14, n_layers=2, dropout=0.0)
    torch.manual_seed(77)
    lm = ref_lm.RNNLM(V, **lm_cfg)
    tmp = tempfile.mkdtemp()
    lm_yaml, lm_ckpt = os.path.join(tmp, 'lm.yaml'), os.path.join(tmp, 'lm.pth')
    yaml.safe_dump({'model': lm_cfg}, open(lm_yaml, 'w'))
    torch.save({'model': lm.state_dict()}, lm_ckpt)
    for k, v in lm.state_dict().items():
        out['lm.' + k] = v.numpy()
    for tag, kw in (('beam_ctc', dict(beam_size=3, ctc_weight=0.4)),
                    ('beam_att', dict(beam_size=4, ctc_weight=0.0)),
                    ('beam_ctc_lm', dict(beam_size=3, ctc_weight=0.4, lm_weight=0.3,
                                         lm_path=lm_ckpt, lm_config=lm_yaml))):
        dec = ref_decode.BeamDecoder(model, None, min_len_ratio=0.01, max_len_ratio=0.5, **kw)
        with torch.no_grad():
            hyps = dec(feat, flen)
        for i, h in enumerate(hyps):
            out['%s.hyp%d' % (tag, i)] = np.asarray(h.outIndex, np.int64)
            out['%s.score%d' % (tag, i)] = np.asarray([float(s) for s in h.output_scores], np.float32)
        out[tag + '.n'] = np.int64(len(hyps))
        print(tag, [h.outIndex for h in hyps])
    # ---- pure CTC beam search on the enc_ctc_concat golden model
    name = 'enc_ctc_concat'
    cfg, D, V, B, T, L, adadelta = CASES[name]
    gold = np.load(os.path.join(OUT, name + '.npz'))
    model = ref_asr.ASR(D, V, adadelta, cfg['ctc_weight'], cfg['encoder'], {}, {})
    model.load_state_dict({k[6:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith('param.')})
    model.eval()
    feat = torch.from_numpy(gold['feat'])[:1]
    flen = torch.from_numpy(gold['feat_len'])[:1]
    dec = ref_ctc.CTCBeamDecoder(model, [1] + list(range(3, V)), beam_size=3, vocab_candidate=4)
    with torch.no_grad():
        hy = dec(feat, flen)
    for i, y in enumerate(hy):
        out['ctcbeam.hyp%d' % i] = np.asarray(y, np.int64)
    out['ctcbeam.n'] = np.int64(len(hy))
    print('ctcbeam', hy)
    np.savez_compressed(os.path.join(OUT, 'decode.npz'), **out)


def decode_cases_more():
    """joint CTC-attention beam search on two more golden models (2-head location-aware attention
    with a 2-layer LSTM decoder; GRU encoder + 2-layer GRU decoder) -> tests/golden/decode_more.npz"""
    import_reference()
    import src.asr as ref_asr
    import src.decode as ref_decode
    out = {}
    for name, kw in (('las_loc_mh', dict(beam_size=3, ctc_weight=0.3)),
                     ('las_loc_mh', dict(beam_size=4, ctc_weight=0.0)),
                     ('las_gru', dict(beam_size=3, ctc_weight=0.0))):   # (GRU + CTC weight: the
        # reference itself raises ValueError in addTopk on this random-weight model)
        cfg, D, V, B, T, L, adadelta = CASES[name]
        gold = np.load(os.path.join(OUT, name + '.npz'))
        model = ref_asr.ASR(D, V, adadelta, cfg['ctc_weight'], cfg['encoder'], cfg['attention'], cfg['decoder'])
        model.load_state_dict({k[6:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith('param.')})
        model.eval()
        feat = torch.from_numpy(gold['feat'])[1:2]
        flen = torch.from_numpy(gold['feat_len'])[1:2]
        tag = '%s.b%d.w%d' % (name, kw['beam_size'], int(10 * kw['ctc_weight']))
        dec = ref_decode.BeamDecoder(model, None, min_len_ratio=0.01, max_len_ratio=0.6, **kw)
        with torch.no_grad():
            hyps = dec(feat, flen)
        for i, h in enumerate(hyps):
            out['%s.hyp%d' % (tag, i)] = np.asarray(h.outIndex, np.int64)
            out['%s.score%d' % (tag, i)] = np.asarray([float(s) for s in h.output_scores], np.float32)
        out[tag + '.n'] = np.int64(len(hyps))
        print(tag, [h.outIndex for h in hyps])
    np.savez_compressed(os.path.join(OUT, 'decode_more.npz'), **out)


CFG3_MODEL = dict(ctc_weight=0.5,
                  encoder=dict(prenet='', module='LSTM', bidirection=True, dim=[1024] * 4,
                               dropout=[0] * 4, layer_norm=[False] * 4, proj=[False] * 4,
                               sample_rate=[2, 2, 2, 1], sample_style='concat'),
                  attention=dict(mode='loc', dim=300, num_head=1, v_proj=False, temperature=0.5,
                                 loc_kernel_size=100, loc_kernel_num=10),
                  decoder=dict(module='LSTM', dim=1024, layer=1, dropout=0))
# BASELINE configs[1] ("cfg2": 2 x pBLSTM-512 concat, CTC only - with ctc_weight == 1 the reference builds no decoder,
# src/asr.py:22-23,30) and configs[0] = the architecture the reference SHIPS (config/libri/asr_example.yaml:34-59: VGG
# prenet on 40 fbank + delta + delta-delta, 5 x BLSTM-512 each followed by Linear + tanh, location-aware attention,
# LSTM-512 decoder, attention only, subword-16k vocabulary, batch 16)
CFG2_MODEL = dict(ctc_weight=1.0,
                  encoder=dict(prenet='', module='LSTM', bidirection=True, dim=[512, 512], dropout=[0, 0],
                               layer_norm=[False, False], proj=[False, False], sample_rate=[2, 2],
                               sample_style='concat'),
                  attention={}, decoder={})
SHIPPED_MODEL = dict(ctc_weight=0.0,
                     encoder=dict(prenet='vgg', module='LSTM', bidirection=True, dim=[512] * 5, dropout=[0] * 5,
                                  layer_norm=[False] * 5, proj=[True] * 5, sample_rate=[1] * 5, sample_style='drop'),
                     attention=dict(mode='loc', dim=300, num_head=1, v_proj=False, temperature=0.5,
                                    loc_kernel_size=100, loc_kernel_num=10),
                     decoder=dict(module='LSTM', dim=512, layer=1, dropout=0))
# the same architecture behind the reference's OTHER prenet (src/module.py:68-90, `prenet: 'cnn'`: two strided Conv1d,
# 120 -> 512 -> 512, no activation, time / 4): the CNN prenet at the shipped model's own size
CNN_MODEL = dict(SHIPPED_MODEL, encoder=dict(SHIPPED_MODEL["encoder"], prenet='cnn'))
CFG5_LM = dict(emb_tying=False, emb_dim=1024, module='LSTM', dim=1024, n_layers=2, dropout=0.0)
CFG5_DECODE = dict(beam_size=16, min_len_ratio=0.01, max_len_ratio=0.07, ctc_weight=0.5, lm_weight=0.5)


def cfg5_weights(V=5000, D=80, seed=11, peak=4.0):
    """BASELINE configs[2]/[4] architecture (config/libri/asr_example.yaml:34-54 widths,
    decode_example.yaml:11-17 search settings) with seeded random weights; the output layers are
    scaled by `peak` so the random model's token posteriors are not flat (well separated beams)."""
    from oracle import asr_oracle as O
    from oracle import beam_oracle as BO
    sd = O.make_state_dict(CFG3_MODEL, D, V, seed=seed)
    lm_sd = BO.make_lm_state_dict(V, CFG5_LM, seed=seed + 1)
    for k in ('decoder.char_trans.weight', 'ctc_layer.weight'):
        sd[k] = sd[k] * peak
    lm_sd['trans.weight'] = lm_sd['trans.weight'] * peak
    lm_sd['emb.weight'] = lm_sd['emb.weight'] * 0.1
    return sd, lm_sd


def cfg5_utterance(T, D=80, seed=5):
    g = torch.Generator().manual_seed(seed + T)
    return torch.randn(1, T, D, generator=g), torch.tensor([T])


def decode_cfg5_cases(T_list=(800, 1600)):
    """REAL reference BeamDecoder at BASELINE configs[4] widths (4 x pBLSTM-1024, loc attention 300 /
    201 taps x 10, LSTM-1024 decoder, V=5000, beam 16, CTC 0.5, 2 x LSTM-1024 LM 0.5) on seeded
    weights -> tests/golden/decode_cfg5.npz (hypotheses + per-token scores only; the weights are
    regenerated from the seed by cfg5_weights)."""
    import tempfile
    import time
    import yaml
    import_reference()
    import src.asr as ref_asr
    import src.decode as ref_decode
    V, D = 5000, 80
    sd, lm_sd = cfg5_weights(V, D)
    model = ref_asr.ASR(D, V, True, CFG3_MODEL['ctc_weight'], CFG3_MODEL['encoder'],
                        CFG3_MODEL['attention'], CFG3_MODEL['decoder'])
    model.load_state_dict(sd, strict=True)
    model.eval()
    tmp = tempfile.mkdtemp()
    lm_yaml, lm_ckpt = os.path.join(tmp, 'lm.yaml'), os.path.join(tmp, 'lm.pth')
    yaml.safe_dump({'model': CFG5_LM}, open(lm_yaml, 'w'))
    torch.save({'model': lm_sd}, lm_ckpt)
    out = {}
    for T in T_list:
        feat, flen = cfg5_utterance(T)
        for tag, kw in (('T%d.lm' % T, dict(CFG5_DECODE, lm_path=lm_ckpt, lm_config=lm_yaml)),
                        ('T%d.nolm' % T, dict(CFG5_DECODE, lm_weight=0.0))):
            dec = ref_decode.BeamDecoder(model, None, **kw)
            t0 = time.time()
            with torch.no_grad():
                hyps = dec(feat, flen)
            for i, h in enumerate(hyps):
                out['%s.hyp%d' % (tag, i)] = np.asarray(h.outIndex, np.int64)
                out['%s.score%d' % (tag, i)] = np.asarray([float(s) for s in h.output_scores], np.float32)
            out[tag + '.n'] = np.int64(len(hyps))
            print(tag, '%.1f s' % (time.time() - t0), hyps[0].outIndex[:12], float(hyps[0].avgScore()))
    np.savez_compressed(os.path.join(OUT, 'decode_cfg5.npz'), **out)


if __name__ == '__main__' and '--decode-cfg5' in sys.argv:
    decode_cfg5_cases()
    sys.exit(0)


def ctc_beam_lm_cases():
    """reference CTCBeamDecoder WITH an RNN-LM (src/ctc.py:241-352, lm_weight > 0) on the enc_ctc_concat
    golden model -> tests/golden/ctcbeam_lm.npz (hypotheses + the LM weights)"""
    import tempfile
    import yaml
    import_reference()
    import src.asr as ref_asr
    import src.ctc as ref_ctc
    import src.lm as ref_lm
    name = 'enc_ctc_concat'
    cfg, D, V, B, T, L, adadelta = CASES[name]
    gold = np.load(os.path.join(OUT, name + '.npz'))
    model = ref_asr.ASR(D, V, adadelta, cfg['ctc_weight'], cfg['encoder'], {}, {})
    model.load_state_dict({k[6:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith('param.')})
    model.eval()
    out = {}
    tmp = tempfile.mkdtemp()
    for tag, lm_cfg in (('lstm', dict(emb_tying=False, emb_dim=6, module='LSTM', dim=9, n_layers=1, dropout=0.0)),
                        ('gru', dict(emb_tying=True, emb_dim=8, module='GRU', dim=8, n_layers=2, dropout=0.0))):
        torch.manual_seed(31)
        lm = ref_lm.RNNLM(V, **lm_cfg)
        with torch.no_grad():
            for p_ in lm.parameters():
                p_.mul_(3.0)                       # make the LM opinionated enough to change the ranking
        lm_yaml, lm_ckpt = os.path.join(tmp, tag + '.yaml'), os.path.join(tmp, tag + '.pth')
        yaml.safe_dump({'model': lm_cfg}, open(lm_yaml, 'w'))
        torch.save({'model': lm.state_dict()}, lm_ckpt)
        for k, v in lm.state_dict().items():
            out['%s.lm.%s' % (tag, k)] = v.numpy()
        for u in (0, 1, 2):
            feat = torch.from_numpy(gold['feat'])[u:u + 1]
            flen = torch.from_numpy(gold['feat_len'])[u:u + 1]
            dec = ref_ctc.CTCBeamDecoder(model, [1] + list(range(3, V)), beam_size=4, vocab_candidate=5,
                                         lm_path=lm_ckpt, lm_config=lm_yaml, lm_weight=0.6, device='cpu')
            with torch.no_grad():
                hy = dec(feat, flen)
            for i, y in enumerate(hy):
                out['%s.u%d.hyp%d' % (tag, u, i)] = np.asarray(y, np.int64)
            out['%s.u%d.n' % (tag, u)] = np.int64(len(hy))
            print(tag, u, hy)
    np.savez_compressed(os.path.join(OUT, 'ctcbeam_lm.npz'), **out)


if __name__ == '__main__' and '--ctc-lm' in sys.argv:
    ctc_beam_lm_cases()
    sys.exit(0)


# ---- pure-CTC prefix beam search at realistic widths: the SEARCH of the real reference class on given CTC logits
CTC_BEAM_BIG = {
    # name: (V, T, beam, cand, seed, hot symbols whose logits are boosted, lm_cfg or None, lm_weight)
    # 'collide': symbols whose decimal strings concatenate ambiguously ([3,45] / [34,5] / [345] ...): the string
    # sort of src/ctc.py:320 then interleaves DIFFERENT sequences with EQUAL keys
    'collide': (400, 40, 8, 10, 5, [3, 4, 5, 34, 45, 345, 53, 334], None, 0.0),
    'wide': (5000, 120, 20, 30, 6, None, None, 0.0),                 # ctc_decode_example.yaml: beam 20, cand 30
    'eos': (60, 50, 6, 8, 7, [1, 7, 11, 17, 1], None, 0.0),          # <eos> = 1 likely: finished hypotheses stay
    'lm': (300, 50, 6, 8, 8, [3, 30, 33, 7], dict(emb_tying=False, emb_dim=8, module='LSTM', dim=12,
                                                 n_layers=1, dropout=0.0), 0.5),
    'lm_gru': (300, 40, 5, 7, 9, None, dict(emb_tying=True, emb_dim=10, module='GRU', dim=10, n_layers=2,
                                             dropout=0.0), 0.7),
}


def ctc_beam_big_logits(name):
    """the logits the stub model returns ([T, V] float32; the reference applies log_softmax to them)"""
    V, T, beam, cand, seed, hot, lm_cfg, lm_w = CTC_BEAM_BIG[name]
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(T, V, generator=g) * 1.5
    x[:, 0] += 3.0                                                  # blank dominates, as in a trained model
    x[:3, 0] += 8.0                                                 # leading all-blank frames (src/ctc.py:265)
    if hot:
        for k in hot:
            x[:, k] += 3.5 * torch.rand(T, generator=g)
    return x


def ctc_beam_big_cases():
    """run the REAL reference CTCBeamDecoder (src/ctc.py:210-352) on a stub acoustic model that returns given
    logits -> tests/golden/ctcbeam_big.npz: hypotheses (+ the LM weights of the LM cases)"""
    import tempfile
    import yaml
    import_reference()
    import src.ctc as ref_ctc
    import src.lm as ref_lm

    class StubASR:
        enable_ctc = True

        def __init__(self, logits, V):
            self.logits, self.vocab_size = logits, V

        def __call__(self, feat, feat_len, step):
            return self.logits.unsqueeze(0), None, None, None, None

    out = {}
    tmp = tempfile.mkdtemp()
    for name, (V, T, beam, cand, seed, hot, lm_cfg, lm_w) in CTC_BEAM_BIG.items():
        logits = ctc_beam_big_logits(name)
        kw = {}
        if lm_cfg is not None:
            torch.manual_seed(100 + seed)
            lm = ref_lm.RNNLM(V, **lm_cfg)
            with torch.no_grad():
                for p_ in lm.parameters():
                    p_.mul_(2.5)
            lm_yaml, lm_ckpt = os.path.join(tmp, name + '.yaml'), os.path.join(tmp, name + '.pth')
            yaml.safe_dump({'model': lm_cfg}, open(lm_yaml, 'w'))
            torch.save({'model': lm.state_dict()}, lm_ckpt)
            for k, v in lm.state_dict().items():
                out['%s.lm.%s' % (name, k)] = v.numpy()
            kw = dict(lm_path=lm_ckpt, lm_config=lm_yaml, lm_weight=lm_w, device='cpu')
        dec = ref_ctc.CTCBeamDecoder(StubASR(logits, V), [1] + list(range(3, V)), beam, cand, **kw)
        with torch.no_grad():
            hy = dec(torch.zeros(1, 4, 2), torch.tensor([4]))
        out[name + '.n'] = np.int64(len(hy))
        for i, y in enumerate(hy):
            out['%s.hyp%d' % (name, i)] = np.asarray(y, np.int64)
        print(name, len(hy), [len(y) for y in hy], hy[:2])
    np.savez_compressed(os.path.join(OUT, 'ctcbeam_big.npz'), **out)


if __name__ == '__main__' and '--ctc-beam-big' in sys.argv:
    ctc_beam_big_cases()
    sys.exit(0)


def prefix_full_cases():
    """CTCPrefixScore.full_compute (src/ctc.py:37-74: every token as continuation, no <eos>
    override) chained over three prefixes -> tests/golden/prefix_full.npz"""
    ref_asr, ref_ctc = import_reference()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 17, 9, generator=g).log_softmax(-1)
    ps = ref_ctc.CTCPrefixScore(x)
    r0 = ps.init_state()
    psi1, r1 = ps.full_compute([], r0)
    psi2, r2 = ps.full_compute([3], r1[3])
    psi3, r3 = ps.full_compute([3, 3], r2[3])
    psi4, r4 = ps.full_compute([3, 3, 7], r3[7])
    np.savez_compressed(os.path.join(OUT, 'prefix_full.npz'), x=x.numpy(), r0=r0, psi1=psi1, r1=r1,
                        psi2=psi2, r2=r2, psi3=psi3, r3=r3, psi4=psi4, r4=r4)
    print('wrote prefix_full', psi1.shape, r1.shape)


if __name__ == '__main__' and '--prefix-full' in sys.argv:
    os.makedirs(OUT, exist_ok=True)
    prefix_full_cases()


if __name__ == '__main__' and '--decode-more' in sys.argv:
    os.makedirs(OUT, exist_ok=True)
    decode_cases_more()


if __name__ == '__main__' and '--decode-only' in sys.argv:
    os.makedirs(OUT, exist_ok=True)
    import_reference()
    decode_cases()


def host_cases():
    """Host-logic vectors from the real reference: lr / teacher-forcing schedules (src/optim.py),
    text encoders (src/text.py), LibriDataset ordering + bucketing (corpus/librispeech.py:30-66) and
    collect_audio_batch ordering / halving / padding (src/data.py:14-46) -> tests/golden/host.json"""
    import json
    import tempfile
    import_reference()
    import src.optim as ref_optim
    import src.text as ref_text
    import src.data as ref_data
    from corpus.librispeech import LibriDataset as RefLibri
    out = {}
    steps = [0, 1, 10, 499, 500, 3999, 4000, 4001, 19999, 20000, 50000, 80000, 80001, 200000]
    out['steps'] = steps
    p = [torch.nn.Parameter(torch.zeros(2))]
    for sch in ('warmup', 'spec-aug-basic', 'spec-aug-double', 'fixed'):
        o = ref_optim.Optimizer(p, 'Adam', 0.001, 1e-8, sch, tf_start=1.0, tf_end=0.6, tf_step=5000)
        lrs, tfs = [], []
        for s in steps:
            tfs.append(float(o.pre_step(s)))
            lrs.append(float(o.opt.param_groups[0]['lr']))
        out['lr.' + sch] = lrs
        out['tf'] = tfs
    # ---- text encoders
    tmp = tempfile.mkdtemp()
    cv = os.path.join(tmp, 'char.txt')
    with open(cv, 'w') as f:
        f.write('\n'.join([' ', "'", 'A', 'B', 'C', 'D', 'E', 'H', 'L', 'O', 'R', 'T', 'W']) + '\n')
    wv = os.path.join(tmp, 'word.txt')
    with open(wv, 'w') as f:
        f.write('\n'.join(['HELLO', 'WORLD', 'THE', 'CAT']) + '\n')
    sents = ['HELLO WORLD', "THE CAT'S  HAT\n", ' A B ', 'XYZ', '']
    ids = [[3, 3, 4, 0, 4, 4, 1, 5], [5, 5, 5], [0, 0, 6, 6, 0, 6, 2, 1], [1, 4]]
    for mode, vf in (('character', cv), ('word', wv)):
        enc = ref_text.load_text_encoder(mode, vf)
        out['text.%s.vocab_size' % mode] = enc.vocab_size
        out['text.%s.encode' % mode] = [enc.encode(s) for s in sents]
        out['text.%s.decode' % mode] = [enc.decode(i) for i in ids]
        out['text.%s.decode_norepeat' % mode] = [enc.decode(i, ignore_repeat=True) for i in ids]
    out['text.sents'], out['text.ids'] = sents, ids
    out['text.char_vocab'] = open(cv).read()
    out['text.word_vocab'] = open(wv).read()
    # ---- dataset ordering / bucketing on an (empty-file) LibriSpeech layout
    enc = ref_text.load_text_encoder('character', cv)
    root = os.path.join(tmp, 'corpus')
    trans = {'train-a': {('11', '22'): ['HELLO', 'A CAT', 'THE WORLD HELLO', 'BE'],
                         ('11', '23'): ['HOLD THE DOOR', 'O']},
             'dev-a': {('31', '41'): ['LATER', 'HELLO WORLD']}}
    for split, chapters in trans.items():
        for (spk, ch), lines in chapters.items():
            d = os.path.join(root, split, spk, ch)
            os.makedirs(d)
            with open(os.path.join(d, '%s-%s.trans.txt' % (spk, ch)), 'w') as f:
                for i, l in enumerate(lines):
                    f.write('%s-%s-%04d %s\n' % (spk, ch, i, l))
                    open(os.path.join(d, '%s-%s-%04d.flac' % (spk, ch, i)), 'w').close()
    out['corpus.trans'] = {s: {'-'.join(k): v for k, v in c.items()} for s, c in trans.items()}
    for asc in (False, True):
        ds = RefLibri(root, ['train-a'], enc, 1, ascending=asc)
        # ties in text length follow filesystem listing order in the reference: record sets per length
        out['libri.asc%d.lens' % asc] = [len(t) for t in ds.text]
        out['libri.asc%d.pairs' % asc] = sorted([[str(f).split('/')[-1].split('.')[0], list(t)]
                                                 for f, t in zip(ds.file_list, ds.text)])
    ds = RefLibri(root, ['train-a'], enc, 4)
    out['libri.bucket4.len'] = len(ds)
    out['libri.bucket4.item0_lens'] = [len(t) for _, t in ds[0]]
    out['libri.bucket4.item5_lens'] = [len(t) for _, t in ds[5]]
    # ---- collate: stub transform = deterministic "features" whose length depends on the file name
    flen = {'u0': 700, 'u1': 820, 'u2': 300, 'u3': 820, 'u4': 10, 'u5': 555}

    def fake_transform(path):
        n = flen[str(path).split('/')[-1].split('.')[0]]
        return torch.arange(n * 2, dtype=torch.float32).view(n, 2) + n
    batch = [('/x/u%d.flac' % i, [3 + i] * (6 - i) + [1]) for i in range(6)]
    for tag, b, mode in (('tr', batch, 'train'), ('tr_half', batch[1:], 'train'), ('dv', batch[1:], 'test'),
                         ('bucket', [batch], 'train')):
        names, feat, alen, txt = ref_data.collect_audio_batch(b, fake_transform, mode)
        out['collate.%s' % tag] = dict(names=list(names), feat_shape=list(feat.shape), alen=alen.tolist(),
                                       txt=txt.tolist(), feat_sum=float(feat.double().sum()))
    out['collate.flen'] = flen
    # ---- text-only dataset (LM training, corpus/librispeech.py:69-125) and its collate (src/data.py:46-61)
    from corpus.librispeech import LibriTextDataset as RefLibriText
    import corpus.librispeech as ref_libri_mod
    tds = RefLibriText(root, ['train-a', 'dev-a'], enc, 1)
    out['libritext.lens'] = [len(t) for t in tds.text]
    out['libritext.sorted_texts'] = sorted(list(t) for t in tds.text)
    tdb = RefLibriText(root, ['train-a'], enc, 3)
    out['libritext.bucket3.len'] = len(tdb)
    out['libritext.bucket3.item0_lens'] = [len(t) for t in tdb[0]]
    out['libritext.bucket3.item5_lens'] = [len(t) for t in tdb[5]]      # clamps to the last full bucket
    # the official text file is encoded lazily and loses its REMOVE_TOP_N_TXT longest lines
    with open(os.path.join(root, 'librispeech-lm-norm.txt'), 'w') as f:
        f.write('THE CAT\nA\nHELLO HELLO HELLO WORLD\nBE THE DOOR\nO HOLD\n')
    keep = ref_libri_mod.REMOVE_TOP_N_TXT
    ref_libri_mod.REMOVE_TOP_N_TXT = 2
    try:
        tdo = RefLibriText(root, ['librispeech-lm-norm.txt', 'dev-a'], enc, 2)
        out['libritext.official.len'] = len(tdo)
        out['libritext.official.item0'] = [list(t) for t in tdo[0]]
        out['libritext.official.item9'] = [list(t) for t in tdo[9]]
    finally:
        ref_libri_mod.REMOVE_TOP_N_TXT = keep
    out['libritext.official.remove_top'] = 2
    tb = [[5, 6, 7, 1], [8, 1], [3, 3, 3, 3, 3, 1], [4, 1]]
    long_first = [[2] * 151 + [1]] + tb
    for tag, b, mode in (('plain', tb, 'train'), ('bucket', [tb], 'train'),
                         ('half', long_first, 'train'), ('nohalf_test', long_first, 'test')):
        t = ref_data.collect_text_batch(b, mode)
        out['textcollate.%s' % tag] = dict(shape=list(t.shape), sum=int(t.sum()), first=t[0, :6].tolist(),
                                           last=t[-1, :6].tolist())
    out['textcollate.inputs'] = dict(tb=tb, long_len=152)
    # ---- create_dataset / create_textset interface (src/data.py:64-135): loader batch sizes, mode,
    #      bucket sizes and the messages shown, for the bucketing / ascending / test-mode combinations
    def ds_summary(r):
        a, b, bs_a, bs_b, mode, msg = r
        return dict(len_a=len(a), len_b=len(b), bucket_a=a.bucket_size, bucket_b=b.bucket_size,
                    bs_a=bs_a, bs_b=bs_b, mode=mode, msg=[m.replace(root, '<root>') for m in msg])
    combos = {'train_bucket': dict(ascending=False, bucketing=True, train_split=['train-a'], dev_split=['dev-a']),
              'train_plain': dict(ascending=False, bucketing=False, train_split=['train-a'], dev_split=['dev-a']),
              'train_asc_bucket': dict(ascending=True, bucketing=True, train_split=['train-a'], dev_split=['dev-a']),
              'test': dict(ascending=False, bucketing=True, dev_split=['dev-a'], test_split=['train-a'])}
    for tag, kw in combos.items():
        kw = dict(kw)
        r = ref_data.create_dataset(enc, kw.pop('ascending'), 'librispeech', root, kw.pop('bucketing'), 3, **kw)
        out['create_dataset.' + tag] = ds_summary(r)
    for tag, bucketing in (('bucket', True), ('plain', False)):
        a, b, bs_a, bs_b, msg = ref_data.create_textset(enc, ['train-a'], ['dev-a'], 'librispeech', root,
                                                        bucketing, 3)
        out['create_textset.' + tag] = dict(len_a=len(a), len_b=len(b), bucket_a=a.bucket_size,
                                            bucket_b=b.bucket_size, bs_a=bs_a, bs_b=bs_b,
                                            msg=[m.replace(root, '<root>') for m in msg])
    # ---- from-seed initialisation (src/asr.py:41-46, src/util.py:47-77; default torch init when the
    #      optimiser is not Adadelta): SHA-1 of every parameter after torch.manual_seed(3), all model
    #      cases x both modes, and the two language-model cases
    import hashlib
    import src.asr as ref_asr_mod
    import src.lm as ref_lm_mod

    def digest(sd):
        return {k: hashlib.sha1(v.detach().contiguous().numpy().tobytes()).hexdigest() for k, v in sd.items()}
    for name, (cfg, D, V, B, T, L, adadelta) in CASES.items():
        for mode in (True, False):
            torch.manual_seed(3)
            m = ref_asr_mod.ASR(D, V, mode, cfg['ctc_weight'], cfg['encoder'], cfg['attention'] or {},
                                cfg['decoder'] or {})
            out['init.%s.%d' % (name, mode)] = digest(m.state_dict())
    for tag, lm_cfg in (('lstm', dict(emb_tying=False, emb_dim=8, module='LSTM', dim=12, n_layers=2, dropout=0.0)),
                        ('gru', dict(emb_tying=True, emb_dim=12, module='GRU', dim=12, n_layers=1, dropout=0.0))):
        torch.manual_seed(3)
        out['init.lm.' + tag] = digest(ref_lm_mod.RNNLM(13, **lm_cfg).state_dict())
    # ---- the model / LM / optimiser summaries the solvers print (create_msg)
    for name, (cfg, D, V, B, T, L, adadelta) in CASES.items():
        out['msg.' + name] = ref_asr_mod.ASR(D, V, True, cfg['ctc_weight'], cfg['encoder'], cfg['attention'] or {},
                                             cfg['decoder'] or {}).create_msg()
    out['msg.lm'] = ref_lm_mod.RNNLM(13, False, 8, 'LSTM', 12, 2, 0.0).create_msg()
    out['msg.lm_tied'] = ref_lm_mod.RNNLM(13, True, 12, 'GRU', 12, 1, 0.0).create_msg()
    for tag, kw in (('adadelta_tf', dict(optimizer='Adadelta', lr=1.0, eps=1e-8, lr_scheduler='fixed',
                                         tf_start=1, tf_end=0.5, tf_step=100)),
                    ('adam_warmup', dict(optimizer='Adam', lr=1e-3, eps=1e-8, lr_scheduler='warmup'))):
        out['msg.optim.' + tag] = ref_optim.Optimizer([torch.nn.Parameter(torch.zeros(2))], **kw).create_msg()
    # ---- subword (sentencepiece BPE) text encoder, src/text.py:96-133: a 40-piece model trained here
    #      (identity normalisation keeps the file at ~0.5 KB) is committed next to the vectors
    import sentencepiece as splib
    spm_txt = os.path.join(tmp, 'spm_corpus.txt')
    with open(spm_txt, 'w') as f:
        f.write('\n'.join(['HELLO WORLD', 'THE CAT HOLD THE DOOR', 'A CAT BE LATER', 'HELLO HELLO THE WORLD',
                           'O HOLD THE DOOR LATER', 'BE THE CAT'] * 4) + '\n')
    splib.SentencePieceTrainer.train(input=spm_txt, model_prefix=os.path.join(OUT, 'spm_tiny'), vocab_size=40,
                                     model_type='bpe', pad_id=0, eos_id=1, unk_id=2, bos_id=-1,
                                     eos_piece='<eos>', normalization_rule_name='identity',
                                     character_coverage=1.0, hard_vocab_limit=False, minloglevel=2)
    os.remove(os.path.join(OUT, 'spm_tiny.vocab'))
    if not hasattr(splib.SentencePieceProcessor, 'set_encode_extra_options'):
        # this sentencepiece release dropped the call the reference makes (src/text.py:123); the
        # harness restores its documented meaning (":eos" = append </s> to every encoding)
        _enc_ids = splib.SentencePieceProcessor.encode_as_ids

        def _set_opts(self, opt):
            self._harness_eos = ':eos' in opt

        def _encode_as_ids(self, text):
            ids = list(_enc_ids(self, text))
            return ids + [self.eos_id()] if getattr(self, '_harness_eos', False) else ids
        splib.SentencePieceProcessor.set_encode_extra_options = _set_opts
        splib.SentencePieceProcessor.encode_as_ids = _encode_as_ids
    sub = ref_text.load_text_encoder('subword', os.path.join(OUT, 'spm_tiny.model'))
    sub_sents = ['HELLO WORLD', 'THE CAT', 'A DOOR LATER BE', 'O', 'ZEBRA HELLO']   # Z, R-less words -> <unk>
    sub_ids = [[14, 14, 13, 0, 13, 1, 9], [5, 5, 5, 0, 0], [1, 4], [20, 21, 21, 22, 1]]
    out['text.subword.vocab_size'] = sub.vocab_size
    out['text.subword.token_type'] = sub.token_type
    out['text.subword.encode'] = [sub.encode(t) for t in sub_sents]
    out['text.subword.decode'] = [sub.decode(i) for i in sub_ids]
    out['text.subword.decode_norepeat'] = [sub.decode(i, ignore_repeat=True) for i in sub_ids]
    out['text.subword.sents'], out['text.subword.ids'] = sub_sents, sub_ids
    # ---- src/util.py: initialisers, number formatting, batch error rate
    import src.util as ref_util
    import torch.nn as nn

    def lev(a, b):   # the harness stands in for the absent `editdistance` package (textbook DP)
        prev = list(range(len(b) + 1))
        for i, x in enumerate(a, 1):
            cur = [i]
            for j, y in enumerate(b, 1):
                cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
            prev = cur
        return prev[-1]
    ref_util.ed.eval = lev
    out['util.human_format'] = {str(n): ref_util.human_format(n) for n in (0, 7, 999, 1000, 15300, 2500000,
                                                                          81234567890)}
    out['util.init_gate'] = ref_util.init_gate(torch.arange(12, dtype=torch.float32) * 0.1).tolist()
    torch.manual_seed(9)
    mods = nn.Sequential(nn.Embedding(5, 3), nn.Linear(3, 4), nn.Conv1d(2, 3, 3), nn.Conv2d(1, 2, 3))
    mods.apply(ref_util.init_weights)
    out['util.init_weights'] = {k: v.flatten().tolist() for k, v in mods.state_dict().items()}
    g = torch.Generator().manual_seed(4)
    logits = torch.randn(3, 9, enc.vocab_size, generator=g)
    ids = logits.argmax(-1)
    truth = torch.tensor([enc.encode('HELLO WORLD') + [0, 0], enc.encode('A CAT BE') + [0] * 5,
                          enc.encode('THE DOOR') + [0] * 5])[:, :13]
    out['util.cal_er.logits'] = logits.flatten().tolist()
    out['util.cal_er.truth'] = truth.tolist()
    for mode in ('wer', 'cer'):
        for ctc in (False, True):
            out['util.cal_er.%s.ctc%d.3d' % (mode, ctc)] = ref_util.cal_er(enc, logits, truth, mode=mode, ctc=ctc)
            out['util.cal_er.%s.ctc%d.2d' % (mode, ctc)] = ref_util.cal_er(enc, ids, truth, mode=mode, ctc=ctc)
    with open(os.path.join(OUT, 'host.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('wrote host.json', len(out), 'keys')


if __name__ == '__main__' and '--host-only' in sys.argv:
    host_cases()


def lm_cases():
    """Reference RNNLM (src/lm.py) whole-sequence training step as bin/train_lm.py:62-70 runs it
    (<sos>-prefixed ragged batch, CrossEntropyLoss(ignore_index=0)): logits, loss, gradients for an
    LSTM and a GRU language model -> tests/golden/lm_train.npz"""
    import_reference()
    import src.lm as ref_lm
    out = {}
    V = 13
    g = torch.Generator().manual_seed(21)
    data = torch.zeros(4, 7, dtype=torch.long)
    for b, n in enumerate((7, 5, 4, 2)):                      # ragged, <eos>=1 terminated
        data[b, :n - 1] = torch.randint(3, V, (n - 1,), generator=g)
        data[b, n - 1] = 1
    out['data'] = data.numpy()
    for tag, cfg in (('lstm', dict(emb_tying=False, emb_dim=8, module='LSTM', dim=12, n_layers=2, dropout=0.0)),
                     ('gru', dict(emb_tying=True, emb_dim=12, module='GRU', dim=12, n_layers=1, dropout=0.0))):
        torch.manual_seed(5)
        lm = ref_lm.RNNLM(V, **cfg)
        lm.train()
        txt = torch.cat((torch.zeros((data.shape[0], 1), dtype=torch.long), data), dim=1)
        txt_len = torch.sum(data != 0, dim=-1)
        pred, _ = lm(txt[:, :-1], txt_len)
        loss = torch.nn.CrossEntropyLoss(ignore_index=0)(pred.view(-1, V), txt[:, 1:].reshape(-1))
        loss.backward()
        out[tag + '.pred'] = pred.detach().numpy()
        out[tag + '.loss'] = loss.detach().numpy()
        for n, p in lm.named_parameters():
            out['%s.param.%s' % (tag, n)] = p.detach().numpy()
            out['%s.grad.%s' % (tag, n)] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'lm_train.npz'), **out)
    print('wrote lm_train', {k: v.shape for k, v in out.items() if 'param' not in k and 'grad' not in k})


if __name__ == '__main__' and '--lm-only' in sys.argv:
    lm_cases()


# (last: main() uses functions defined further up AND down the file)
if __name__ == '__main__' and not ({'--decode-only', '--decode-more', '--prefix-full', '--host-only', '--lm-only', '--decode-cfg5', '--ctc-lm', '--ctc-beam-big', '--sched-sampling'} & set(sys.argv)):
    main()
