"""Host-side plumbing (text encoders, lr / teacher-forcing schedules, dataset ordering, batch
collation, CLI surface) against tests/golden/host.json — vectors produced by the REAL reference
(oracle/gen_golden.py --host-only).  Exact equality for ids / strings / orderings, 1e-12 relative
for the float schedules.  No GPU, no HIP library calls."""
import importlib
import json
import os

import numpy as np
import pytest
import torch

PKG = "end-to-end-asr-pytorch_amd"
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def gold():
    with open(os.path.join(HERE, 'golden', 'host.json')) as f:
        return json.load(f)


def _mod(name):
    return importlib.import_module(PKG + '.' + name)


def test_schedules_match_reference(gold):
    optim = _mod('src.optim')
    p = [torch.nn.Parameter(torch.zeros(2))]
    for sch in ('warmup', 'spec-aug-basic', 'spec-aug-double', 'fixed'):
        o = optim.Optimizer(p, 'Adam', 0.001, 1e-8, sch, tf_start=1.0, tf_end=0.6, tf_step=5000)
        for s, lr_ref, tf_ref in zip(gold['steps'], gold['lr.' + sch], gold['tf']):
            tf = o.pre_step(s)
            assert tf == pytest.approx(tf_ref, rel=1e-12)
            assert o.opt.param_groups[0]['lr'] == pytest.approx(lr_ref, rel=1e-12)
    assert o.create_msg()[0].startswith('Optim.spec.| Algo. = Adam')


def test_text_encoders_match_reference(gold, tmp_path):
    text = _mod('src.text')
    files = {}
    for mode, key in (('character', 'text.char_vocab'), ('word', 'text.word_vocab')):
        files[mode] = str(tmp_path / (mode + '.txt'))
        with open(files[mode], 'w') as f:
            f.write(gold[key])
    for mode in ('character', 'word'):
        enc = text.load_text_encoder(mode, files[mode])
        assert enc.token_type == mode
        assert enc.vocab_size == gold['text.%s.vocab_size' % mode]
        assert (enc.pad_idx, enc.eos_idx, enc.unk_idx) == (0, 1, 2)
        assert [enc.encode(s) for s in gold['text.sents']] == gold['text.%s.encode' % mode]
        assert [enc.decode(i) for i in gold['text.ids']] == gold['text.%s.decode' % mode]
        assert [enc.decode(i, ignore_repeat=True) for i in gold['text.ids']] == \
            gold['text.%s.decode_norepeat' % mode]
    with pytest.raises(NotImplementedError):
        text.load_text_encoder('bert-base-uncased', '')


def _make_corpus(root, trans, suffix):
    for split, chapters in trans.items():
        for key, lines in chapters.items():
            spk, ch = key.split('-')
            d = os.path.join(root, split, spk, ch)
            os.makedirs(d)
            with open(os.path.join(d, '%s-%s.trans.txt' % (spk, ch)), 'w') as f:
                for i, l in enumerate(lines):
                    f.write('%s-%s-%04d %s\n' % (spk, ch, i, l))
                    open(os.path.join(d, '%s-%s-%04d.%s' % (spk, ch, i, suffix)), 'w').close()


@pytest.mark.parametrize('suffix', ['wav', 'flac'])
def test_libri_dataset_order_and_buckets(gold, tmp_path, suffix):
    text, libri = _mod('src.text'), _mod('corpus.librispeech')
    vf = str(tmp_path / 'char.txt')
    with open(vf, 'w') as f:
        f.write(gold['text.char_vocab'])
    enc = text.load_text_encoder('character', vf)
    root = str(tmp_path / 'corpus')
    _make_corpus(root, gold['corpus.trans'], suffix)
    for asc in (0, 1):
        ds = libri.LibriDataset(root, ['train-a'], enc, 1, ascending=bool(asc))
        assert [len(t) for t in ds.text] == gold['libri.asc%d.lens' % asc]
        pairs = sorted([[str(f).split('/')[-1].split('.')[0], list(t)] for f, t in zip(ds.file_list, ds.text)])
        assert pairs == gold['libri.asc%d.pairs' % asc]
        f0, t0 = ds[0]
        assert str(f0).endswith(suffix) and list(t0) == list(ds.text[0])
    ds = libri.LibriDataset(root, ['train-a'], enc, 4)
    assert len(ds) == gold['libri.bucket4.len']
    assert [len(t) for _, t in ds[0]] == gold['libri.bucket4.item0_lens']
    assert [len(t) for _, t in ds[5]] == gold['libri.bucket4.item5_lens']      # tail clamps
    with pytest.raises(AssertionError):
        libri.LibriDataset(root, ['no-such-split'], enc, 1)


def test_libri_text_dataset_matches_reference(gold, tmp_path, monkeypatch):
    """text-only dataset for LM training: length ordering, bucket indexing with tail clamp, lazy
    encoding of the official text file minus its REMOVE_TOP_N_TXT longest lines"""
    text, libri = _mod('src.text'), _mod('corpus.librispeech')
    vf = str(tmp_path / 'char.txt')
    with open(vf, 'w') as f:
        f.write(gold['text.char_vocab'])
    enc = text.load_text_encoder('character', vf)
    root = str(tmp_path / 'corpus')
    _make_corpus(root, gold['corpus.trans'], 'flac')
    tds = libri.LibriTextDataset(root, ['train-a', 'dev-a'], enc, 1)
    assert [len(tds[i]) for i in range(len(tds))] == gold['libritext.lens']
    assert sorted(list(tds[i]) for i in range(len(tds))) == gold['libritext.sorted_texts']
    tdb = libri.LibriTextDataset(root, ['train-a'], enc, 3)
    assert len(tdb) == gold['libritext.bucket3.len']
    assert [len(t) for t in tdb[0]] == gold['libritext.bucket3.item0_lens']
    assert [len(t) for t in tdb[5]] == gold['libritext.bucket3.item5_lens']
    with open(os.path.join(root, 'librispeech-lm-norm.txt'), 'w') as f:
        f.write('THE CAT\nA\nHELLO HELLO HELLO WORLD\nBE THE DOOR\nO HOLD\n')
    monkeypatch.setattr(libri, 'REMOVE_TOP_N_TXT', gold['libritext.official.remove_top'])
    tdo = libri.LibriTextDataset(root, ['librispeech-lm-norm.txt', 'dev-a'], enc, 2)
    assert len(tdo) == gold['libritext.official.len']
    assert [list(t) for t in tdo[0]] == gold['libritext.official.item0']
    assert [list(t) for t in tdo[9]] == gold['libritext.official.item9']


def test_text_collate_matches_reference(gold):
    data = _mod('src.data')
    tb = gold['textcollate.inputs']['tb']
    long_first = [[2] * (gold['textcollate.inputs']['long_len'] - 1) + [1]] + tb
    for tag, b, mode in (('plain', tb, 'train'), ('bucket', [tb], 'train'),
                         ('half', long_first, 'train'), ('nohalf_test', long_first, 'test')):
        t = data.collect_text_batch(b, mode)
        g = gold['textcollate.' + tag]
        assert t.dtype == torch.int64 and list(t.shape) == g['shape']
        assert int(t.sum()) == g['sum']
        assert t[0, :6].tolist() == g['first'] and t[-1, :6].tolist() == g['last']


def test_create_dataset_interfaces_match_reference(gold, tmp_path):
    """src/data.py create_dataset / create_textset: loader batch sizes, bucket sizes, mode and the
    messages shown, for bucketing x ascending and for test mode"""
    text, data = _mod('src.text'), _mod('src.data')
    vf = str(tmp_path / 'char.txt')
    with open(vf, 'w') as f:
        f.write(gold['text.char_vocab'])
    enc = text.load_text_encoder('character', vf)
    root = str(tmp_path / 'corpus')
    _make_corpus(root, gold['corpus.trans'], 'flac')
    combos = {'train_bucket': dict(ascending=False, bucketing=True, train_split=['train-a'], dev_split=['dev-a']),
              'train_plain': dict(ascending=False, bucketing=False, train_split=['train-a'], dev_split=['dev-a']),
              'train_asc_bucket': dict(ascending=True, bucketing=True, train_split=['train-a'], dev_split=['dev-a']),
              'test': dict(ascending=False, bucketing=True, dev_split=['dev-a'], test_split=['train-a'])}
    for tag, kw in combos.items():
        kw = dict(kw)
        a, b, bs_a, bs_b, mode, msg = data.create_dataset(enc, kw.pop('ascending'), 'librispeech', root,
                                                          kw.pop('bucketing'), 3, **kw)
        g = gold['create_dataset.' + tag]
        assert (len(a), len(b), a.bucket_size, b.bucket_size, bs_a, bs_b, mode) == \
            (g['len_a'], g['len_b'], g['bucket_a'], g['bucket_b'], g['bs_a'], g['bs_b'], g['mode']), tag
        assert [m.replace(root, '<root>') for m in msg] == g['msg'], tag
    for tag, bucketing in (('bucket', True), ('plain', False)):
        a, b, bs_a, bs_b, msg = data.create_textset(enc, ['train-a'], ['dev-a'], 'librispeech', root, bucketing, 3)
        g = gold['create_textset.' + tag]
        assert (len(a), len(b), a.bucket_size, b.bucket_size, bs_a, bs_b) == \
            (g['len_a'], g['len_b'], g['bucket_a'], g['bucket_b'], g['bs_a'], g['bs_b']), tag
        assert [m.replace(root, '<root>') for m in msg] == g['msg'], tag
    with pytest.raises(NotImplementedError):
        data.create_dataset(enc, False, 'no-such-corpus', root, True, 3, train_split=['train-a'], dev_split=['dev-a'])


def test_from_seed_initialisation_is_bit_identical_to_reference(gold):
    """Same seed -> the same initial parameters as the reference, key for key and bit for bit, with
    and without the Adadelta re-initialisation (src/asr.py:41-46), for every golden model case and
    the two language models: a run started from a seed starts from the reference's weights."""
    import hashlib
    from helpers import CASES
    asr, lm = _mod('src.asr'), _mod('src.lm')

    def digest(sd):
        return {k: hashlib.sha1(v.detach().contiguous().numpy().tobytes()).hexdigest() for k, v in sd.items()}
    for name, (cfg, D, V, B, T, L, adadelta) in CASES.items():
        for mode in (True, False):
            torch.manual_seed(3)
            m = asr.ASR(D, V, mode, cfg['ctc_weight'], cfg['encoder'], cfg['attention'] or {},
                        cfg['decoder'] or {})
            ref = gold['init.%s.%d' % (name, mode)]
            got = digest(m.state_dict())
            assert list(got.keys()) == list(ref.keys()), (name, mode)
            assert got == ref, (name, mode, [k for k in ref if got[k] != ref[k]])
    for tag, lm_cfg in (('lstm', dict(emb_tying=False, emb_dim=8, module='LSTM', dim=12, n_layers=2, dropout=0.0)),
                        ('gru', dict(emb_tying=True, emb_dim=12, module='GRU', dim=12, n_layers=1, dropout=0.0))):
        torch.manual_seed(3)
        got = digest(lm.RNNLM(13, **lm_cfg).state_dict())
        assert got == gold['init.lm.' + tag], tag


def test_create_msg_texts_match_reference(gold):
    from helpers import CASES
    asr, lm, optim = _mod('src.asr'), _mod('src.lm'), _mod('src.optim')
    for name, (cfg, D, V, B, T, L, adadelta) in CASES.items():
        m = asr.ASR(D, V, True, cfg['ctc_weight'], cfg['encoder'], cfg['attention'] or {}, cfg['decoder'] or {})
        assert m.create_msg() == gold['msg.' + name], name
    assert lm.RNNLM(13, False, 8, 'LSTM', 12, 2, 0.0).create_msg() == gold['msg.lm']
    assert lm.RNNLM(13, True, 12, 'GRU', 12, 1, 0.0).create_msg() == gold['msg.lm_tied']
    for tag, kw in (('adadelta_tf', dict(optimizer='Adadelta', lr=1.0, eps=1e-8, lr_scheduler='fixed',
                                         tf_start=1, tf_end=0.5, tf_step=100)),
                    ('adam_warmup', dict(optimizer='Adam', lr=1e-3, eps=1e-8, lr_scheduler='warmup'))):
        assert optim.Optimizer([torch.nn.Parameter(torch.zeros(2))], **kw).create_msg() == gold['msg.optim.' + tag]


def test_subword_text_encoder_matches_reference(gold):
    """sentencepiece BPE encoder on the committed 40-piece model (tests/golden/spm_tiny.model):
    encodings end with <eos>=1, decode stops at <eos>, drops pads and (optionally) repeats"""
    text = _mod('src.text')
    model = os.path.join(os.path.dirname(__file__), 'golden', 'spm_tiny.model')
    enc = text.load_text_encoder('subword', model)
    assert enc.vocab_size == gold['text.subword.vocab_size']
    assert enc.token_type == gold['text.subword.token_type']
    assert [list(enc.encode(t)) for t in gold['text.subword.sents']] == gold['text.subword.encode']
    assert [enc.decode(i) for i in gold['text.subword.ids']] == gold['text.subword.decode']
    assert [enc.decode(i, ignore_repeat=True) for i in gold['text.subword.ids']] == \
        gold['text.subword.decode_norepeat']


def test_util_functions_match_reference(gold, tmp_path):
    """src/util.py: human_format, init_gate, init_weights (same RNG stream -> bit-identical tensors),
    cal_er (wer / cer, CTC repeat merging, 3-D logits and 2-D id input)"""
    import torch.nn as nn
    util, text = _mod('src.util'), _mod('src.text')
    for n, ref in gold['util.human_format'].items():
        assert util.human_format(int(n)) == ref
    assert util.init_gate(torch.arange(12, dtype=torch.float32) * 0.1).tolist() == gold['util.init_gate']
    torch.manual_seed(9)
    mods = nn.Sequential(nn.Embedding(5, 3), nn.Linear(3, 4), nn.Conv1d(2, 3, 3), nn.Conv2d(1, 2, 3))
    mods.apply(util.init_weights)
    for k, v in mods.state_dict().items():
        assert v.flatten().tolist() == gold['util.init_weights'][k], k
    vf = str(tmp_path / 'char.txt')
    with open(vf, 'w') as f:
        f.write(gold['text.char_vocab'])
    enc = text.load_text_encoder('character', vf)
    logits = torch.tensor(gold['util.cal_er.logits']).view(3, 9, enc.vocab_size)
    truth = torch.tensor(gold['util.cal_er.truth'])
    for mode in ('wer', 'cer'):
        for ctc in (False, True):
            for tag, pred in (('3d', logits), ('2d', logits.argmax(-1))):
                got = util.cal_er(enc, pred, truth, mode=mode, ctc=ctc)
                assert abs(got - gold['util.cal_er.%s.ctc%d.%s' % (mode, int(ctc), tag)]) < 1e-12, (mode, ctc, tag)
    assert np.isnan(util.cal_er(enc, None, truth))


def test_collate_matches_reference(gold, monkeypatch):
    data = _mod('src.data')
    flen = gold['collate.flen']

    def fake_transform(path):
        n = flen[str(path).split('/')[-1].split('.')[0]]
        return torch.arange(n * 2, dtype=torch.float32).view(n, 2) + n
    monkeypatch.setattr(data, 'load_wav', lambda p: p)      # "waveform" = the path itself
    batch = [('/x/u%d.flac' % i, [3 + i] * (6 - i) + [1]) for i in range(6)]
    for tag, b, mode, jobs in (('tr', batch, 'train', 1), ('tr_half', batch[1:], 'train', 3),
                               ('dv', batch[1:], 'test', 1), ('bucket', [batch], 'train', 3)):
        names, feat, alen, txt = data.collect_audio_batch(b, fake_transform, mode, n_jobs=jobs)
        g = gold['collate.' + tag]
        assert list(names) == g['names']
        assert list(feat.shape) == g['feat_shape']
        assert alen.tolist() == g['alen'] and alen.dtype == torch.int64
        assert txt.tolist() == g['txt'] and txt.dtype == torch.int64
        assert float(feat.double().sum()) == g['feat_sum']


def test_cli_surface_and_cpu_refusal(tmp_path):
    main = _mod('main')
    paras = main.build_parser().parse_args(['--config', 'x.yaml', '--njobs', '2', '--test', '--no-msg'])
    for k in ('config', 'name', 'logdir', 'ckpdir', 'outdir', 'load', 'seed', 'cudnn_ctc', 'njobs', 'cpu',
              'no_pin', 'test', 'no_msg', 'lm', 'amp', 'reserve_gpu', 'jit'):
        assert hasattr(paras, k)
    assert (paras.logdir, paras.ckpdir, paras.outdir, paras.seed) == ('log/', 'ckpt/', 'result/', 0)
    # no CPU fallback: the solver must refuse --cpu instead of silently training on the host
    cfg = tmp_path / 'c.yaml'
    cfg.write_text('hparas: {valid_step: 1, max_step: 1, curriculum: 0}\n')
    with pytest.raises(RuntimeError, match='not supported'):
        main.main(['--config', str(cfg), '--cpu', '--no-msg'])


def test_default_hparas_match_reference():
    opt = _mod('src.option')
    assert opt.default_hparas == {'GRAD_CLIP': 5.0, 'PROGRESS_STEP': 100, 'DEV_STEP_RATIO': 1.2,
                                  'DEV_N_EXAMPLE': 4, 'TB_FLUSH_FREQ': 180}


def test_gru_params_stacked_for_the_decoder_loop_compute_a_gru_cell():
    """speller_ops.stack_gru_params: nn.GRU rows (r, z, n) -> the decoder loop's four-rows-per-unit layout
    (asrk_speller_t::cell = 1: r, z, n_x, n_h).  Host arithmetic only: the stacked pre-activations, pushed through
    the loop's cell formula, equal torch's GRUCell (src/asr.py:172 with module 'GRU' runs nn.GRU), and autograd routes
    the gradients of the stacked tensors back to the original blocks (zero blocks receive none)."""
    sops = importlib.import_module(PKG + ".speller_ops")
    torch.manual_seed(3)
    H, In, B = 6, 9, 4
    cell = torch.nn.GRUCell(In, H)
    x, h = torch.randn(B, In), torch.randn(B, H)
    ps = [p.detach().clone().requires_grad_(True) for p in (cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh)]
    w_ih4, w_hh4, b_ih4, b_hh4 = sops.stack_gru_params(*ps)
    assert w_ih4.shape == (4 * H, In) and w_hh4.shape == (4 * H, H) and b_ih4.shape == b_hh4.shape == (4 * H,)
    pre = x @ w_ih4.t() + b_ih4 + h @ w_hh4.t() + b_hh4                       # what the gate GEMM + eproj deliver
    r, z = torch.sigmoid(pre[:, :H]), torch.sigmoid(pre[:, H:2 * H])
    n = torch.tanh(pre[:, 2 * H:3 * H] + r * pre[:, 3 * H:])
    h_new = (1 - z) * n + z * h
    ref = cell(x, h)
    assert torch.allclose(h_new, ref, atol=1e-6)
    wsum = torch.randn(B, H)
    (h_new * wsum).sum().backward()
    refg = torch.autograd.grad((ref * wsum).sum(), list(cell.parameters()))
    for got, want in zip(ps, refg):
        assert torch.allclose(got.grad, want, atol=1e-5)
