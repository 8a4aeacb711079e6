"""End-to-end drop-in check on the GPU: `main.py --config ...` trains a small hybrid CTC-attention
model on a synthetic LibriSpeech-layout wav corpus through the reference's solver surface
(load_data -> set_model -> exec), checkpoints, resumes, then `--test` decodes greedily and with the
joint CTC-attention beam search and writes the reference's output files.  Everything on the path
(fbank/delta/CMVN, encoder, attention decoder, both losses, prefix scoring) is the HIP library."""
import importlib
import json
import math
import os
import wave

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
PKG = "end-to-end-asr-pytorch_amd"

WORDS = ['HELLO', 'WORLD', 'THE', 'CAT', 'SAT', 'ON', 'A', 'MAT', 'RED', 'DOOR']


def _write_wav(path, seconds, f0, seed):
    rng = np.random.default_rng(seed)
    n = int(16000 * seconds)
    t = np.arange(n) / 16000.0
    x = 0.3 * np.sin(2 * np.pi * f0 * t) + 0.1 * np.sin(2 * np.pi * 3.1 * f0 * t) + 0.02 * rng.standard_normal(n)
    with wave.open(path, 'wb') as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes((np.clip(x, -1, 1) * 32767).astype('<i2').tobytes())


def _make_corpus(root):
    rng = np.random.default_rng(0)
    for split, n_utt in (('train-x', 12), ('dev-x', 3), ('test-x', 2)):
        d = os.path.join(root, split, '7', '9')
        os.makedirs(d)
        with open(os.path.join(d, '7-9.trans.txt'), 'w') as f:
            for i in range(n_utt):
                nw = int(rng.integers(1, 4))
                words = [WORDS[int(k)] for k in rng.integers(0, len(WORDS), nw)]
                f.write('7-9-%04d %s\n' % (i, ' '.join(words)))
                _write_wav(os.path.join(d, '7-9-%04d.wav' % i), 0.5 + 0.25 * nw + 0.05 * i,
                           180.0 + 40 * i, seed=i)
    vocab = os.path.join(root, 'char.txt')
    with open(vocab, 'w') as f:
        f.write('\n'.join([' '] + sorted(set(''.join(WORDS)))) + '\n')
    return vocab


def _configs(root, vocab, tmp):
    train = {
        'data': {'corpus': {'name': 'Librispeech', 'path': root, 'train_split': ['train-x'],
                            'dev_split': ['dev-x'], 'bucketing': True, 'batch_size': 4},
                 'audio': {'feat_type': 'fbank', 'feat_dim': 40, 'frame_length': 25, 'frame_shift': 10,
                           'dither': 0, 'apply_cmvn': True, 'delta_order': 2, 'delta_window_size': 2},
                 'text': {'mode': 'character', 'vocab_file': vocab}},
        'hparas': {'valid_step': 4, 'max_step': 8, 'tf_start': 1.0, 'tf_end': 1.0, 'tf_step': 100,
                   'optimizer': 'Adadelta', 'lr': 1.0, 'eps': 1e-8, 'lr_scheduler': 'fixed', 'curriculum': 0},
        'model': {'ctc_weight': 0.5,
                  'encoder': {'prenet': '', 'module': 'LSTM', 'bidirection': True, 'dim': [32, 32],
                              'dropout': [0, 0], 'layer_norm': [False, False], 'proj': [True, True],
                              'sample_rate': [2, 2], 'sample_style': 'drop'},
                  'attention': {'mode': 'loc', 'dim': 24, 'num_head': 1, 'v_proj': False, 'temperature': 0.5,
                                'loc_kernel_size': 11, 'loc_kernel_num': 4},
                  'decoder': {'module': 'LSTM', 'dim': 32, 'layer': 1, 'dropout': 0}},
    }
    tr_path = os.path.join(tmp, 'asr_tiny.yaml')
    yaml.safe_dump(train, open(tr_path, 'w'))
    return train, tr_path


def _decode_cfg(tmp, tr_path, ckpt, name, **decode):
    cfg = {'src': {'ckpt': ckpt, 'config': tr_path},
           'data': {'corpus': {'name': 'Librispeech', 'dev_split': ['dev-x'], 'test_split': ['test-x']}},
           'decode': decode}
    p = os.path.join(tmp, name + '.yaml')
    yaml.safe_dump(cfg, open(p, 'w'))
    return p


def test_train_resume_and_decode_through_main(tmp_path):
    main = importlib.import_module(PKG + '.main')
    tmp = str(tmp_path)
    root = os.path.join(tmp, 'corpus')
    vocab = _make_corpus(root)
    train, tr_path = _configs(root, vocab, tmp)
    common = ['--logdir', os.path.join(tmp, 'log'), '--ckpdir', os.path.join(tmp, 'ckpt'),
              '--outdir', os.path.join(tmp, 'result'), '--njobs', '2', '--no-msg']

    # ---- train 8 steps (validation + checkpoint at steps 1, 4, 8)
    solver = main.main(['--config', tr_path] + common)
    assert solver.step >= 8
    ckdir = os.path.join(tmp, 'ckpt', 'asr_tiny_sd0')
    latest = os.path.join(ckdir, 'latest.pth')
    assert os.path.exists(latest)
    ck = torch.load(latest, map_location='cpu')
    assert set(ck.keys()) == {'model', 'optimizer', 'global_step', 'wer'}
    # like the reference, the loop leaves the epoch once step > max_step; the last validation /
    # checkpoint happened at step 8
    assert ck['global_step'] == 8 and solver.step == 9
    # reference state_dict naming survives (checkpoint portability)
    for k in ('encoder.layers.0.layer.weight_ih_l0', 'encoder.layers.0.layer.weight_hh_l0_reverse',
              'encoder.layers.1.pj.weight', 'ctc_layer.weight', 'pre_embed.weight',
              'decoder.layers.weight_ih_l0', 'decoder.char_trans.weight', 'attention.proj_q.weight',
              'attention.att_layer.loc_conv.weight'):
        assert k in ck['model'], k
    logs = [json.loads(l) for l in open(os.path.join(tmp, 'log', 'asr_tiny_sd0', 'log.jsonl'))] \
        if os.path.exists(os.path.join(tmp, 'log', 'asr_tiny_sd0', 'log.jsonl')) else []
    if logs:        # (TensorBoard present -> event files instead)
        losses = [r['scalars'] for r in logs if r['name'] == 'loss']
        assert losses and all(math.isfinite(v) for d in losses for v in d.values())

    # ---- resume from the checkpoint for 4 more steps
    train['hparas']['max_step'] = 12
    yaml.safe_dump(train, open(tr_path, 'w'))
    solver2 = main.main(['--config', tr_path, '--load', latest] + common)
    assert solver2.step >= 12

    # ---- greedy decode (batch-wise), then joint CTC-attention beam search (instance-wise)
    gcfg = _decode_cfg(tmp, tr_path, latest, 'dec_greedy', beam_size=1, min_len_ratio=0.01, max_len_ratio=0.3)
    main.main(['--config', gcfg, '--test'] + common)
    for s, n in (('dev', 3), ('test', 2)):
        lines = open(os.path.join(tmp, 'result', 'dec_greedy_%s_output.csv' % s)).read().splitlines()
        assert lines[0] == 'idx\thyp\ttruth' and len(lines) == n + 1
        assert all(len(l.split('\t')) == 3 for l in lines[1:])
    bcfg = _decode_cfg(tmp, tr_path, latest, 'dec_beam', beam_size=2, min_len_ratio=0.01, max_len_ratio=0.1,
                       lm_path='', lm_config='', lm_weight=0.0, ctc_weight=0.3)
    main.main(['--config', bcfg, '--test'] + common)
    out = open(os.path.join(tmp, 'result', 'dec_beam_test_output.csv')).read().splitlines()
    beams = open(os.path.join(tmp, 'result', 'dec_beam_test_beam-2-0.0.csv')).read().splitlines()
    assert len(out) == 3 and beams[0] == 'idx\tbeam\thyp\ttruth' and len(beams) >= 3
    # truth column round-trips the transcripts through the tokenizer
    truths = {l.split('\t')[-1] for l in out[1:]}
    assert all(set(t.split(' ')) <= set(WORDS) for t in truths)
    # ---- pure CTC beam search
    ccfg = _decode_cfg(tmp, tr_path, latest, 'dec_ctc', beam_size=2, vocab_candidate=4, min_len_ratio=0.01,
                       max_len_ratio=0.1, lm_path='', lm_config='', lm_weight=0.0, ctc_weight=1.0)
    main.main(['--config', ccfg, '--test'] + common)
    assert len(open(os.path.join(tmp, 'result', 'dec_ctc_dev_output.csv')).read().splitlines()) == 4


@pytest.mark.parametrize("prenet", ["vgg", "cnn"])
def test_shipped_architecture_in_miniature(tmp_path, prenet):
    """config/libri/asr_example.yaml's shape (conv prenet -> projected BLSTM stack with dropout and
    LayerNorm -> location-aware attention decoder, scheduled sampling) trains and decodes"""
    main = importlib.import_module(PKG + '.main')
    tmp = str(tmp_path)
    root = os.path.join(tmp, 'corpus')
    vocab = _make_corpus(root)
    train, tr_path = _configs(root, vocab, tmp)
    train['model']['ctc_weight'] = 0.3
    train['model']['encoder'].update(prenet=prenet, dim=[32, 32, 32], dropout=[0.1, 0.1, 0.0],
                                     layer_norm=[True, False, True], proj=[True, True, True],
                                     sample_rate=[1, 1, 1], sample_style='drop')
    train['model']['decoder'].update(layer=2, dropout=0.1)
    train['hparas'].update(max_step=6, valid_step=3, tf_start=1.0, tf_end=0.5, tf_step=4)
    yaml.safe_dump(train, open(tr_path, 'w'))
    common = ['--logdir', os.path.join(tmp, 'log'), '--ckpdir', os.path.join(tmp, 'ckpt'),
              '--outdir', os.path.join(tmp, 'result'), '--njobs', '2', '--no-msg']
    solver = main.main(['--config', tr_path] + common)
    assert solver.step >= 6
    latest = os.path.join(tmp, 'ckpt', 'asr_tiny_sd0', 'latest.pth')
    ck = torch.load(latest, map_location='cpu')
    key = 'encoder.layers.0.extractor.0.weight'
    assert key in ck['model'] and ck['model'][key].dim() == (4 if prenet == 'vgg' else 3)
    assert all(torch.isfinite(v).all() for v in ck['model'].values())
    gcfg = _decode_cfg(tmp, tr_path, latest, 'dec_greedy', beam_size=1, min_len_ratio=0.01, max_len_ratio=0.3)
    main.main(['--config', gcfg, '--test'] + common)
    assert len(open(os.path.join(tmp, 'result', 'dec_greedy_test_output.csv')).read().splitlines()) == 3


def test_training_reduces_loss_on_fixed_batch(tmp_path):
    """the whole solver step (forward, CTC + CE, backward, clip, Adadelta) actually learns"""
    main = importlib.import_module(PKG + '.main')
    tmp = str(tmp_path)
    root = os.path.join(tmp, 'corpus')
    vocab = _make_corpus(root)
    train, tr_path = _configs(root, vocab, tmp)
    train['hparas'].update(max_step=60, valid_step=1000)
    train['data']['corpus'].update(batch_size=12, bucketing=False)
    yaml.safe_dump(train, open(tr_path, 'w'))
    common = ['--logdir', os.path.join(tmp, 'log'), '--ckpdir', os.path.join(tmp, 'ckpt'), '--njobs', '1', '--no-msg']
    # capture the loss each step by wrapping Solver.backward
    mod = importlib.import_module(PKG + '.bin.train_asr')
    seen = []
    orig = mod.Solver.backward

    def spy(self, loss):
        seen.append(float(loss.detach()))
        return orig(self, loss)
    mod.Solver.backward = spy
    try:
        main.main(['--config', tr_path] + common)
    finally:
        mod.Solver.backward = orig
    assert len(seen) >= 60 and all(math.isfinite(v) for v in seen)
    assert np.mean(seen[-5:]) < 0.7 * np.mean(seen[:5]), (seen[:5], seen[-5:])


def test_gru_model_trains_and_beam_decodes(tmp_path):
    """module: 'GRU' in the encoder and the decoder (src/module.py:112-113, src/asr.py:175-176) through
    the same solver / beam-search surface"""
    main = importlib.import_module(PKG + '.main')
    tmp = str(tmp_path)
    root = os.path.join(tmp, 'corpus')
    vocab = _make_corpus(root)
    train, tr_path = _configs(root, vocab, tmp)
    train['model']['encoder'].update(module='GRU', dim=[24, 24], proj=[False, False], sample_rate=[2, 2],
                                     sample_style='concat')
    train['model']['decoder'].update(module='GRU', layer=2)
    train['hparas'].update(max_step=4, valid_step=2, optimizer='Adam', lr=0.001, lr_scheduler='warmup')
    yaml.safe_dump(train, open(tr_path, 'w'))
    common = ['--logdir', os.path.join(tmp, 'log'), '--ckpdir', os.path.join(tmp, 'ckpt'),
              '--outdir', os.path.join(tmp, 'result'), '--njobs', '1', '--no-msg']
    solver = main.main(['--config', tr_path] + common)
    assert solver.step >= 4 and solver.optimizer.fused           # fused Adam under the warm-up schedule
    latest = os.path.join(tmp, 'ckpt', 'asr_tiny_sd0', 'latest.pth')
    ck = torch.load(latest, map_location='cpu')
    assert ck['model']['encoder.layers.0.layer.weight_ih_l0'].shape[0] == 3 * 24
    assert all(torch.isfinite(v).all() for v in ck['model'].values())
    bcfg = _decode_cfg(tmp, tr_path, latest, 'dec_beam', beam_size=2, min_len_ratio=0.01, max_len_ratio=0.1,
                       lm_path='', lm_config='', lm_weight=0.0, ctc_weight=0.3)
    main.main(['--config', bcfg, '--test'] + common)
    assert len(open(os.path.join(tmp, 'result', 'dec_beam_test_output.csv')).read().splitlines()) == 3


def test_rnnlm_training_through_main_and_lm_fusion_decode(tmp_path):
    """`main.py --lm` (bin/train_lm.py) on the corpus transcripts, then the trained LM is fused into
    the joint beam search of an ASR checkpoint (decode_example.yaml's lm_path / lm_config / lm_weight)"""
    main = importlib.import_module(PKG + '.main')
    tmp = str(tmp_path)
    root = os.path.join(tmp, 'corpus')
    vocab = _make_corpus(root)
    train, tr_path = _configs(root, vocab, tmp)
    train['hparas'].update(max_step=3, valid_step=3)
    yaml.safe_dump(train, open(tr_path, 'w'))
    common = ['--logdir', os.path.join(tmp, 'log'), '--ckpdir', os.path.join(tmp, 'ckpt'),
              '--outdir', os.path.join(tmp, 'result'), '--njobs', '1', '--no-msg']
    main.main(['--config', tr_path] + common)
    asr_ckpt = os.path.join(tmp, 'ckpt', 'asr_tiny_sd0', 'latest.pth')
    lm_cfg = {'data': {'corpus': {'name': 'Librispeech', 'path': root, 'train_split': ['train-x'],
                                  'dev_split': ['dev-x'], 'bucketing': True, 'batch_size': 4},
                       'text': {'mode': 'character', 'vocab_file': vocab}},
              'hparas': {'valid_step': 5, 'max_step': 10, 'optimizer': 'Adam', 'lr': 0.01, 'eps': 1e-8,
                         'lr_scheduler': 'fixed'},
              'model': {'emb_tying': False, 'emb_dim': 16, 'module': 'LSTM', 'dim': 24, 'n_layers': 2,
                        'dropout': 0.1}}
    lm_path = os.path.join(tmp, 'lm_tiny.yaml')
    yaml.safe_dump(lm_cfg, open(lm_path, 'w'))
    solver = main.main(['--config', lm_path, '--lm'] + common)
    assert solver.step >= 10
    lm_ckpt = os.path.join(tmp, 'ckpt', 'lm_tiny_sd0', 'best_ppx.pth')
    ck = torch.load(lm_ckpt, map_location='cpu')
    assert {'model', 'optimizer', 'global_step', 'perplexity'} <= set(ck.keys())
    assert set(ck['model'].keys()) >= {'emb.weight', 'rnn.weight_ih_l0', 'rnn.weight_hh_l1', 'trans.weight'}
    assert math.isfinite(ck['perplexity']) and ck['perplexity'] > 1.0
    bcfg = _decode_cfg(tmp, tr_path, asr_ckpt, 'dec_lm', beam_size=2, min_len_ratio=0.01, max_len_ratio=0.1,
                       lm_path=lm_ckpt, lm_config=lm_path, lm_weight=0.3, ctc_weight=0.3)
    main.main(['--config', bcfg, '--test'] + common)
    assert len(open(os.path.join(tmp, 'result', 'dec_lm_test_output.csv')).read().splitlines()) == 3
