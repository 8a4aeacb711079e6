"""Batch assembly for ASR training / testing — mirror of the reference's src/data.py
(`load_dataset`, `create_dataset`, `collect_audio_batch`; same arguments, same return tuples, same
batch-halving and length-sorting rules).

MI355X-first difference: the feature pipeline (fbank -> delta -> CMVN) is a chain of gfx950
kernels, so collation runs in the training process on the device — wav files are read by a small
host thread pool, features are computed and padded in HBM, and the batch that reaches
`Solver.fetch_data` is already resident (its `.to(device)` is a no-op).  The DataLoader therefore
runs with num_workers=0: forked workers cannot share the HIP context, and there is no host-side
feature tensor left to pin.
"""
from concurrent.futures import ThreadPoolExecutor
from functools import partial

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, Sampler
from torch.nn.utils.rnn import pad_sequence

from .text import load_text_encoder
import os

from .audio import create_transform, load_wav, load_pcm, audio_num_samples

# Batch size will be halved if the longest wavefile surpasses threshold (reference: src/data.py:8-11)
HALF_BATCHSIZE_AUDIO_LEN = 800
HALF_BATCHSIZE_TEXT_LEN = 150

_POOL = None


def _pool(n_jobs):
    global _POOL
    if _POOL is None and n_jobs > 1:
        _POOL = ThreadPoolExecutor(max_workers=n_jobs)
    return _POOL


class _BatchPathUnsupported(Exception):
    ''' an input the whole-batch front end does not take (not 16-bit PCM, mixed sample rates in one batch):
        the collate function then uses the per-file chain, which accepts everything the reference does '''


def _half(n, shard):
    ''' size of a halved batch (src/data.py:22-24); a data-parallel GLOBAL batch keeps at least one utterance per rank '''
    return max(n // 2, shard[1]) if shard is not None and shard[1] > 1 else n // 2


def deal_global_batch(lengths, rank, world):
    ''' SURVEY §8e sharding rule: positions of a GLOBAL batch (already halved, src/data.py:22-24) that rank `rank`
        of `world` trains on - sort by descending length (stable, as src/data.py:36-37), deal round-robin.  Every
        rank's shard is itself in descending order, the per-rank longest utterances (hence the step times) differ by
        at most one position of the global order, and shard sizes differ by at most one utterance. '''
    order = sorted(range(len(lengths)), key=lambda i: lengths[i], reverse=True)
    return order[rank::world]


class SharedShuffleSampler(Sampler):
    ''' Data-parallel index stream: EVERY rank draws the same global batches (shuffle seeded by seed + epoch), and the
        collate function keeps this rank's share of each (deal_global_batch).  `n_draws` indices per epoch. '''

    def __init__(self, n_items, n_draws, shuffle, seed=0):
        self.n_items, self.n_draws, self.shuffle, self.seed, self.epoch = n_items, n_draws, shuffle, seed, 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n_items, generator=g).tolist()
        else:
            idx = list(range(self.n_items))
        return iter(idx[:self.n_draws])

    def __len__(self):
        return self.n_draws


def collect_audio_batch(batch, audio_transform, mode, n_jobs=1, shard=None):
    ''' [(audio_path, [token ids]), ...] (or one bucket of them) ->
        (names, feat [B,T,D] on the transform's device, feat_len [B], text [B,L])
        (reference: src/data.py:14-46).  shard = (rank, world): `batch` is the GLOBAL batch of a data-parallel step;
        the halving rule is applied to it as a whole, then this rank keeps its length-balanced share (§8e cond. 5). '''
    if type(batch[0]) is not tuple:
        batch = batch[0]
    paths = [str(b[0]) for b in batch]
    pool = _pool(n_jobs)
    bt = getattr(audio_transform, 'batch', None)
    if bt is not None and os.environ.get('ASRK_BATCH_FBANK', '1') != '0':
        try:
            return _collect_audio_batch_device(batch, paths, bt, mode, pool, shard)
        except _BatchPathUnsupported:
            pass
    with torch.no_grad():
        # the first utterance of a bucket is the longest transcript: its frame count decides halving
        first = audio_transform(paths[0])
        if first.shape[0] > HALF_BATCHSIZE_AUDIO_LEN and mode == 'train':
            batch, paths = batch[:_half(len(batch), shard)], paths[:_half(len(batch), shard)]
        if shard is not None and shard[1] > 1:
            fc = None
            if bt is not None:                    # frame counts from the file headers: only this rank's share is extracted
                try:
                    fc = [first.shape[0]] + [bt.frame_count(*_num_samples_or_defer(p)) for p in paths[1:]]
                except _BatchPathUnsupported:     # a header the reader rejects: extract everything to learn the lengths
                    fc = None
            if fc is not None:
                mine = deal_global_batch(fc, *shard)
                feats = [first if i == 0 else audio_transform(load_wav(paths[i])) for i in mine]
            else:
                allf = [first] + [audio_transform(load_wav(p)) for p in paths[1:]]
                mine = deal_global_batch([f.shape[0] for f in allf], *shard)
                feats = [allf[i] for i in mine]
            batch, paths = [batch[i] for i in mine], [paths[i] for i in mine]
        else:
            waves = list(pool.map(load_wav, paths[1:])) if pool is not None else [load_wav(p) for p in paths[1:]]
            feats = [first] + [audio_transform(w) for w in waves]
    names = [p.split('/')[-1].split('.')[0] for p in paths]
    text = [torch.LongTensor(b[1]) for b in batch]
    # descending audio length within the batch; sorted() is stable, so ties keep their order
    order = sorted(range(len(feats)), key=lambda i: feats[i].shape[0], reverse=True)
    names = tuple(names[i] for i in order)
    audio_len = torch.LongTensor([feats[i].shape[0] for i in order])
    audio_feat = pad_sequence([feats[i] for i in order], batch_first=True)
    text = pad_sequence([text[i] for i in order], batch_first=True)
    return names, audio_feat, audio_len, text


def _num_samples_or_defer(path):
    ''' header-only length of a file for the whole-batch front end; a header the reader rejects (float / EXTENSIBLE
        .wav: wave.Error; a FLAC stream libasrk cannot parse: AsrkError) sends the batch to the per-file chain '''
    import wave
    from .._lib import AsrkError
    try:
        return audio_num_samples(path)
    except (wave.Error, AsrkError, EOFError, ValueError) as e:
        raise _BatchPathUnsupported(str(e))


def _load_pcm_or_defer(path):
    try:
        return load_pcm(path)
    except ValueError as e:
        raise _BatchPathUnsupported(str(e))


def _collect_audio_batch_device(batch, paths, bt, mode, pool, shard=None):
    ''' the same contract through the whole-batch front end (src/audio.py:BatchFeatureTransform): host threads
        read raw 16-bit PCM, ONE padded int16 upload, 7 launches for the batch.  Frame counts follow from the
        sample counts (snip-edges framing), so the halving rule (src/data.py:22-24) and the descending-length
        order (src/data.py:36-37) are decided before anything is extracted - no file is processed twice. '''
    n0, sr = _num_samples_or_defer(paths[0])
    if bt.frame_count(n0, sr) > HALF_BATCHSIZE_AUDIO_LEN and mode == 'train':
        batch, paths = batch[:_half(len(batch), shard)], paths[:_half(len(batch), shard)]
    if shard is not None and shard[1] > 1:
        # data parallel: the halved GLOBAL batch is dealt by length (read from the file headers); only this
        # rank's share is decoded, uploaded and extracted
        fc = [bt.frame_count(*_num_samples_or_defer(p)) for p in paths]
        mine = deal_global_batch(fc, *shard)
        batch, paths = [batch[i] for i in mine], [paths[i] for i in mine]
    loaded = list(pool.map(_load_pcm_or_defer, paths)) if pool is not None else [_load_pcm_or_defer(p) for p in paths]
    if any(r[1] != sr for r in loaded):
        raise _BatchPathUnsupported('mixed sample rates in one batch')
    pcm = [r[0] for r in loaded]
    frames = [bt.frame_count(len(x), sr) for x in pcm]
    # descending audio length within the batch; sorted() is stable, so ties keep their order
    order = sorted(range(len(pcm)), key=lambda i: frames[i], reverse=True)
    with torch.no_grad():
        audio_feat, audio_len = bt([pcm[i] for i in order], sr)
    names = tuple(paths[i].split('/')[-1].split('.')[0] for i in order)
    text = pad_sequence([torch.LongTensor(batch[i][1]) for i in order], batch_first=True)
    return names, audio_feat, audio_len, text


def collect_text_batch(batch, mode, shard=None):
    ''' [txt1 <list>, txt2 <list>, ...] (or one bucket of them) -> LongTensor [B, L] zero-padded
        (reference: src/data.py:46-61); shard = (rank, world) as in collect_audio_batch '''
    if type(batch[0][0]) is list:
        batch = batch[0]
    if len(batch[0]) > HALF_BATCHSIZE_TEXT_LEN and mode == 'train':
        batch = batch[:_half(len(batch), shard)]
    if shard is not None and shard[1] > 1:
        batch = [batch[i] for i in deal_global_batch([len(b) for b in batch], *shard)]
    return pad_sequence([torch.LongTensor(b) for b in batch], batch_first=True)


def _dp_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def create_dataset(tokenizer, ascending, name, path, bucketing, batch_size,
                   train_split=None, dev_split=None, test_split=None, world=1):
    ''' (reference: src/data.py:63-101).  world > 1 (data-parallel training): `batch_size` stays the PER-RANK batch
        (weak scaling); buckets / loader batches are GLOBAL batches of batch_size * world utterances that the collate
        function halves as a whole and deals over the ranks by length. '''
    if name.lower() == "librispeech":
        from ..corpus.librispeech import LibriDataset as Dataset
    else:
        raise NotImplementedError
    if train_split is not None:
        mode = 'train'
        use_bucket = bucketing and (not ascending)
        tr_loader_bs = 1 if use_bucket else batch_size * world
        bucket_size = batch_size * world if use_bucket else 1
        dv_set = Dataset(path, dev_split, tokenizer, 1)
        tr_set = Dataset(path, train_split, tokenizer, bucket_size, ascending=ascending)
        msg_list = _data_msg(name, path, str(train_split), len(tr_set), str(dev_split), len(dv_set),
                             batch_size, bucketing)
        return tr_set, dv_set, tr_loader_bs, batch_size, mode, msg_list
    mode = 'test'
    dv_set = Dataset(path, dev_split, tokenizer, 1)
    tt_set = Dataset(path, test_split, tokenizer, 1)
    msg_list = _data_msg(name, path, str(dev_split), len(dv_set), str(test_split), len(tt_set),
                         batch_size, False)
    msg_list = [m.replace('Dev', 'Test').replace('Train', 'Dev') for m in msg_list]
    return dv_set, tt_set, batch_size, batch_size, mode, msg_list


def create_textset(tokenizer, train_split, dev_split, name, path, bucketing, batch_size, world=1):
    ''' (reference: src/data.py:104-125); world as in create_dataset '''
    if name.lower() == "librispeech":
        from ..corpus.librispeech import LibriTextDataset as Dataset
    else:
        raise NotImplementedError
    bucket_size = batch_size * world if bucketing else 1
    tr_loader_bs = 1 if bucketing else batch_size * world
    dv_set = Dataset(path, dev_split, tokenizer, 1)          # no bucketing for the dev set
    tr_set = Dataset(path, train_split, tokenizer, bucket_size)
    msg_list = _data_msg(name, path, str(train_split), len(tr_set), str(dev_split), len(dv_set),
                         batch_size, bucketing)
    return tr_set, dv_set, tr_loader_bs, batch_size, msg_list


def load_textset(n_jobs, use_gpu, pin_memory, corpus, text):
    ''' text-only loaders for RNN-LM training (reference: src/data.py:160-181) '''
    tokenizer = load_text_encoder(**text)
    rank, world = _dp_world()
    tr_set, dv_set, tr_loader_bs, dv_loader_bs, data_msg = create_textset(tokenizer, world=world, **corpus)
    sampler, shard = None, None
    if world > 1:
        # one process per GPU: every rank draws the same GLOBAL batches and keeps its length-balanced share
        sampler, shard = _dp_sampler(len(tr_set), tr_loader_bs, world, True), (rank, world)
    tr_set = DataLoader(tr_set, batch_size=tr_loader_bs, shuffle=sampler is None, sampler=sampler,
                        drop_last=True, collate_fn=partial(collect_text_batch, mode='train', shard=shard),
                        num_workers=0)
    dv_set = DataLoader(dv_set, batch_size=dv_loader_bs, shuffle=False, drop_last=False,
                        collate_fn=partial(collect_text_batch, mode='dev'), num_workers=0)
    data_msg.append('I/O spec.  | Token type = {}\t| Vocab size = {}'.format(tokenizer.token_type,
                                                                              tokenizer.vocab_size))
    return tr_set, dv_set, tokenizer.vocab_size, tokenizer, data_msg


def load_dataset(n_jobs, use_gpu, pin_memory, ascending, corpus, audio, text):
    ''' (reference: src/data.py:128-157) -> (tr_loader, dv_loader, feat_dim, vocab_size, tokenizer, msg).
        `use_gpu` must be true (there is no CPU feature path); `pin_memory` is accepted and unused. '''
    if not use_gpu:
        raise RuntimeError("the feature pipeline runs in gfx950 kernels; --cpu is not supported")
    audio_transform, feat_dim = create_transform(audio.copy(), device='cuda')
    tokenizer = load_text_encoder(**text)
    rank, world = _dp_world()
    if 'train_split' not in corpus or corpus['train_split'] is None:
        world = 1                                  # decoding shards utterances itself (bin/test_asr.py)
    tr_set, dv_set, tr_loader_bs, dv_loader_bs, mode, data_msg = create_dataset(tokenizer, ascending, world=world,
                                                                                **corpus)
    shuffle = (mode == 'train' and not ascending)
    sampler, shard = None, None
    if mode == 'train' and world > 1:
        # one process per GPU: every rank draws the same GLOBAL batches (batch_size stays PER RANK: weak scaling of
        # the global batch) and keeps its length-balanced share of each (deal_global_batch)
        sampler, shard = _dp_sampler(len(tr_set), tr_loader_bs, world, shuffle), (rank, world)
    collect_tr = partial(collect_audio_batch, audio_transform=audio_transform, mode=mode, n_jobs=n_jobs, shard=shard)
    collect_dv = partial(collect_audio_batch, audio_transform=audio_transform, mode='test', n_jobs=n_jobs)
    tr_set = DataLoader(tr_set, batch_size=tr_loader_bs, shuffle=shuffle and sampler is None,
                        sampler=sampler, drop_last=shuffle,
                        collate_fn=collect_tr, num_workers=0)
    dv_set = DataLoader(dv_set, batch_size=dv_loader_bs, shuffle=False, drop_last=False,
                        collate_fn=collect_dv, num_workers=0)
    data_msg.append('I/O spec.  | Audio feature = {}\t| feature dim = {}\t| Token type = {}\t| Vocab size = {}'
                    .format(audio['feat_type'], feat_dim, tokenizer.token_type, tokenizer.vocab_size))
    return tr_set, dv_set, feat_dim, tokenizer.vocab_size, tokenizer, data_msg


def _dp_sampler(n_items, loader_bs, world, shuffle):
    ''' index stream of a data-parallel epoch.  Bucketed sets (loader batch 1: every index is a window of
        batch_size * world neighbours, corpus/librispeech.py:52-58) draw n / world windows per epoch - the same
        number of utterance visits per epoch as the single-process loader, which draws n windows of batch_size;
        plain sets are walked once (the loader cuts global batches of batch_size * world). '''
    if loader_bs == 1:
        n_draws = max(1, n_items // world)
    else:
        tail = n_items % loader_bs              # a last global batch with fewer utterances than ranks is dropped
        n_draws = n_items - tail if 0 < tail < world else n_items
    # the shuffle follows --seed like the single-process loader's: main.py seeds torch's global generator with it on
    # every rank before the solver is built, so initial_seed() is the same number everywhere
    return SharedShuffleSampler(n_items, n_draws, shuffle, seed=int(torch.initial_seed()) & 0x7fffffff)


def _data_msg(name, path, train_split, tr_set, dev_split, dv_set, batch_size, bucketing):
    return ['Data spec. | Corpus = {} (from {})'.format(name, path),
            '           | Train sets = {}\t| Number of utts = {}'.format(train_split, tr_set),
            '           | Dev sets = {}\t| Number of utts = {}'.format(dev_split, dv_set),
            '           | Batch size = {}\t\t| Bucketing = {}'.format(batch_size, bucketing)]
