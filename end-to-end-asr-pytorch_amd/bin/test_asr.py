"""Testing solver (greedy / CTC beam / joint CTC-attention(+LM) beam) — mirror of the reference's
bin/test_asr.py:12-223: same config handling (the training config named by `src.config` fixes
audio / text / model), same output files `<outdir>/<name>_{dev,test}_output.csv` and
`..._beam-<k>-<lm_w>.csv`, same `(name, hyps, truth)` result tuples.

The reference fans utterances out over CPU processes (joblib, bin/test_asr.py:163-167) and advances
one hypothesis at a time; here every live hypothesis of an utterance is a row of the same device batch
(src/decode.py) and the fan-out is over GPUS: under `torch.distributed.run` (one process per GPU) rank
r decodes utterances r, r + world, ... and rank 0 gathers the rows and writes the files
(parallel.shard_indices / gather_in_order; no data-path collective - decoding is embarrassingly
parallel over utterances).
"""
import os

import torch
from torch.utils.data import DataLoader

from .. import ops
from ..src.solver import BaseSolver
from ..src.asr import ASR
from ..src.decode import BeamDecoder
from ..src.ctc import CTCBeamDecoder
from ..src.data import load_dataset
from ..parallel import shard_indices, gather_in_order


class Solver(BaseSolver):
    ''' Solver for testing'''

    def __init__(self, config, paras, mode):
        super().__init__(config, paras, mode)
        assert self.config['data']['corpus']['name'] == self.src_config['data']['corpus']['name']
        self.config['data']['corpus']['path'] = self.src_config['data']['corpus']['path']
        self.config['data']['corpus']['bucketing'] = False
        # The following attributes are identical to the training config
        self.config['data']['audio'] = self.src_config['data']['audio']
        self.config['data']['text'] = self.src_config['data']['text']
        self.config['hparas'] = self.src_config['hparas']
        self.config['model'] = self.src_config['model']

        self.output_file = str(self.ckpdir) + '_{}_{}.csv'
        # Beam decoding is instance-wise
        self.greedy = self.config['decode']['beam_size'] == 1
        if not self.greedy:
            self.config['data']['corpus']['batch_size'] = 1
        else:
            self.config['data']['corpus'].setdefault('batch_size', self.src_config['data']['corpus']['batch_size'])
        self.step = 0

    def fetch_data(self, data):
        _, feat, feat_len, txt = data
        feat = feat.to(self.device)
        feat_len = feat_len.to(self.device)
        txt = txt.to(self.device)
        txt_len = torch.sum(txt != 0, dim=-1)
        return feat, feat_len, txt, txt_len

    def load_data(self):
        self.dv_set, self.tt_set, self.feat_dim, self.vocab_size, self.tokenizer, msg = \
            load_dataset(self.paras.njobs, self.paras.gpu, self.paras.pin_memory, False, **self.config['data'])
        self.verbose(msg)

    def set_model(self):
        init_adadelta = self.config['hparas']['optimizer'] == 'Adadelta'
        self.model = ASR(self.feat_dim, self.vocab_size, init_adadelta, **self.config['model']).to(self.device)
        self.load_ckpt()        # eval mode

        dcfg = self.config['decode']
        self.ctc_only = False
        if self.greedy:
            # attention-based if the ASR has a decoder, else CTC
            self.decoder = self.model
            self.verbose(['Decode spec| Greedy decoding \t| Max len ratio = {}'.format(dcfg['max_len_ratio'])])
        else:
            if (not self.model.enable_att) or dcfg.get('ctc_weight', 0.0) == 1.0:
                assert dcfg['beam_size'] <= dcfg['vocab_candidate']
                self.decoder = CTCBeamDecoder(self.model, [1] + [r for r in range(3, self.vocab_size)],
                                              dcfg['beam_size'], dcfg['vocab_candidate'],
                                              lm_path=dcfg.get('lm_path', ''), lm_config=dcfg.get('lm_config', ''),
                                              lm_weight=dcfg.get('lm_weight', 0.0), device=self.device)
                self.ctc_only = True
            else:
                self.decoder = BeamDecoder(self.model, self.emb_decoder, **dcfg)
            self.verbose(self.decoder.create_msg())
        self.enable_att = self.model.enable_att

    def _my_share(self, ds):
        ''' (loader over this rank's batches, their global batch indices, total number of batches) '''
        n = len(ds)
        if self.world == 1:
            return ds, list(range(n)), n
        if ds.batch_size == 1:      # instance-wise decoding: shard the utterances themselves
            mine = shard_indices(n, self.rank, self.world)
            return DataLoader(ds.dataset, batch_size=1, sampler=mine, collate_fn=ds.collate_fn,
                              num_workers=0), mine, n
        # batch-wise (greedy): shard whole batches so that batch boundaries stay the reference's
        mine = shard_indices(n, self.rank, self.world)
        bs = ds.batch_size
        idx = [j for b in mine for j in range(b * bs, min((b + 1) * bs, len(ds.dataset)))]
        return DataLoader(ds.dataset, batch_size=bs, sampler=idx, collate_fn=ds.collate_fn,
                          num_workers=0), mine, n

    def greedy_decode(self, dv_set):
        ''' batch-wise greedy decoding (reference: bin/test_asr.py:103-123) '''
        results = []
        dv_set, batch_ids, _ = self._my_share(dv_set)
        for k, data in enumerate(dv_set):
            i = batch_ids[k]
            self.progress('Valid step - {}/{}'.format(i + 1, len(dv_set)))
            feat, feat_len, txt, txt_len = self.fetch_data(data)
            with torch.no_grad():
                ctc_output, encode_len, att_output, att_align, dec_state = \
                    self.decoder(feat, feat_len, int(float(feat_len.max()) * self.config['decode']['max_len_ratio']))
            out = att_output if att_output is not None else ctc_output
            hyps = ops.argmax(out).tolist()
            for j in range(len(txt)):
                idx = j + self.config['data']['corpus']['batch_size'] * i
                results.append((str(idx), [hyps[j]], txt[j].tolist()))
        return results

    def exec(self):
        ''' Testing End-to-end ASR system '''
        dcfg = self.config['decode']
        for s, ds in zip(['dev', 'test'], [self.dv_set, self.tt_set]):
            self.cur_output_path = self.output_file.format(s, 'output')
            if self.rank == 0:
                with open(self.cur_output_path, 'w', encoding='UTF-8') as f:
                    f.write('idx\thyp\ttruth\n')
            if self.greedy:
                self.verbose('Performing batch-wise greedy decoding on {} set, num of batch = {}.'.format(s, len(ds)))
                local = self.greedy_decode(ds)
                # per-batch result groups travel to rank 0 and are flattened in batch order
                _, batch_ids, n_b = self._my_share(ds)
                bs = ds.batch_size
                groups, pos = [], 0
                for b in batch_ids:
                    cnt = min(bs, len(ds.dataset) - b * bs)
                    groups.append(local[pos:pos + cnt])
                    pos += cnt
                merged = gather_in_order(groups, n_b, self.dist, self.rank, self.world)
                if merged is None:
                    continue
                results = [r for g in merged for r in g]
                self.verbose('Results will be stored at {}'.format(self.cur_output_path))
                self.write_hyp(results, self.cur_output_path, '-')
            else:
                self.cur_beam_path = self.output_file.format(
                    s, 'beam-{}-{}'.format(dcfg['beam_size'], dcfg.get('lm_weight', 0.0)))
                if self.rank == 0:
                    with open(self.cur_beam_path, 'w') as f:
                        f.write('idx\tbeam\thyp\ttruth\n')
                func = ctc_beam_decode if self.ctc_only else beam_decode
                self.verbose('Performing instance-wise {}beam decoding on {} set, num of batch = {}.'.format(
                    'CTC ' if self.ctc_only else '', s, len(ds)))
                mine, ids, n_utt = self._my_share(ds)
                local = []
                # joint CTC-attention(+LM) search: DECODE_BATCH utterances per device step (BeamDecoder.forward_batch;
                # same hypotheses as one at a time - the reference's parallel axis is the utterance too, over CPU
                # processes: bin/test_asr.py:163-167); ASRK_DECODE_BATCH=1 restores utterance-by-utterance decoding
                group = max(1, int(os.environ.get('ASRK_DECODE_BATCH', '16')))
                if self.ctc_only and self.decoder.apply_lm:
                    group = 1                    # CTC search with LM fusion: one launch + one LM step per frame
                many = ctc_beam_decode_many if self.ctc_only else beam_decode_many
                pending = []
                for k, data in enumerate(mine):
                    self.progress('Decode - {}/{}'.format(ids[k] + 1, n_utt))
                    if group == 1:
                        local.append(func(data, self.decoder, self.device))
                        continue
                    pending.append(data)
                    if len(pending) == group:
                        local += many(pending, self.decoder, self.device)
                        pending = []
                if pending:
                    local += many(pending, self.decoder, self.device)
                results = gather_in_order(local, n_utt, self.dist, self.rank, self.world)
                if results is None:          # not rank 0: its rows have been handed over
                    continue
                self.verbose('Results/Beams will be stored at {} / {}.'.format(self.cur_output_path, self.cur_beam_path))
                self.write_hyp(results, self.cur_output_path, self.cur_beam_path)
        self.verbose('All done !')

    def write_hyp(self, results, best_path, beam_path):
        ''' Record decoding results (reference: bin/test_asr.py:173-197) '''
        # NOTE bug-compatible: the reference computes `ignore_repeat = not enable_att` for greedy
        # decoding (bin/test_asr.py:175-179) but never passes it on (line 185), so greedy CTC
        # hypotheses are written WITHOUT repeat merging.  The output files match the reference's;
        # set ASRK_GREEDY_CTC_MERGE=1 to write collapsed CTC hypotheses instead.
        ignore_repeat = self.greedy and (not self.enable_att) and os.environ.get('ASRK_GREEDY_CTC_MERGE') == '1'
        for name, hyp_seqs, truth in results:
            if self.ctc_only and not self.greedy:
                hyp_seqs = [self.tokenizer.decode(hyp, ignore_repeat=False) for hyp in hyp_seqs[:-1]] + \
                           [self.tokenizer.decode(hyp_seqs[-1], ignore_repeat=True)]
            else:
                hyp_seqs = [self.tokenizer.decode(hyp, ignore_repeat=ignore_repeat) for hyp in hyp_seqs]
            truth = self.tokenizer.decode(truth)
            with open(best_path, 'a') as f:
                if len(hyp_seqs[0]) == 0:
                    hyp_seqs[0] = ' '        # keep the column non-empty
                f.write('\t'.join([name, hyp_seqs[0], truth]) + '\n')
            if not self.greedy:
                with open(beam_path, 'a', encoding='UTF-8') as f:
                    for b, hyp in enumerate(hyp_seqs):
                        f.write('\t'.join([name, str(b), hyp, truth]) + '\n')


def beam_decode(data, model, device):
    ''' one utterance (batch size 1) -> (name, [beam hypotheses as id lists], truth ids) '''
    name, feat, feat_len, txt = data
    with torch.no_grad():
        hyps = model(feat.to(device), feat_len.to(device))
    return (name[0], [hyp.outIndex for hyp in hyps], txt[0].cpu().tolist())


def _pad_items(items, device):
    lens = [int(d[2][0]) for d in items]
    feat = torch.zeros((len(items), max(lens), items[0][1].shape[-1]), dtype=torch.float32, device=device)
    for u, d in enumerate(items):
        feat[u, :lens[u]] = d[1][0, :lens[u]].to(device)
    return feat, torch.tensor(lens, device=device)


def beam_decode_many(items, model, device):
    ''' several batch-1 loader items -> their (name, hypotheses, truth) rows, decoded together '''
    feat, lens = _pad_items(items, device)
    with torch.no_grad():
        hyps = model.forward_batch(feat, lens)
    return [(d[0][0], [h.outIndex for h in hyps[u]], d[3][0].cpu().tolist()) for u, d in enumerate(items)]


def ctc_beam_decode_many(items, model, device):
    feat, lens = _pad_items(items, device)
    with torch.no_grad():
        hyps = model.forward_batch(feat, lens)
    return [(d[0][0], hyps[u], d[3][0].cpu().tolist()) for u, d in enumerate(items)]


def ctc_beam_decode(data, model, device):
    name, feat, feat_len, txt = data
    with torch.no_grad():
        hyp = model(feat.to(device), feat_len.to(device))
    return (name[0], hyp, txt[0].cpu().tolist())
