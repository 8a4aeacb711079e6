"""Training solver for hybrid CTC-attention ASR — mirror of the reference's bin/train_asr.py:9-247
(`Solver.fetch_data / load_data / set_model / exec / validate`, same step order: pre_step ->
forward -> CTC + CE losses -> backward -> clip -> step -> log -> validate -> checkpoint).

The model, both losses and the feature pipeline are the gfx950 kernels of this package; under
torch.distributed.run each rank takes its own shard of every epoch and gradients are averaged by
parallel.DataParallelEngine while BPTT is still running.
"""
import torch

from .. import ops
from ..src.solver import BaseSolver
from ..src.asr import ASR
from ..src.optim import Optimizer
from ..src.data import load_dataset
from ..src.util import human_format, cal_er


class Solver(BaseSolver):
    ''' Solver for training'''

    def __init__(self, config, paras, mode):
        super().__init__(config, paras, mode)
        self.best_wer = {'att': 3.0, 'ctc': 3.0}
        # Curriculum learning affects data loader
        self.curriculum = self.config['hparas']['curriculum']
        if 'emb' in self.config and self.config['emb']['enable']:
            raise NotImplementedError('embedding regulariser/fusion plugin (src/plugin.py) is out of scope')

    def fetch_data(self, data):
        ''' batch is already resident in HBM (src/data.py); compute text seq. length '''
        _, feat, feat_len, txt = data
        # transcripts arrive on the host: their lengths (and the number of decoder steps, the longest of them) are
        # computed THERE - reading `int(txt_len.max())` back from the device would stall the host on the previous
        # step's kernels once per step
        txt_len = torch.sum(txt != 0, dim=-1)
        self.decode_step = int(txt_len.max())
        feat = feat.to(self.device)
        feat_len = feat_len.to(self.device)
        txt = txt.to(self.device)
        txt_len = txt_len.to(self.device)
        return feat, feat_len, txt, txt_len

    def load_data(self):
        self.tr_set, self.dv_set, self.feat_dim, self.vocab_size, self.tokenizer, msg = \
            load_dataset(self.paras.njobs, self.paras.gpu, self.paras.pin_memory,
                         self.curriculum > 0, **self.config['data'])
        self.verbose(msg)

    def set_model(self):
        ''' Setup ASR model and optimizer '''
        init_adadelta = self.config['hparas']['optimizer'] == 'Adadelta'
        self.model = ASR(self.feat_dim, self.vocab_size, init_adadelta, **self.config['model']).to(self.device)
        self.verbose(self.model.create_msg())
        model_paras = [{'params': self.model.parameters()}]
        # Losses (gfx950 kernels behind the torch.nn loss-module surface)
        self.seq_loss = ops.CrossEntropyLoss(ignore_index=0)
        self.ctc_loss = ops.CTCLoss(blank=0, zero_infinity=False)
        self.emb_fuse, self.emb_reg = False, False
        self.optimizer = Optimizer(model_paras, **self.config['hparas'])
        self.verbose(self.optimizer.create_msg())
        self.load_ckpt()
        self.enable_data_parallel()

    def compute_losses(self, ctc_output, encode_len, att_output, txt, txt_len):
        ''' (reference: bin/train_asr.py:115-133) -> (loss to back-propagate, ctc_loss, att_loss, total loss as the
            reference would log it for this rank's batch).

            Data parallel (one process per GPU, gradients AVERAGED over ranks by parallel.DataParallelEngine): for the
            update to equal the single-process step on the GLOBAL batch (SURVEY §8e cond. 1, 2)
              * CTCLoss 'mean' = mean over utterances of nll_b / len_b  -> weight B_rank / (B_global / world);
              * CrossEntropy(ignore_index=0) 'mean' = mean over non-pad tokens -> weight N_rank / (N_global / world).
            Both counts travel in ONE scalar all-reduce; the logged losses stay the rank's own unweighted ones. '''
        total_loss, shown_loss, ctc_loss, att_loss = 0, 0, None, None
        w_ctc = w_att = None
        if self.dp is not None:
            counts = torch.stack([txt_len.new_tensor(txt.shape[0]), txt_len.sum()]).to(torch.float64)
            w = (counts / self.dp.count_normaliser(counts)).to(torch.float32)
            w_ctc, w_att = w[0], w[1]
        if ctc_output is not None:
            ctc_loss = self.ctc_loss(ctc_output.transpose(0, 1), txt, encode_len, txt_len)
            shown_loss = shown_loss + ctc_loss.detach() * self.model.ctc_weight
            total_loss += (ctc_loss if w_ctc is None else ctc_loss * w_ctc) * self.model.ctc_weight
        if att_output is not None:
            b, t, _ = att_output.shape
            att_loss = self.seq_loss(att_output.view(b * t, -1), txt.view(-1))
            shown_loss = shown_loss + att_loss.detach() * (1 - self.model.ctc_weight)
            total_loss += (att_loss if w_att is None else att_loss * w_att) * (1 - self.model.ctc_weight)
        return total_loss, ctc_loss, att_loss, shown_loss

    def exec(self):
        ''' Training End-to-end ASR system '''
        self.verbose('Total training steps {}.'.format(human_format(self.max_step)))
        ctc_loss, att_loss = None, None
        n_epochs = 0
        self.timer.set()
        while self.step < self.max_step:
            # Renew dataloader to enable random sampling
            if self.curriculum > 0 and n_epochs == self.curriculum:
                self.verbose('Curriculum learning ends after {} epochs, starting random sampling.'.format(n_epochs))
                self.tr_set, _, _, _, _, _ = load_dataset(self.paras.njobs, self.paras.gpu,
                                                          self.paras.pin_memory, False, **self.config['data'])
            if hasattr(self.tr_set.sampler, 'set_epoch'):     # data parallel: the shared shuffle of this epoch
                self.tr_set.sampler.set_epoch(n_epochs)
            for data in self.tr_set:
                # Pre-step : update tf_rate/lr_rate and do zero_grad
                tf_rate = self.optimizer.pre_step(self.step)
                feat, feat_len, txt, txt_len = self.fetch_data(data)
                self.timer.cnt('rd')

                # Note: txt should NOT start w/ <sos>
                ctc_output, encode_len, att_output, att_align, dec_state = \
                    self.model(feat, feat_len, self.decode_step, tf_rate=tf_rate, teacher=txt)
                total_loss, ctc_loss, att_loss, shown_loss = \
                    self.compute_losses(ctc_output, encode_len, att_output, txt, txt_len)
                self.timer.cnt('fw')

                grad_norm = self.backward(total_loss)
                self.step += 1
                self.poll_device_errors()

                if (self.step == 1) or (self.step % self.PROGRESS_STEP == 0):
                    self.progress('Tr stat | Loss - {:.2f} | Grad. Norm - {:.2f} | {}'
                                  .format(shown_loss.detach().cpu().item(), float(grad_norm), self.timer.show()))
                    self.write_log('loss', {'tr_ctc': ctc_loss, 'tr_att': att_loss})
                    self.write_log('wer', {'tr_att': cal_er(self.tokenizer, att_output, txt),
                                           'tr_ctc': cal_er(self.tokenizer, ctc_output, txt, ctc=True)})

                if (self.step == 1) or (self.step % self.valid_step == 0):
                    self.validate()

                self.timer.set()
                if self.step > self.max_step:
                    break
            n_epochs += 1
        self.poll_device_errors(force=True)
        if self.log is not None:
            self.log.close()

    def validate(self):
        self.poll_device_errors(force=True)     # a validation score / checkpoint of parameters that are known good
        self.model.eval()
        dev_wer = {'att': [], 'ctc': []}
        for i, data in enumerate(self.dv_set):
            self.progress('Valid step - {}/{}'.format(i + 1, len(self.dv_set)))
            feat, feat_len, txt, txt_len = self.fetch_data(data)
            with torch.no_grad():
                ctc_output, encode_len, att_output, att_align, dec_state = \
                    self.model(feat, feat_len, int(self.decode_step * self.DEV_STEP_RATIO))
            dev_wer['att'].append(cal_er(self.tokenizer, att_output, txt))
            dev_wer['ctc'].append(cal_er(self.tokenizer, ctc_output, txt, ctc=True))

            # Log some examples
            if i == len(self.dv_set) // 2:
                for j in range(min(len(txt), self.DEV_N_EXAMPLE)):
                    if self.step == 1:
                        self.write_log('true_text{}'.format(j), self.tokenizer.decode(txt[j].tolist()))
                    if att_output is not None:
                        self.write_log('att_text{}'.format(j), self.tokenizer.decode(
                            ops.argmax(att_output[j]).tolist()))
                    if ctc_output is not None:
                        self.write_log('ctc_text{}'.format(j), self.tokenizer.decode(
                            ops.argmax(ctc_output[j]).tolist(), ignore_repeat=True))

        # Ckpt if performance improves
        for task in ['att', 'ctc']:
            dev_wer[task] = sum(dev_wer[task]) / len(dev_wer[task])
            if dev_wer[task] < self.best_wer[task]:
                self.best_wer[task] = dev_wer[task]
                self.save_checkpoint('best_{}.pth'.format(task), 'wer', dev_wer[task])
            self.write_log('wer', {'dv_' + task: dev_wer[task]})
        self.save_checkpoint('latest.pth', 'wer', dev_wer['att'], show_msg=False)
        self.model.train()
