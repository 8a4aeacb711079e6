"""RNN language-model training solver — mirror of the reference's bin/train_lm.py:9-128 (same
step order, `best_ppx.pth` checkpointing on dev perplexity, `entropy` / `perplexity` logs).  The
model's whole-sequence forward runs the persistent LSTM recurrence kernel per layer, the loss is the
fused cross-entropy kernel, the update the fused optimiser."""
import torch

from .. import ops
from ..src.solver import BaseSolver
from ..src.lm import RNNLM
from ..src.optim import Optimizer
from ..src.data import load_textset
from ..src.util import human_format


class Solver(BaseSolver):
    ''' Solver for training language models'''

    def __init__(self, config, paras, mode):
        super().__init__(config, paras, mode)
        self.best_loss = 10

    def fetch_data(self, data):
        ''' Move data to device, insert <sos> and compute text seq. length'''
        txt = torch.cat((torch.zeros((data.shape[0], 1), dtype=torch.long), data), dim=1).to(self.device)
        txt_len = torch.sum(data != 0, dim=-1)
        return txt, txt_len

    def load_data(self):
        self.tr_set, self.dv_set, self.vocab_size, self.tokenizer, msg = \
            load_textset(self.paras.njobs, self.paras.gpu, self.paras.pin_memory, **self.config['data'])
        self.verbose(msg)

    def set_model(self):
        self.model = RNNLM(self.vocab_size, **self.config['model']).to(self.device)
        self.verbose(self.model.create_msg())
        self.seq_loss = ops.CrossEntropyLoss(ignore_index=0)
        self.optimizer = Optimizer(self.model.parameters(), **self.config['hparas'])
        self.verbose(self.optimizer.create_msg())
        self.load_ckpt()
        self.enable_data_parallel()

    def _loss(self, txt, txt_len, train=False):
        ''' -> (pred, loss to back-propagate, loss as logged: this rank's own token mean, as the reference's) '''
        pred, _ = self.model(txt[:, :-1], txt_len)
        tgt = txt[:, 1:].reshape(-1)
        loss = self.seq_loss(pred.view(-1, self.vocab_size), tgt)
        shown = loss.detach()
        if train and getattr(self, 'dp', None) is not None:
            # CrossEntropy(ignore_index=0) is a mean over THIS rank's non-pad targets; the engine averages
            # gradients over ranks, so weight by n_local / (n_global / world): the update is then the mean over
            # the global batch's tokens, as on one device (same correction as bin/train_asr.py)
            n_tok = (tgt != 0).sum()
            loss = loss * (n_tok / self.dp.token_normaliser(n_tok)).reshape(())
        return pred, loss, shown

    def exec(self):
        self.verbose('Total training steps {}.'.format(human_format(self.max_step)))
        self.timer.set()
        # epoch index from the step counter, so a resumed run continues the shuffle sequence instead of replaying it
        n_epochs = self.step // max(1, len(self.tr_set))
        while self.step < self.max_step:
            if hasattr(self.tr_set.sampler, 'set_epoch'):     # DistributedSampler: reshuffle per epoch
                self.tr_set.sampler.set_epoch(n_epochs)
            n_epochs += 1
            for data in self.tr_set:
                self.optimizer.pre_step(self.step)
                txt, txt_len = self.fetch_data(data)
                self.timer.cnt('rd')
                pred, bp_loss, lm_loss = self._loss(txt, txt_len, train=True)
                self.timer.cnt('fw')
                grad_norm = self.backward(bp_loss)
                self.step += 1
                self.poll_device_errors()
                if self.step % self.PROGRESS_STEP == 0:
                    self.progress('Tr stat | Loss - {:.2f} | Grad. Norm - {:.2f} | {}'
                                  .format(lm_loss.cpu().item(), float(grad_norm), self.timer.show()))
                    self.write_log('entropy', {'tr': lm_loss})
                    self.write_log('perplexity', {'tr': torch.exp(lm_loss.detach()).cpu().item()})
                if (self.step == 1) or (self.step % self.valid_step == 0):
                    self.validate()
                self.timer.set()
                if self.step > self.max_step:
                    break
        self.poll_device_errors(force=True)
        if self.log is not None:
            self.log.close()

    def validate(self):
        self.poll_device_errors(force=True)     # a validation score / checkpoint of parameters that are known good
        self.model.eval()
        dev_loss = []
        for i, data in enumerate(self.dv_set):
            self.progress('Valid step - {}/{}'.format(i + 1, len(self.dv_set)))
            txt, txt_len = self.fetch_data(data)
            with torch.no_grad():
                pred, _, lm_loss = self._loss(txt, txt_len)
            dev_loss.append(lm_loss)
        dev_loss = sum(dev_loss) / len(dev_loss)
        dev_ppx = torch.exp(dev_loss).cpu().item()
        if dev_loss < self.best_loss:
            self.best_loss = dev_loss
            self.save_checkpoint('best_ppx.pth', 'perplexity', dev_ppx)
        self.write_log('entropy', {'dv': dev_loss})
        self.write_log('perplexity', {'dv': dev_ppx})
        for i in range(min(len(txt), self.DEV_N_EXAMPLE)):
            if self.step == 1:
                self.write_log('true_text{}'.format(i), self.tokenizer.decode(txt[i].tolist()))
            self.write_log('pred_text{}'.format(i), self.tokenizer.decode(ops.argmax(pred[i]).tolist()))
        self.model.train()
