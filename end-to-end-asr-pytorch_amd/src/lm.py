"""RNN language model used for shallow fusion in decoding — MI355X mirror of the reference's
src/lm.py (same constructor, state_dict keys `emb.*`, `rnn.*`, `trans.*`, same forward contract).
Decode-time calls (one token, carried state) step the cell kernels; whole-sequence calls (LM
training / validation, bin/train_lm.py) run the persistent recurrence kernel per layer.
"""
import torch
import torch.nn as nn

from .. import ops
from .. import decoder_ops as dops
from .module import RNNParams


class RNNLM(nn.Module):
    ''' RNN Language Model (reference: src/lm.py:6-45) '''

    def __init__(self, vocab_size, emb_tying, emb_dim, module, dim, n_layers, dropout):
        super().__init__()
        self.dim = dim
        self.n_layers = n_layers
        self.emb_tying = emb_tying
        if emb_tying:
            assert emb_dim == dim, "Output dim of RNN should be identical to embedding if using weight tying."
        if module.upper() not in ('LSTM', 'GRU'):
            raise NotImplementedError("RNNLM module '{}' is not supported (LSTM / GRU)".format(module))
        self.rnn_type = module.upper()
        self.vocab_size = vocab_size
        self.emb = nn.Embedding(vocab_size, emb_dim)
        self.dp1 = nn.Dropout(dropout)
        self.dp2 = nn.Dropout(dropout)
        self.rnn = RNNParams(module.upper(), emb_dim, dim, num_layers=n_layers, dropout=dropout,
                             batch_first=True)
        if not self.emb_tying:
            self.trans = nn.Linear(dim, vocab_size)

    def create_msg(self):
        return ['Model spec.| RNNLM weight tying = {}, # of layers = {}, dim = {}'.format(
            self.emb_tying, self.n_layers, self.dim)]

    def forward(self, x, lens, hidden=None):
        ''' x [B,L] token ids, lens [B] (all == L at decode time) -> (logits [B,L,V], (h,c) [n_layers,B,dim]) '''
        B, L = x.shape
        dev = self.emb.weight.device
        if hidden is None and (self.training or L > 1):
            return self._forward_sequence(x)
        if self.rnn_type == 'GRU':
            return self._forward_gru(x, hidden)
        if hidden is None:
            h = [torch.zeros((B, self.dim), device=dev) for _ in range(self.n_layers)]
            c = [torch.zeros((B, self.dim), device=dev) for _ in range(self.n_layers)]
        else:
            h = [hidden[0][l].to(dev) for l in range(self.n_layers)]
            c = [hidden[1][l].to(dev) for l in range(self.n_layers)]
        emb_x = dops.embedding(x.to(dev), self.emb.weight)                    # [B,L,E]
        w = self.emb.weight if self.emb_tying else self.trans.weight
        b = None if self.emb_tying else self.trans.bias
        if L == 1 and not torch.is_grad_enabled():
            # one decode position: the cells write the new state straight into the [n_layers,B,dim] tensors that are
            # returned (no stack copies); many rows take the cached-panel GEMMs
            hs = torch.empty((self.n_layers, B, self.dim), dtype=torch.float32, device=dev)
            cs = torch.empty((self.n_layers, B, self.dim), dtype=torch.float32, device=dev)
            inp = emb_x[:, 0, :]
            for l in range(self.n_layers):
                dops.lstm_cell_infer(inp, h[l], c[l], *self.rnn.layer_params(l), out=(hs[l], cs[l]))
                inp = hs[l]
            return dops.linear_infer(inp, w, b).unsqueeze(1), (hs, cs)
        outs = []
        for t in range(L):
            inp = emb_x[:, t, :]
            for l in range(self.n_layers):
                h[l], c[l] = dops.lstm_cell_infer(inp, h[l], c[l], *self.rnn.layer_params(l))
                inp = h[l]
            outs.append(inp)
        top = outs[0].unsqueeze(1) if L == 1 else torch.stack(outs, dim=1)    # [B,L,dim]
        logits = ops.linear(top, w, b)
        return logits, (torch.stack(h, 0), torch.stack(c, 0))

    def _forward_sequence(self, x):
        """whole (padded) sequences from a zero state: language-model training / validation
        (bin/train_lm.py:66-70).  The reference packs the sequences; running the unidirectional
        layers over the zero-padded tail instead changes nothing the loss can see: padded targets are
        ignored (ignore_index=0) and no valid position comes after a padded one.  The persistent
        recurrence kernel does the time loop; returns (logits [B,L,V], None)."""
        from .. import gru_ops
        dev = self.emb.weight.device
        emb_x = ops.dropout(dops.embedding(x.to(dev), self.emb.weight), self.dp1.p, self.training)
        h = ops.swap_bt(emb_x)                                               # [L,B,E] time-major
        for l in range(self.n_layers):
            if self.rnn_type == 'LSTM':
                h = ops.lstm_layer(h, self.rnn.layer_params(l))
            else:
                h = gru_ops.gru_layer(h, self.rnn.layer_params(l))
            if l + 1 < self.n_layers:
                h = ops.dropout(h, self.rnn.dropout, self.training)          # nn.LSTM inter-layer dropout
        top = ops.dropout(ops.swap_bt(h), self.dp2.p, self.training)         # [B,L,dim]
        w = self.emb.weight if self.emb_tying else self.trans.weight
        b = None if self.emb_tying else self.trans.bias
        return ops.linear(top, w, b), None

    def _forward_gru(self, x, hidden):
        """nn.GRU variant: the state is one tensor [n_layers, B, dim] (reference: src/lm.py:38)"""
        from .. import gru_ops
        B, L = x.shape
        dev = self.emb.weight.device
        h = [torch.zeros((B, self.dim), device=dev) if hidden is None else hidden[l].to(dev)
             for l in range(self.n_layers)]
        emb_x = dops.embedding(x.to(dev), self.emb.weight)
        outs = []
        for t in range(L):
            inp = emb_x[:, t, :]
            for l in range(self.n_layers):
                h[l] = gru_ops.gru_cell_infer(inp, h[l], *self.rnn.layer_params(l))
                inp = h[l]
            outs.append(inp)
        top = outs[0].unsqueeze(1) if L == 1 else torch.stack(outs, dim=1)
        w = self.emb.weight if self.emb_tying else self.trans.weight
        b = None if self.emb_tying else self.trans.bias
        return ops.linear(top, w, b), torch.stack(h, 0)
