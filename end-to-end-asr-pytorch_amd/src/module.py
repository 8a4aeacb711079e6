"""Encoder building blocks + attention score kernels — MI355X mirror of the reference's
src/module.py (same class names, constructor arguments, attribute and state_dict key names).

All arithmetic goes through ``ops`` (hand-written gfx950 kernels behind the libasrk C ABI).
"""
import math

import torch
import torch.nn as nn

from .. import ops


class RNNParams(nn.Module):
    """Parameter container with exactly nn.LSTM / nn.GRU's parameter names, shapes, registration
    order and default init (uniform(-1/sqrt(H), 1/sqrt(H)) in registration order), so checkpoints
    and from-seed initialisation match the reference's ``nn.LSTM`` modules
    (src/module.py:112-113, src/asr.py:175-176) key-for-key.  It never runs ATen's RNN."""

    def __init__(self, mode, input_size, hidden_size, num_layers=1, bidirectional=False,
                 dropout=0.0, batch_first=True):
        super().__init__()
        assert mode in ('LSTM', 'GRU')
        self.mode = mode
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.bidirectional = bidirectional
        self.dropout = dropout
        self.batch_first = batch_first
        gate = 4 if mode == 'LSTM' else 3
        ndir = 2 if bidirectional else 1
        for layer in range(num_layers):
            in_sz = input_size if layer == 0 else hidden_size * ndir
            for d in range(ndir):
                sfx = '_reverse' if d == 1 else ''
                self.register_parameter('weight_ih_l{}{}'.format(layer, sfx),
                                        nn.Parameter(torch.empty(gate * hidden_size, in_sz)))
                self.register_parameter('weight_hh_l{}{}'.format(layer, sfx),
                                        nn.Parameter(torch.empty(gate * hidden_size, hidden_size)))
                self.register_parameter('bias_ih_l{}{}'.format(layer, sfx),
                                        nn.Parameter(torch.empty(gate * hidden_size)))
                self.register_parameter('bias_hh_l{}{}'.format(layer, sfx),
                                        nn.Parameter(torch.empty(gate * hidden_size)))
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.hidden_size) if self.hidden_size > 0 else 0
        for weight in self.parameters():
            nn.init.uniform_(weight, -stdv, stdv)

    def flatten_parameters(self):
        """API parity with nn.LSTM (src/module.py:127-128); see colocate_directions."""
        self.colocate_directions()
        return None

    def colocate_directions(self):
        """weight_ih of the two directions of a layer as the two halves of ONE device buffer ([2 * gate * H, in]).
        Both directions multiply the same input, so the input projection runs as one GEMM over the stacked weights
        (ops.LSTMLayerFn); with the parameters already adjacent that stack is a view - rounds 1-5 copied 2 x 4H x Din
        floats per layer and step into a fresh buffer (1.1 GB of device copies per cfg3 step, 0.5 ms).  Parameter
        objects, names, shapes and values are unchanged (only `.data` is re-pointed), so optimiser state, hooks and
        state_dict keys stay valid; moving the module (`.to()`) re-runs this through `_apply`."""
        if not self.bidirectional:
            return
        for layer in range(self.num_layers):
            wf = getattr(self, 'weight_ih_l{}'.format(layer))
            wr = getattr(self, 'weight_ih_l{}_reverse'.format(layer))
            if not (wf.is_cuda and wr.is_cuda and wf.dtype == wr.dtype and wf.shape == wr.shape):
                continue
            n = wf.numel()
            if (wf.is_contiguous() and wr.is_contiguous()
                    and wf.untyped_storage().data_ptr() == wr.untyped_storage().data_ptr()
                    and wr.storage_offset() == wf.storage_offset() + n):
                continue
            with torch.no_grad():
                buf = torch.empty((2 * wf.shape[0], wf.shape[1]), dtype=wf.dtype, device=wf.device)
                buf[:wf.shape[0]].copy_(wf)
                buf[wf.shape[0]:].copy_(wr)
                wf.data = buf[:wf.shape[0]]
                wr.data = buf[wf.shape[0]:]

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.colocate_directions()
        return out

    def layer_params(self, layer, reverse=False):
        sfx = '_reverse' if reverse else ''
        return tuple(getattr(self, '{}_l{}{}'.format(n, layer, sfx))
                     for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'))

    def extra_repr(self):
        return '{}, {}, {}, num_layers={}, bidirectional={}'.format(
            self.mode, self.input_size, self.hidden_size, self.num_layers, self.bidirectional)


class VGGExtractor(nn.Module):
    ''' VGG extractor for ASR (reference: src/module.py:7-66; https://arxiv.org/pdf/1706.02737.pdf).
    `self.extractor` is the reference's nn.Sequential (same parameter names `extractor.{0,2,5,7}.*`,
    shapes and default init) used purely as a parameter container: the arithmetic is
    im2col + MFMA GEMM (+fused ReLU) and 2x2 max-pool kernels on channels-last activations. '''

    def __init__(self, input_dim):
        super(VGGExtractor, self).__init__()
        self.init_dim = 64
        self.hide_dim = 128
        in_channel, freq_dim, out_dim = self.check_dim(input_dim)
        self.in_channel = in_channel
        self.freq_dim = freq_dim
        self.out_dim = out_dim

        self.extractor = nn.Sequential(
            nn.Conv2d(in_channel, self.init_dim, 3, stride=1, padding=1),
            nn.ReLU(),
            nn.Conv2d(self.init_dim, self.init_dim, 3, stride=1, padding=1),
            nn.ReLU(),
            nn.MaxPool2d(2, stride=2),  # Half-time dimension
            nn.Conv2d(self.init_dim, self.hide_dim, 3, stride=1, padding=1),
            nn.ReLU(),
            nn.Conv2d(self.hide_dim, self.hide_dim, 3, stride=1, padding=1),
            nn.ReLU(),
            nn.MaxPool2d(2, stride=2)  # Half-time dimension
        )

    def check_dim(self, input_dim):
        ''' delta features are stacked over channels: 13k -> MFCC, 40k -> fbank '''
        if input_dim % 13 == 0:
            return int(input_dim / 13), 13, (13 // 4) * self.hide_dim
        elif input_dim % 40 == 0:
            return int(input_dim / 40), 40, (40 // 4) * self.hide_dim
        else:
            raise ValueError('Acoustic feature dimension for VGG should be 13/26/39(MFCC) or '
                             '40/80/120(Fbank) but got ' + str(input_dim))

    def forward_bm2tm(self, feature, feat_len):
        ''' [B,T,C*F] batch-major -> ([T//4, B, 128*(F//4)] time-major, feat_len//4).
        view_input's crop (t % 4) and [B,T,C,F]->[B,C,T,F] transpose (src/module.py:44-57) are
        folded into the first im2col's strides; the final [B,128,T/4,F/4]->[B,T/4,128*F/4]
        transpose (62-65) and the batch/time swap into the last pool's output strides. '''
        from .. import conv_ops as C
        feat_len = feat_len // 4
        bs, ts, ds = feature.shape
        Cin, Fq = self.in_channel, self.freq_dim
        T = ts - ts % 4
        if T < 4:
            raise RuntimeError('VGG prenet needs at least 4 frames, got {}'.format(ts))
        ex = self.extractor
        g = C.Geom(bs, T, Fq, Cin, 3, 3, 1, 1, 1, 1, ts * ds, ds, 1, Fq)          # reads [B,T,C,F] in place
        h = C.conv(feature, ex[0].weight, ex[0].bias, g, relu=True)              # [B*T*F, 64]
        g = C.Geom(bs, T, Fq, 64, 3, 3, 1, 1, 1, 1, T * Fq * 64, Fq * 64, 64, 1)
        h = C.conv(h, ex[2].weight, ex[2].bias, g, relu=True)
        T2, F2 = T // 2, Fq // 2
        h = C.maxpool2x2(h, (bs, T, Fq, 64), (bs * T2 * F2, 64), (T2 * F2 * 64, F2 * 64, 64, 1))
        g = C.Geom(bs, T2, F2, 64, 3, 3, 1, 1, 1, 1, T2 * F2 * 64, F2 * 64, 64, 1)
        h = C.conv(h, ex[5].weight, ex[5].bias, g, relu=True)                    # [B*T2*F2, 128]
        g = C.Geom(bs, T2, F2, 128, 3, 3, 1, 1, 1, 1, T2 * F2 * 128, F2 * 128, 128, 1)
        h = C.conv(h, ex[7].weight, ex[7].bias, g, relu=True)
        T4, F4 = T2 // 2, F2 // 2
        # out[t, b, c*F4 + f]: time-major, channel-major features
        h = C.maxpool2x2(h, (bs, T2, F2, 128), (T4, bs, 128 * F4), (128 * F4, bs * 128 * F4, 1, F4))
        return h, feat_len

    def forward(self, feature, feat_len):
        ''' reference API: BSxTxD -> BSxT/4x(128*D/4) '''
        out, feat_len = self.forward_bm2tm(feature, feat_len)
        return ops.swap_bt(out), feat_len


class CNNExtractor(nn.Module):
    ''' A simple 2-layer CNN extractor for acoustic feature down-sampling (reference:
    src/module.py:68-90; note: no activation between the two convolutions).  Conv1d over time is
    the KW=1 case of the channels-last convolution: [T(,B), Din] patches x weight.view(out, Din*4). '''

    def __init__(self, input_dim, out_dim):
        super(CNNExtractor, self).__init__()
        self.out_dim = out_dim
        self.extractor = nn.Sequential(
            nn.Conv1d(input_dim, out_dim, 4, stride=2, padding=1),
            nn.Conv1d(out_dim, out_dim, 4, stride=2, padding=1),
        )

    def forward_bm2tm(self, feature, feat_len):
        ''' [B,T,D] batch-major -> ([T', B, out_dim] time-major, feat_len//4).  The "width" axis
        of the convolution geometry is the batch, so GEMM rows come out ordered (t', b). '''
        from .. import conv_ops as C
        feat_len = feat_len // 4
        bs, ts, ds = feature.shape
        ex = self.extractor
        g = C.Geom(1, ts, bs, ds, 4, 1, 2, 1, 1, 0, 0, ds, ts * ds, 1)            # reads [B,T,D] in place
        h = C.conv(feature, ex[0].weight, ex[0].bias, g)                         # [T1*B, out]
        T1, O = g.Ho, self.out_dim
        g = C.Geom(1, T1, bs, O, 4, 1, 2, 1, 1, 0, 0, bs * O, O, 1)
        h = C.conv(h, ex[1].weight, ex[1].bias, g)
        return h.view(g.Ho, bs, O), feat_len

    def forward(self, feature, feat_len):
        out, feat_len = self.forward_bm2tm(feature, feat_len)
        return ops.swap_bt(out), feat_len


class RNNLayer(nn.Module):
    ''' RNN wrapper, includes time-downsampling (reference: src/module.py:93-158) '''

    def __init__(self, input_dim, module, dim, bidirection, dropout, layer_norm, sample_rate,
                 sample_style, proj):
        super(RNNLayer, self).__init__()
        rnn_out_dim = 2 * dim if bidirection else dim
        self.out_dim = sample_rate * rnn_out_dim \
            if sample_rate > 1 and sample_style == 'concat' else rnn_out_dim
        self.dropout = dropout
        self.layer_norm = layer_norm
        self.sample_rate = sample_rate
        self.sample_style = sample_style
        self.proj = proj
        self.bidirection = bidirection

        if self.sample_style not in ['drop', 'concat']:
            raise ValueError('Unsupported Sample Style: ' + self.sample_style)
        if proj and sample_rate > 1 and sample_style == 'concat':
            # the reference builds pj = Linear(rnn_out_dim, rnn_out_dim) but feeds it the
            # concatenated sample_rate*rnn_out_dim frames (src/module.py:123,152-156): its first
            # forward dies with a shape error.  Fail at construction with the reason instead.
            raise ValueError("proj=True cannot follow sample_style='concat' with sample_rate>1 "
                             "(projection expects {} features, concat yields {})".format(
                                 rnn_out_dim, sample_rate * rnn_out_dim))
        if module.upper() not in ('LSTM', 'GRU'):
            raise NotImplementedError("encoder module '{}' is not supported (LSTM / GRU)".format(module))
        self.rnn_type = module.upper()

        # Recurrent layer (parameters only; math is ops.lstm_layer / gru_ops.gru_layer)
        self.layer = RNNParams(module.upper(), input_dim, dim, num_layers=1,
                               bidirectional=bidirection, batch_first=True)

        if self.layer_norm:
            self.ln = nn.LayerNorm(rnn_out_dim)
        if self.dropout > 0:
            self.dp = nn.Dropout(p=dropout)
        if self.proj:
            self.pj = nn.Linear(rnn_out_dim, rnn_out_dim)

    def supports_packed(self):
        ''' can run a padded batch with every row computed as if alone and unpadded (forward_tm(packed=True)) '''
        return self.rnn_type == 'LSTM' and self.layer.hidden_size % 4 == 0

    def forward_tm(self, x_tm, x_len, packed=False, frames=None):
        ''' time-major core: x_tm [T,B,D] -> ([T',B,D'], x_len').  packed (inference only): row b is a sequence of
            frames[b] frames (default x_len[b]) and is computed exactly as if it were encoded alone, unpadded
            (ops.lstm_layer_packed) - how the reference encodes for decoding (src/decode.py:88 on batch 1); returns
            (out, x_len', frames').  `frames` is the length of the tensor a batch-1 run would hold, which the
            reference's LSTM runs over in full; it differs from the REPORTED length x_len after a 'drop' reduction of
            an odd length (ceil(T/r) frames kept, T // r reported: src/module.py:141-146). '''
        pf = self.layer.layer_params(0, False)
        pr = self.layer.layer_params(0, True) if self.bidirection else None
        if packed:
            frames = x_len if frames is None else frames
            fuse = self.sample_rate > 1 and not self.layer_norm
            output = ops.lstm_layer_packed(x_tm, pf, pr, frames,
                                           pyramid=(self.sample_rate, self.sample_style) if fuse else None)
            if self.layer_norm:
                output = ops.layer_norm(output, self.ln.weight, self.ln.bias, self.ln.eps)
            if self.sample_rate > 1:
                r = self.sample_rate
                x_len = x_len // r
                frames = frames // r if self.sample_style == 'concat' else (frames + r - 1) // r
                if not fuse:
                    output = ops.pyramid(output, r, self.sample_style)
            if self.proj:
                output = ops.tanh(ops.linear(output, self.pj.weight, self.pj.bias))
            return output, x_len, frames
        if self.sample_rate > 1 and not self.layer_norm and not (self.dropout > 0 and self.training):
            # nothing sits between the recurrence and the time reduction: the kernel writes the reduced
            # layout itself (and reads its gradient from it)
            if self.rnn_type == 'LSTM':
                output = ops.lstm_layer(x_tm, pf, pr, pyramid=(self.sample_rate, self.sample_style))
            else:
                from .. import gru_ops
                output = gru_ops.gru_layer(x_tm, pf, pr, pyramid=(self.sample_rate, self.sample_style))
            x_len = x_len // self.sample_rate
            if self.proj:
                output = ops.tanh(ops.linear(output, self.pj.weight, self.pj.bias))
            return output, x_len
        if self.rnn_type == 'LSTM':
            output = ops.lstm_layer(x_tm, pf, pr)
        else:
            from .. import gru_ops
            output = gru_ops.gru_layer(x_tm, pf, pr)

        if self.layer_norm:
            output = ops.layer_norm(output, self.ln.weight, self.ln.bias, self.ln.eps)
        if self.dropout > 0:
            output = ops.dropout(output, self.dropout, self.training)

        if self.sample_rate > 1:
            x_len = x_len // self.sample_rate
            output = ops.pyramid(output, self.sample_rate, self.sample_style)

        if self.proj:
            output = ops.tanh(ops.linear(output, self.pj.weight, self.pj.bias))

        return output, x_len

    def forward(self, input_x, x_len):
        ''' batch-major API of the reference: input_x [B,T,D] -> ([B,T',D'], x_len') '''
        out, x_len = self.forward_tm(ops.swap_bt(input_x), x_len)
        return ops.swap_bt(out), x_len


class BaseAttention(nn.Module):
    ''' Base module for attentions (reference: src/module.py:161-195).  The score arithmetic
    (energy -> masked softmax(energy / temperature) -> context) runs in the fused gfx950 kernels
    driven from src/asr.py Attention.forward; these classes keep the reference's parameters,
    attribute names (mask, k_len, prev_att) and memory protocol. '''

    def __init__(self, temperature, num_head):
        super().__init__()
        self.temperature = temperature
        self.num_head = num_head
        self.reset_mem()

    def reset_mem(self):
        self.mask = None
        self.k_len = None

    def set_mem(self, prev_att):
        pass

    def compute_mask(self, k, k_len):
        ''' Padded-frame mask [B*N, T] (True = masked), as src/module.py:179-187 but without the
        host-side numpy loop (no device->host sync).  The kernels mask from k_len directly. '''
        self.k_len = k_len
        bs, ts, _ = k.shape
        idx = torch.arange(ts, device=k_len.device).unsqueeze(0)
        mask = idx >= k_len.to(idx.device).unsqueeze(1)                    # [B,T]
        self.mask = mask.unsqueeze(1).expand(bs, self.num_head, ts).reshape(-1, ts)


class ScaleDotAttention(BaseAttention):
    ''' Scaled Dot-Product Attention (reference: src/module.py:198-212) '''

    def __init__(self, temperature, num_head):
        super().__init__(temperature, num_head)


class LocationAwareAttention(BaseAttention):
    ''' Location-Awared Attention (reference: src/module.py:215-258) '''

    def __init__(self, kernel_size, kernel_num, dim, num_head, temperature):
        super().__init__(temperature, num_head)
        self.prev_att = None
        # parameter holders with the reference's names/shapes:
        # loc_conv.weight [K, N, 2ks+1], loc_proj.weight [A, K], gen_energy.{weight [1,A], bias [1]}
        self.loc_conv = nn.Conv1d(num_head, kernel_num, kernel_size=2 * kernel_size + 1,
                                  padding=kernel_size, bias=False)
        self.loc_proj = nn.Linear(kernel_num, dim, bias=False)
        self.gen_energy = nn.Linear(dim, 1)
        self.dim = dim

    def reset_mem(self):
        super().reset_mem()
        self.prev_att = None

    def set_mem(self, prev_att):
        self.prev_att = prev_att

    def uniform_init(self, bs, ts, device):
        ''' prev_att[b, :, :len_b] = 1/len_b (src/module.py:239-242), built without a host loop '''
        k_len = self.k_len.to(device)
        idx = torch.arange(ts, device=device).unsqueeze(0)
        valid = (idx < k_len.unsqueeze(1)).to(torch.float32)               # [B,T]
        att = valid / k_len.clamp(min=1).to(torch.float32).unsqueeze(1)
        return att.unsqueeze(1).expand(bs, self.num_head, ts).contiguous()
