"""ASR model (Listener / Attention / Speller + CTC head) — MI355X mirror of the reference's
src/asr.py: same class names, constructor kwargs, attributes, state_dict keys and return tuple.
"""
import torch
import torch.nn as nn

from .. import ops
from .util import init_weights, init_gate
from .module import VGGExtractor, CNNExtractor, RNNLayer


class ASR(nn.Module):
    ''' ASR model, including Encoder/Decoder(s) (reference: src/asr.py:12-155) '''

    def __init__(self, input_size, vocab_size, init_adadelta, ctc_weight, encoder, attention,
                 decoder, emb_drop=0.0):
        super(ASR, self).__init__()

        assert 0 <= ctc_weight <= 1
        self.vocab_size = vocab_size
        self.ctc_weight = ctc_weight
        self.enable_ctc = ctc_weight > 0
        self.enable_att = ctc_weight != 1
        self.lm = None

        self.encoder = Encoder(input_size, **encoder)
        if self.enable_ctc:
            self.ctc_layer = nn.Linear(self.encoder.out_dim, vocab_size)
        if self.enable_att:
            from .decoder import Decoder, Attention  # attention decoder path
            self.dec_dim = decoder['dim']
            self.pre_embed = nn.Embedding(vocab_size, self.dec_dim)
            self.embed_drop = nn.Dropout(emb_drop)
            self.decoder = Decoder(self.encoder.out_dim + self.dec_dim, vocab_size, **decoder)
            query_dim = self.dec_dim * self.decoder.layer
            self.attention = Attention(self.encoder.out_dim, query_dim, **attention)

        if init_adadelta:
            self.apply(init_weights)
            if self.enable_att:
                for l in range(self.decoder.layer):
                    bias = getattr(self.decoder.layers, 'bias_ih_l{}'.format(l))
                    bias = init_gate(bias)

    def set_state(self, prev_state, prev_attn):
        ''' Setting up all memory states for beam decoding'''
        self.decoder.set_state(prev_state)
        self.attention.set_mem(prev_attn)

    def create_msg(self):
        msg = []
        msg.append('Model spec.| Encoder\'s downsampling rate of time axis is {}.'.format(
            self.encoder.sample_rate))
        if self.encoder.vgg:
            msg.append('           | VGG Extractor w/ time downsampling rate = 4 in encoder enabled.')
        if self.encoder.cnn:
            msg.append('           | CNN Extractor w/ time downsampling rate = 4 in encoder enabled.')
        if self.enable_ctc:
            msg.append('           | CTC training on encoder enabled ( lambda = {}).'.format(
                self.ctc_weight))
        if self.enable_att:
            msg.append('           | {} attention decoder enabled ( lambda = {}).'.format(
                self.attention.mode, 1 - self.ctc_weight))
        return msg

    def forward(self, audio_feature, feature_len, decode_step, tf_rate=0.0, teacher=None,
                emb_decoder=None, get_dec_state=False):
        ''' Same contract as the reference (src/asr.py:72-155): returns
            (ctc_output [B,T',V] log-probs | None, encode_len [B], att_output [B,L,V] logits | None,
             att_seq [B,N,L,T'] | None, dec_state [B,L,D] | None) '''
        bs = audio_feature.shape[0]
        ctc_output, att_output, att_seq = None, None, None
        dec_state = [] if get_dec_state else None

        encode_feature, encode_len = self.encoder(audio_feature, feature_len)

        if self.enable_ctc:
            ctc_output = ops.log_softmax(
                ops.linear(encode_feature, self.ctc_layer.weight, self.ctc_layer.bias))

        if self.enable_att:
            att_output, att_seq, dec_state = self._attention_decode(
                bs, encode_feature, encode_len, decode_step, tf_rate, teacher, emb_decoder,
                get_dec_state)

        return ctc_output, encode_len, att_output, att_seq, dec_state

    def _attention_decode(self, bs, encode_feature, encode_len, decode_step, tf_rate, teacher,
                          emb_decoder, get_dec_state):
        from .decoder import run_decoder_loop
        return run_decoder_loop(self, bs, encode_feature, encode_len, decode_step, tf_rate,
                                teacher, emb_decoder, get_dec_state)


class Encoder(nn.Module):
    ''' Encoder (a.k.a. Listener in LAS) (reference: src/asr.py:316-366).  Layers are chained in a
    time-major layout internally ([T,B,D], what the persistent recurrence kernels want); the
    module's own interface stays batch-major like the reference. '''

    def __init__(self, input_size, prenet, module, bidirection, dim, dropout, layer_norm, proj,
                 sample_rate, sample_style):
        super(Encoder, self).__init__()

        self.vgg = prenet == 'vgg'
        self.cnn = prenet == 'cnn'
        self.sample_rate = 1
        assert len(sample_rate) == len(dropout), 'Number of layer mismatch'
        assert len(dropout) == len(dim), 'Number of layer mismatch'
        num_layers = len(dim)
        assert num_layers >= 1, 'Encoder should have at least 1 layer'

        module_list = []
        input_dim = input_size

        if self.vgg:
            vgg_extractor = VGGExtractor(input_size)
            module_list.append(vgg_extractor)
            input_dim = vgg_extractor.out_dim
            self.sample_rate = self.sample_rate * 4
        if self.cnn:
            cnn_extractor = CNNExtractor(input_size, out_dim=dim[0])
            module_list.append(cnn_extractor)
            input_dim = cnn_extractor.out_dim
            self.sample_rate = self.sample_rate * 4

        if module in ['LSTM', 'GRU']:
            for l in range(num_layers):
                module_list.append(RNNLayer(input_dim, module, dim[l], bidirection, dropout[l],
                                            layer_norm[l], sample_rate[l], sample_style, proj[l]))
                input_dim = module_list[-1].out_dim
                self.sample_rate = self.sample_rate * sample_rate[l]
        else:
            raise NotImplementedError

        self.in_dim = input_size
        self.out_dim = input_dim
        self.layers = nn.ModuleList(module_list)

    def forward(self, input_x, enc_len):
        x = ops.swap_bt(input_x)  # [B,T,D] -> [T,B,D] once
        for _, layer in enumerate(self.layers):
            x, enc_len = layer.forward_tm(x, enc_len)
        return ops.swap_bt(x), enc_len
