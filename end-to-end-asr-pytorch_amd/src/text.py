"""Text encoders (character / subword / word) — host-side mirror of the reference's src/text.py so
that configs, vocab files and checkpoints carry over (<pad>=0, <eos>=1, <unk>=2; reference:
src/text.py:13-31).  The BERT-vocabulary encoder is out of scope.
"""

class _BaseTextEncoder:
    """What every encoder shares: the special ids and the hypothesis clean-up.  A concrete encoder provides
    encode(str) -> ids (+ <eos>), decode(ids, ignore_repeat) -> str, vocab_size, token_type and load_from_file."""
    pad_idx, eos_idx, unk_idx = 0, 1, 2

    def _crop(self, ids, ignore_repeat):
        ''' token ids up to (excluding) the first <eos>, pads dropped, CTC repeats merged on request
            (the rule every reference decoder shares: src/text.py:54-65) '''
        out, prev = [], None
        for t, i in enumerate(ids):
            if i == self.eos_idx:
                break
            if i != self.pad_idx and not (ignore_repeat and t > 0 and i == prev):
                out.append(i)
            prev = i
        return out

    def __repr__(self):
        return "<{} vocab_size={}>".format(type(self).__name__, self.vocab_size)


class CharacterTextEncoder(_BaseTextEncoder):
    ''' (reference: src/text.py:34-93) '''
    _joiner = ""

    def __init__(self, vocab_list):
        self._vocab_list = ["<pad>", "<eos>", "<unk>"] + list(vocab_list)
        self._vocab2idx = {v: i for i, v in enumerate(self._vocab_list)}

    def _tokens(self, s):
        return list(s.strip("\r\n "))

    def encode(self, s):
        return [self.vocab_to_idx(v) for v in self._tokens(s)] + [self.eos_idx]

    def decode(self, ids, ignore_repeat=False):
        return self._joiner.join(self.idx_to_vocab(i) for i in self._crop(ids, ignore_repeat))

    @classmethod
    def load_from_file(cls, vocab_file):
        with open(vocab_file, "r") as f:
            # no .strip(): the character vocabulary holds a space token
            return cls([line.strip("\r\n") for line in f])

    @property
    def vocab_size(self):
        return len(self._vocab_list)

    @property
    def token_type(self):
        return 'character'

    def vocab_to_idx(self, vocab):
        return self._vocab2idx.get(vocab, self.unk_idx)

    def idx_to_vocab(self, idx):
        return self._vocab_list[idx]


class WordTextEncoder(CharacterTextEncoder):
    ''' (reference: src/text.py:128-152) '''
    _joiner = " "

    def _tokens(self, s):
        return s.strip("\r\n ").split(" ")

    @property
    def token_type(self):
        return 'word'


class SubwordTextEncoder(_BaseTextEncoder):
    ''' sentencepiece BPE (reference: src/text.py:96-125) '''

    def __init__(self, spm):
        if spm.pad_id() != 0 or spm.eos_id() != 1 or spm.unk_id() != 2:
            raise ValueError("sentencepiece model must be trained with --pad_id=0 --eos_id=1 --unk_id=2 "
                             "--bos_id=-1 --model_type=bpe --eos_piece=<eos>")
        self.spm = spm

    def encode(self, s):
        # the reference relies on set_encode_extra_options(":eos"); passing it per call gives the
        # same ids on every sentencepiece release
        return self.spm.encode(s, add_eos=True)

    def decode(self, ids, ignore_repeat=False):
        return self.spm.decode([int(i) for i in self._crop(ids, ignore_repeat)])

    @classmethod
    def load_from_file(cls, filepath):
        import sentencepiece as splib
        spm = splib.SentencePieceProcessor()
        spm.load(filepath)
        return cls(spm)

    @property
    def vocab_size(self):
        return len(self.spm)

    @property
    def token_type(self):
        return 'subword'


def load_text_encoder(mode, vocab_file):
    ''' (reference: src/text.py:222-233) '''
    if mode == "character":
        return CharacterTextEncoder.load_from_file(vocab_file)
    elif mode == "subword":
        return SubwordTextEncoder.load_from_file(vocab_file)
    elif mode == "word":
        return WordTextEncoder.load_from_file(vocab_file)
    raise NotImplementedError("`{}` is not yet supported.".format(mode))
