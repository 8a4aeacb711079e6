"""LibriSpeech-layout dataset (<split>/<speaker>/<chapter>/<utt>.{wav,flac} + <chapter>.trans.txt) —
host-side mirror of the reference's corpus/librispeech.py:30-66 (same constructor, same
text-length ordering, same bucket indexing rule).

Differences, neither visible downstream: audio is looked up as `.wav` first, then `.flac` (the stock
LibriSpeech format, read by the native decoder behind src/audio.load_wav), and the transcripts of a
chapter are read once and cached instead of re-opened per utterance.
"""
from os.path import join
from pathlib import Path

from torch.utils.data import Dataset

AUDIO_SUFFIXES = ('*.wav', '*.flac')

_TRANS_CACHE = {}


def read_text(file):
    ''' transcription of one utterance file (reference: corpus/librispeech.py:15-27) '''
    src_file = '-'.join(file.split('-')[:-1]) + '.trans.txt'
    idx = file.split('/')[-1].split('.')[0]
    table = _TRANS_CACHE.get(src_file)
    if table is None:
        table = {}
        with open(src_file, 'r') as fp:
            for line in fp:
                key, _, txt = line.rstrip('\n').partition(' ')
                table[key] = txt
        _TRANS_CACHE[src_file] = table
    return table.get(idx)


class LibriDataset(Dataset):
    def __init__(self, path, split, tokenizer, bucket_size, ascending=False):
        self.path = path
        self.bucket_size = bucket_size
        file_list = []
        for s in split:
            split_list = []
            for pat in AUDIO_SUFFIXES:
                split_list = sorted(Path(join(path, s)).rglob(pat))
                if len(split_list) > 0:
                    break
            assert len(split_list) > 0, "No data found @ {}".format(join(path, s))
            file_list += split_list
        text = [tokenizer.encode(read_text(str(f))) for f in file_list]
        # longest transcript first unless ascending (curriculum); python's sort is stable
        order = sorted(range(len(text)), key=lambda i: len(text[i]), reverse=not ascending)
        self.file_list = tuple(file_list[i] for i in order)
        self.text = tuple(text[i] for i in order)

    def __getitem__(self, index):
        if self.bucket_size > 1:
            # a bucket of neighbours in length order; the tail clamps to the last full bucket
            index = min(len(self.file_list) - self.bucket_size, index)
            return list(zip(self.file_list[index:index + self.bucket_size],
                            self.text[index:index + self.bucket_size]))
        return self.file_list[index], self.text[index]

    def __len__(self):
        return len(self.file_list)


# Additional (official) text source provided with LibriSpeech, and how many of its longest lines to drop
OFFICIAL_TXT_SRC = ['librispeech-lm-norm.txt']
REMOVE_TOP_N_TXT = 5000000


class LibriTextDataset(Dataset):
    ''' transcripts (and optionally the official LM text file) as token-id lists, longest first
        (reference: corpus/librispeech.py:69-125).  Lines of the big text file are encoded lazily. '''

    def __init__(self, path, split, tokenizer, bucket_size):
        self.path = path
        self.bucket_size = bucket_size
        self.encode_on_fly = False
        file_list, all_sent = [], []
        for s in split:
            if s in OFFICIAL_TXT_SRC:
                self.encode_on_fly = True
                with open(join(path, s), 'r') as f:
                    all_sent += f.readlines()
                continue
            for pat in AUDIO_SUFFIXES:
                found = sorted(Path(join(path, s)).rglob(pat))
                if found:
                    file_list += found
                    break
        assert (len(file_list) > 0) or (len(all_sent) > 0), "No data found @ {}".format(path)
        all_sent.extend(read_text(str(f)) for f in file_list)
        if self.encode_on_fly:
            self.tokenizer = tokenizer
            self.text = all_sent
        else:
            self.text = [tokenizer.encode(txt) for txt in all_sent]
        self.text = sorted(self.text, reverse=True, key=lambda x: len(x))
        if self.encode_on_fly:
            del self.text[:REMOVE_TOP_N_TXT]

    def _encoded(self, i):
        if self.encode_on_fly and type(self.text[i]) is str:
            self.text[i] = self.tokenizer.encode(self.text[i])
        return self.text[i]

    def __getitem__(self, index):
        if self.bucket_size > 1:
            index = min(len(self.text) - self.bucket_size, index)
            return [self._encoded(i) for i in range(index, index + self.bucket_size)]
        return self._encoded(index)

    def __len__(self):
        return len(self.text)
