"""On-the-fly audio feature pipeline — MI355X mirror of the reference's src/audio.py
(ExtractAudioFeature -> Delta -> CMVN -> Postprocess, built by create_transform).

Same module names, constructor arguments, tensor shapes between stages ([C, D, T]) and final
[T, C*D] output; the arithmetic runs in the gfx950 kernels of csrc/audio.hip (+ two MFMA GEMMs).
`torchaudio` is not required: 16-bit PCM wav is read with the stdlib `wave` module
(torchaudio.load semantics: float32 in [-1, 1), [channel, samples]).
"""
import math
import wave as _wave

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .. import ops
from ..ops import _L, _p, _stream, _f32c

FLT_EPS = 1.1920928955078125e-07


def load_flac(filepath, verify_md5=True):
    """FLAC (what LibriSpeech ships) -> (FloatTensor [C, N] in [-1,1), sample_rate) through the native
    decoder in libasrk (csrc/flac.cpp: frame CRCs are checked there; the STREAMINFO MD5 of the decoded
    audio is checked here when the encoder stored one)."""
    import ctypes
    import hashlib
    lib = _lib.load()
    path = str(filepath).encode()
    sr, nch, bps, total = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int64(0)
    md5 = (ctypes.c_uint8 * 16)()
    _lib.check(lib.asrk_flac_info(path, ctypes.byref(sr), ctypes.byref(nch), ctypes.byref(bps),
                                  ctypes.byref(total), md5), "flac_info(%s)" % filepath)
    import os
    cap = total.value if total.value > 0 else os.path.getsize(filepath) * 8 // max(1, nch.value)
    buf = np.empty((cap, nch.value), dtype=np.int32)
    got = ctypes.c_int64(0)
    rc = lib.asrk_flac_decode_i32(path, buf.ctypes.data_as(ctypes.c_void_p), cap, ctypes.byref(got))
    if rc == -3:        # ASRK_EWORKSPACE: a stream without a sample count that outgrew the guess; got = size needed
        cap = got.value
        buf = np.empty((cap, nch.value), dtype=np.int32)
        rc = lib.asrk_flac_decode_i32(path, buf.ctypes.data_as(ctypes.c_void_p), cap, ctypes.byref(got))
    _lib.check(rc, "flac_decode(%s)" % filepath)
    buf = buf[:got.value]
    if total.value > 0 and got.value != total.value:
        raise ValueError('%s: decoded %d of %d samples' % (filepath, got.value, total.value))
    if verify_md5 and any(md5):
        nbytes = (bps.value + 7) // 8
        raw = buf.astype('<i%d' % nbytes).tobytes() if nbytes in (1, 2, 4) else \
            b''.join(int(v).to_bytes(3, 'little', signed=True) for v in buf.reshape(-1))
        if hashlib.md5(raw).digest() != bytes(md5):
            raise ValueError('%s: MD5 of the decoded audio does not match the FLAC STREAMINFO' % filepath)
    x = buf.astype(np.float32).T / float(1 << (bps.value - 1))
    return torch.from_numpy(np.ascontiguousarray(x)), sr.value


def load_wav(filepath):
    """-> (FloatTensor [C, N] in [-1,1), sample_rate)   (what torchaudio.load returns, audio.py:102);
    16-bit PCM .wav through the standard library, .flac through the native decoder"""
    if str(filepath).lower().endswith('.flac'):
        return load_flac(filepath)
    with _wave.open(filepath, 'rb') as w:
        sr, nch, sw, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if sw != 2:
        raise ValueError('only 16-bit PCM .wav and .flac files are supported')
    x = np.frombuffer(raw, dtype='<i2').astype(np.float32).reshape(-1, nch).T / 32768.0
    return torch.from_numpy(np.ascontiguousarray(x)), sr


class _FbankTables:
    """Immutable per-(geometry, device) tables: povey window, DFT cos|sin basis, mel weights."""
    _cache = {}

    @classmethod
    def get(cls, sample_rate, frame_length, frame_shift, num_mel_bins, low_freq, high_freq, device):
        key = (sample_rate, frame_length, frame_shift, num_mel_bins, low_freq, high_freq, str(device))
        t = cls._cache.get(key)
        if t is None:
            t = cls(sample_rate, frame_length, frame_shift, num_mel_bins, low_freq, high_freq, device)
            cls._cache[key] = t
        return t

    def __init__(self, sr, frame_length, frame_shift, nmel, low_freq, high_freq, device):
        self.win = int(sr * frame_length * 0.001)
        self.shift = int(sr * frame_shift * 0.001)
        self.padded = 1 << (self.win - 1).bit_length()           # round_to_power_of_two
        self.ldf = (self.win + 3) // 4 * 4
        nb = self.padded // 2 + 1
        self.nb = (nb + 3) // 4 * 4                               # padded bin count (GEMM alignment)
        n = np.arange(self.win, dtype=np.float64)
        window = np.hanning(self.win) ** 0.85 if self.win > 1 else np.ones(1)   # povey
        # real DFT of the zero-padded frame: X[k] = sum_n x[n] e^{-2 pi i k n / padded}
        k = np.arange(nb, dtype=np.float64)
        ang = 2.0 * math.pi * np.outer(n, k) / self.padded
        basis = np.zeros((self.ldf, 2 * self.nb))
        basis[:self.win, :nb] = np.cos(ang)
        basis[:self.win, self.nb:self.nb + nb] = -np.sin(ang)
        # mel banks (Kaldi / torchaudio get_mel_banks, vtln_warp = 1)
        nyq = 0.5 * sr
        hi = high_freq + nyq if high_freq <= 0 else high_freq
        mel = lambda f: 1127.0 * np.log(1.0 + np.asarray(f, np.float64) / 700.0)
        ml, mh = mel(low_freq), mel(hi)
        delta = (mh - ml) / (nmel + 1)
        b = np.arange(nmel, dtype=np.float64)[:, None]
        left, center, right = ml + b * delta, ml + (b + 1) * delta, ml + (b + 2) * delta
        mf = mel(sr / self.padded * np.arange(self.padded // 2, dtype=np.float64))[None, :]
        w = np.maximum(0.0, np.minimum((mf - left) / (center - left), (right - mf) / (right - center)))
        melT = np.zeros((self.nb, nmel))
        melT[:self.padded // 2, :] = w.T                          # bin padded/2 has zero weight
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
        self.window, self.basis, self.melT = f32(window), f32(basis), f32(melT)
        self.nmel = nmel
        # tables of the fused kernel (asrk_fbank_logmel_batch_f32): the zero-padded real DFT of `padded` points runs as a
        # complex FFT of padded / 2 points; twiddles in float64, rounded once
        self.log2n = self.padded.bit_length() - 1
        M = self.padded // 2
        km = np.arange(M, dtype=np.float64)
        ku = np.arange(M + 1, dtype=np.float64)
        pair = lambda ang: np.stack([np.cos(ang), -np.sin(ang)], axis=1)
        self.tw_fft = f32(pair(2.0 * math.pi * km / M))
        self.tw_unpack = f32(pair(2.0 * math.pi * ku / self.padded))
        rng = np.zeros((nmel, 2), dtype=np.int32)
        for m_ in range(nmel):
            nz = np.flatnonzero(melT[:, m_] != 0.0)
            rng[m_] = (nz[0], nz[-1] + 1) if len(nz) else (0, 0)
        self.mel_range = torch.from_numpy(rng).to(device)
        self.fused = 8 <= self.log2n <= 10


def kaldi_fbank(waveform, sample_frequency, num_mel_bins=23, frame_length=25.0, frame_shift=10.0,
                dither=0.0, channel=-1, preemphasis_coefficient=0.97, remove_dc_offset=True,
                low_freq=20.0, high_freq=0.0, **unused):
    """torchaudio.compliance.kaldi.fbank restated for the device (subset used at audio.py:104-108):
    waveform [C, N] on the GPU -> log-mel energies [m, num_mel_bins]."""
    if dither != 0.0:
        raise NotImplementedError('dither > 0 is random; the shipped configs set dither: 0')
    if unused:
        raise NotImplementedError('unsupported fbank options: %s' % sorted(unused))
    ops._require_gpu(waveform)
    L = _L()
    x = _f32c(waveform[max(channel, 0)])
    tb = _FbankTables.get(int(sample_frequency), frame_length, frame_shift, num_mel_bins, low_freq,
                          high_freq, x.device)
    n = x.numel()
    m = 0 if n < tb.win else 1 + (n - tb.win) // tb.shift
    dev = x.device
    if m == 0:
        return torch.empty((0, num_mel_bins), dtype=torch.float32, device=dev)
    if tb.fused:
        return _pcm_to_logmel(x, 4, n, None, torch.tensor([0, m], dtype=torch.int64, device=dev), 1, m, m, tb,
                              num_mel_bins, 1.0, preemphasis_coefficient, remove_dc_offset)
    frames = torch.empty((m, tb.ldf), dtype=torch.float32, device=dev)
    _lib.check(L.asrk_fbank_frames_f32(_p(x), n, _p(tb.window), _p(frames), m, tb.win, tb.shift, tb.ldf,
                                       preemphasis_coefficient, int(remove_dc_offset), _stream()),
               'fbank_frames')
    return _frames_to_logmel(frames, tb, num_mel_bins)


def _pcm_to_logmel(wave, sample_bytes, ld_wave, n_host, frame_off, B, max_m, rows, tb, num_mel_bins, scale, preemph,
                   remove_dc):
    """padded PCM batch [B, ld_wave] -> log-mel energies [sum m, num_mel_bins] in ONE launch (csrc/audio.hip
    fbank_logmel_batch_kernel: framing, FFT in LDS, power, mel weights, log) - no frames / spectrum / power tensors"""
    mel = torch.empty((rows, num_mel_bins), dtype=torch.float32, device=wave.device)
    _lib.check(_L().asrk_fbank_logmel_batch_f32(_p(wave), sample_bytes, ld_wave,
                                                n_host.ctypes.data if n_host is not None else None, _p(frame_off), B,
                                                max_m, _p(tb.window), _p(tb.tw_fft), _p(tb.tw_unpack), _p(tb.melT),
                                                _p(tb.mel_range), num_mel_bins, num_mel_bins, _p(mel), tb.win, tb.shift,
                                                tb.log2n, float(scale), float(preemph), int(remove_dc), FLT_EPS,
                                                _stream()), 'fbank_logmel_batch')
    return mel


def _frames_to_logmel(frames, tb, num_mel_bins):
    """windowed frames [rows, ldf] (any number of utterances stacked) -> log-mel energies [rows, num_mel_bins]:
    DFT as a GEMM against the cos|sin basis, |X|^2, mel weights as a second GEMM, log(max(., FLT_EPSILON))"""
    L = _L()
    m, dev = frames.shape[0], frames.device
    spec = torch.empty((m, 2 * tb.nb), dtype=torch.float32, device=dev)
    ops.gemm(0, 0, m, 2 * tb.nb, tb.ldf, frames, tb.ldf, tb.basis, 2 * tb.nb, spec, 2 * tb.nb)
    power = torch.empty((m, tb.nb), dtype=torch.float32, device=dev)
    _lib.check(L.asrk_power_spectrum_f32(_p(spec), _p(power), m, tb.nb, _stream()), 'power_spectrum')
    mel = torch.empty((m, num_mel_bins), dtype=torch.float32, device=dev)
    ops.gemm(0, 0, m, num_mel_bins, tb.nb, power, tb.nb, tb.melT, num_mel_bins, mel, num_mel_bins)
    _lib.check(L.asrk_log_floor_f32(_p(mel), mel.numel(), FLT_EPS, _stream()), 'log_floor')
    return mel


_MFCC_TABLES = {}


def _mfcc_table(num_mel_bins, num_ceps, cepstral_lifter, device):
    key = (num_mel_bins, num_ceps, float(cepstral_lifter), str(device))
    tab = _MFCC_TABLES.get(key)
    if tab is None:
        n = np.arange(num_mel_bins, dtype=np.float64)
        k = np.arange(num_mel_bins, dtype=np.float64)[:, None]
        dct = np.cos(np.pi / num_mel_bins * (n + 0.5) * k) * math.sqrt(2.0 / num_mel_bins)   # [k, n]
        dct[0, :] = math.sqrt(1.0 / num_mel_bins)
        mat = dct.T[:, :num_ceps].copy()                                                     # [n_mel, ceps]
        if cepstral_lifter != 0.0:
            i = np.arange(num_ceps, dtype=np.float64)
            mat *= (1.0 + 0.5 * cepstral_lifter * np.sin(np.pi * i / cepstral_lifter))[None, :]
        tab = torch.from_numpy(np.ascontiguousarray(mat, dtype=np.float32)).to(device)
        _MFCC_TABLES[key] = tab
    return tab


def kaldi_mfcc(waveform, sample_frequency, num_mel_bins=23, num_ceps=13, cepstral_lifter=22.0, **fbank_kwargs):
    """torchaudio.compliance.kaldi.mfcc restated for the device (feat_type: 'mfcc', audio.py:96): log-mel
    energies (the fbank chain above) x Kaldi's DCT-II matrix (orthonormal, first column sqrt(1/N),
    first num_ceps columns) as one more GEMM, then cepstral liftering 1 + Q/2 sin(pi i / Q) folded into
    the DCT table.  use_energy=False, subtract_mean=False (the defaults the reference relies on)."""
    if num_ceps > num_mel_bins:
        raise ValueError('num_ceps cannot be larger than num_mel_bins: %d vs %d' % (num_ceps, num_mel_bins))
    mel = kaldi_fbank(waveform, sample_frequency, num_mel_bins=num_mel_bins, **fbank_kwargs)
    m = mel.shape[0]
    tab = _mfcc_table(num_mel_bins, num_ceps, cepstral_lifter, mel.device)
    out = torch.empty((m, num_ceps), dtype=torch.float32, device=mel.device)
    if m > 0:
        ops.gemm(0, 0, m, num_ceps, num_mel_bins, mel, num_mel_bins, tab, num_ceps, out, num_ceps)
    return out


def _transpose2d(x):
    xc = _f32c(x)
    R, C = xc.shape
    y = torch.empty((C, R), dtype=torch.float32, device=x.device)
    _lib.check(_L().asrk_transpose_f32(_p(xc), _p(y), R, C, _stream()), 'transpose')
    return y


class CMVN(nn.Module):
    ''' per (channel, feature) mean/variance normalisation over time (reference: audio.py:7-30) '''

    def __init__(self, mode="global", dim=2, eps=1e-10):
        super(CMVN, self).__init__()
        if mode != "global":
            raise NotImplementedError("Only support global mean variance normalization.")
        if dim != 2:
            raise NotImplementedError("CMVN over the time axis (dim=2) only")
        self.mode, self.dim, self.eps = mode, dim, eps

    def forward(self, x):
        ops._require_gpu(x)
        xc = _f32c(x)
        C, D, T = xc.shape
        y = torch.empty_like(xc)
        _lib.check(_L().asrk_cmvn_f32(_p(xc), _p(y), C * D, T, self.eps, _stream()), 'cmvn')
        return y

    def extra_repr(self):
        return "mode={}, dim={}, eps={}".format(self.mode, self.dim, self.eps)


class Delta(nn.Module):
    ''' delta / delta-delta features (reference: audio.py:33-80) '''

    def __init__(self, order=1, window_size=2):
        super(Delta, self).__init__()
        self.order = order
        self.window_size = window_size
        filters = self._create_filters(order, window_size)
        self.register_buffer("filters", filters)        # [order+1, 1, 1, L] like the reference
        self.padding = (0, (filters.shape[-1] - 1) // 2)

    def forward(self, x):
        ops._require_gpu(x)
        xc = _f32c(x)
        assert xc.shape[0] == 1, 'Delta expects [1, D, T] (audio.py:50-54)'
        _, D, T = xc.shape
        C, Lf = self.order + 1, self.filters.shape[-1]
        filt = _f32c(self.filters.reshape(C, Lf).to(x.device))
        y = torch.empty((C, D, T), dtype=torch.float32, device=x.device)
        _lib.check(_L().asrk_delta_f32(_p(xc), _p(filt), _p(y), C, D, T, Lf, _stream()), 'delta')
        return y

    def _create_filters(self, order, window_size):
        ''' regression filters, order i built by convolving order i-1 (audio.py:57-77) '''
        scales = [[1.0]]
        for i in range(1, order + 1):
            prev_offset = (len(scales[i - 1]) - 1) // 2
            curr_offset = prev_offset + window_size
            curr = [0.0] * (len(scales[i - 1]) + 2 * window_size)
            normalizer = 0.0
            for j in range(-window_size, window_size + 1):
                normalizer += j * j
                for k in range(-prev_offset, prev_offset + 1):
                    curr[j + k + curr_offset] += (j * scales[i - 1][k + prev_offset])
            scales.append([v / normalizer for v in curr])
        max_len = len(scales[-1])
        for i, scale in enumerate(scales[:-1]):
            padding = (max_len - len(scale)) // 2
            scales[i] = [0.0] * padding + scale + [0.0] * padding
        return torch.tensor(scales, dtype=torch.float32).unsqueeze(1).unsqueeze(1)

    def extra_repr(self):
        return "order={}, window_size={}".format(self.order, self.window_size)


class Postprocess(nn.Module):
    ''' [channel, feature_dim, time] -> [time, channel * feature_dim] (reference: audio.py:83-89) '''

    def forward(self, x):
        C, D, T = x.shape
        return _transpose2d(_f32c(x).view(C * D, T)).detach()


class ExtractAudioFeature(nn.Module):
    ''' file (or waveform tensor) -> [1, feat_dim, T] (reference: audio.py:93-112) '''

    def __init__(self, mode="fbank", num_mel_bins=40, device='cuda', **kwargs):
        super(ExtractAudioFeature, self).__init__()
        if mode not in ("fbank", "mfcc"):
            raise ValueError("feat_type must be 'fbank' or 'mfcc', got {!r}".format(mode))
        self.mode = mode
        self.num_mel_bins = num_mel_bins
        self.kwargs = kwargs
        self.device = device

    def forward(self, filepath):
        if isinstance(filepath, (tuple, list)):
            waveform, sample_rate = filepath
        else:
            waveform, sample_rate = load_wav(filepath)
        waveform = waveform.to(self.device)
        extract = kaldi_fbank if self.mode == "fbank" else kaldi_mfcc
        y = extract(waveform, num_mel_bins=self.num_mel_bins, channel=-1,
                    sample_frequency=sample_rate, **self.kwargs)
        return _transpose2d(y).unsqueeze(0).detach()

    def extra_repr(self):
        return "mode={}, num_mel_bins={}".format(self.mode, self.num_mel_bins)


def audio_num_samples(filepath):
    """-> (samples per channel, sample_rate) from the file header only (wav: RIFF chunk sizes; flac: STREAMINFO) -
    what the collate function needs to order / halve / shard a batch before anything is decoded.  A FLAC stream
    without a stored sample count is decoded to find out."""
    if str(filepath).lower().endswith('.flac'):
        import ctypes
        lib = _lib.load()
        sr, nch, bps, total = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int64(0)
        md5 = (ctypes.c_uint8 * 16)()
        _lib.check(lib.asrk_flac_info(str(filepath).encode(), ctypes.byref(sr), ctypes.byref(nch), ctypes.byref(bps),
                                      ctypes.byref(total), md5), "flac_info(%s)" % filepath)
        if total.value > 0:
            return int(total.value), int(sr.value)
        x, rate = load_flac(filepath, verify_md5=False)
        return int(x.shape[1]), rate
    with _wave.open(str(filepath), 'rb') as w:
        return w.getnframes(), w.getframerate()


def load_pcm(filepath):
    """-> (int16 numpy [N] of channel 0, sample_rate): the raw 16-bit PCM the batched front end uploads (2 B per
    sample over PCIe; the device applies torchaudio.load's x / 32768).  .wav through the standard library,
    .flac through the native decoder."""
    if str(filepath).lower().endswith('.flac'):
        x, sr = load_flac(filepath)
        q = torch.round(x[0] * 32768.0).clamp_(-32768, 32767).to(torch.int16).numpy()
        if not np.array_equal(q.astype(np.float32) / 32768.0, x[0].numpy()):
            raise ValueError('%s: not 16-bit audio' % filepath)
        return q, sr
    with _wave.open(filepath, 'rb') as w:
        sr, nch, sw, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if sw != 2:
        raise ValueError('only 16-bit PCM .wav and .flac files are supported')
    return np.ascontiguousarray(np.frombuffer(raw, dtype='<i2').reshape(-1, nch)[:, 0]), sr


class BatchFeatureTransform:
    """The reference's per-file transform (create_transform: ExtractAudioFeature -> Delta -> CMVN -> Postprocess,
    src/audio.py:115-133) followed by pad_sequence (src/data.py:39) for a WHOLE BATCH in 7 launches: padded
    PCM [B, Nmax] -> frames of all utterances stacked [sum m, 400] -> one DFT GEMM -> power -> one mel GEMM ->
    log -> delta + CMVN + layout + zero padding in one kernel -> (feat [B, Tmax, (order+1)*D], feat_len [B]).
    Same arithmetic per frame as the per-file modules (the framing kernel is the same code; CMVN sums in a
    different order: agreement ~1e-6)."""

    def __init__(self, audio_config, device='cuda'):
        cfg = dict(audio_config)
        self.feat_type = cfg.pop("feat_type")
        if self.feat_type not in ("fbank", "mfcc"):
            raise ValueError("feat_type must be 'fbank' or 'mfcc', got {!r}".format(self.feat_type))
        self.feat_dim = cfg.pop("feat_dim")
        self.delta_order = cfg.pop("delta_order", 0)
        self.delta_window_size = cfg.pop("delta_window_size", 2)
        self.apply_cmvn = cfg.pop("apply_cmvn")
        self.frame_length = cfg.pop("frame_length", 25.0)
        self.frame_shift = cfg.pop("frame_shift", 10.0)
        self.preemph = cfg.pop("preemphasis_coefficient", 0.97)
        self.remove_dc = cfg.pop("remove_dc_offset", True)
        self.low_freq, self.high_freq = cfg.pop("low_freq", 20.0), cfg.pop("high_freq", 0.0)
        self.num_ceps, self.lifter = cfg.pop("num_ceps", 13), cfg.pop("cepstral_lifter", 22.0)
        if cfg.pop("dither", 0.0) != 0.0:
            raise NotImplementedError('dither > 0 is random; the shipped configs set dither: 0')
        cfg.pop("channel", None)
        if cfg:
            raise NotImplementedError('unsupported fbank options: %s' % sorted(cfg))
        self.device = torch.device(device)
        self.filters = Delta(self.delta_order, self.delta_window_size).filters if self.delta_order >= 1 else \
            torch.ones(1, 1, 1, 1)
        self.out_dim = (self.feat_dim if self.feat_type == "fbank" else self.num_ceps) * (self.delta_order + 1)
        if self.filters.shape[-1] > 16:
            raise NotImplementedError('delta filters longer than 16 taps')
        self._staging, self._staging_ev = {}, {}

    def frame_count(self, n_samples, sample_rate):
        win, shift = int(sample_rate * self.frame_length * 0.001), int(sample_rate * self.frame_shift * 0.001)
        return 0 if n_samples < win else 1 + (n_samples - win) // shift

    def __call__(self, waves, sample_rate):
        """waves: list of 1-D int16 numpy arrays (raw PCM; uploaded as 2-byte samples) or float32 arrays /
        tensors in [-1, 1) -> (feat [B, Tmax, out_dim] on the device, zero padded, feat_len LongTensor [B])"""
        L = _L()
        dev, B = self.device, len(waves)
        is_i16 = all(isinstance(w, np.ndarray) and w.dtype == np.int16 for w in waves)
        arrs = [w if is_i16 else torch.as_tensor(w, dtype=torch.float32).reshape(-1).numpy() for w in waves]
        ns = [int(a.shape[0]) for a in arrs]
        tb = _FbankTables.get(int(sample_rate), self.frame_length, self.frame_shift, self.feat_dim, self.low_freq,
                              self.high_freq, dev)
        ms = [self.frame_count(n, sample_rate) for n in ns]
        Tmax, total = max(ms), sum(ms)
        D = self.feat_dim if self.feat_type == "fbank" else self.num_ceps
        C, Lf = self.delta_order + 1, self.filters.shape[-1]
        feat_len = torch.LongTensor(ms)
        out = torch.empty((B, Tmax, C * D), dtype=torch.float32, device=dev)
        if total == 0:
            return out, feat_len
        nmax = max(ns)
        # persistent pinned staging buffer (pinning per batch costs more than the whole front end); rows are
        # NOT cleared: the framing kernel never reads beyond an utterance's own last frame
        tdt = torch.int16 if is_i16 else torch.float32
        need = B * nmax
        st = self._staging.get(tdt)
        if st is None or st.numel() < need:
            if st is not None and self._staging_ev.get(tdt) is not None:
                self._staging_ev[tdt].synchronize()
            st = torch.empty((need + need // 4,), dtype=tdt).pin_memory()
            self._staging[tdt] = st
        elif self._staging_ev.get(tdt) is not None:
            self._staging_ev[tdt].synchronize()      # the previous batch's upload has left the buffer
        host_t = st[:need].view(B, nmax)
        host = host_t.numpy()
        for b, a in enumerate(arrs):
            host[b, :ns[b]] = a
        wave = host_t.to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._staging_ev[tdt] = ev
        offs = np.zeros(B + 1, dtype=np.int64)
        offs[1:] = np.cumsum(ms)
        frame_off = torch.from_numpy(offs).to(dev)
        n_host = np.asarray(ns, dtype=np.int64)
        if tb.fused:
            mel = torch.empty((total, self.feat_dim), dtype=torch.float32, device=dev)
            _lib.check(L.asrk_fbank_logmel_batch_f32(_p(wave), 2 if is_i16 else 4, nmax, n_host.ctypes.data,
                                                     _p(frame_off), B, Tmax, _p(tb.window), _p(tb.tw_fft),
                                                     _p(tb.tw_unpack), _p(tb.melT), _p(tb.mel_range), self.feat_dim,
                                                     self.feat_dim, _p(mel), tb.win, tb.shift, tb.log2n,
                                                     1.0 / 32768.0 if is_i16 else 1.0, self.preemph,
                                                     int(self.remove_dc), FLT_EPS, _stream()), 'fbank_logmel_batch')
        else:
            frames = torch.empty((total, tb.ldf), dtype=torch.float32, device=dev)
            _lib.check(L.asrk_fbank_frames_batch_f32(_p(wave), 2 if is_i16 else 4, nmax,
                                                     n_host.ctypes.data, _p(frame_off), B, Tmax, _p(tb.window),
                                                     _p(frames), tb.win, tb.shift, tb.ldf,
                                                     1.0 / 32768.0 if is_i16 else 1.0, self.preemph,
                                                     int(self.remove_dc), _stream()), 'fbank_frames_batch')
            mel = _frames_to_logmel(frames, tb, self.feat_dim)
        if self.feat_type == "mfcc":
            tab = _mfcc_table(self.feat_dim, self.num_ceps, self.lifter, dev)
            cep = torch.empty((total, self.num_ceps), dtype=torch.float32, device=dev)
            ops.gemm(0, 0, total, self.num_ceps, self.feat_dim, mel, self.feat_dim, tab, self.num_ceps, cep,
                     self.num_ceps)
            mel = cep
        filt = _f32c(self.filters.reshape(C, Lf).to(dev))
        _lib.check(L.asrk_delta_cmvn_batch_f32(_p(mel), _p(frame_off), B, D, _p(filt), C, Lf, int(self.apply_cmvn),
                                               1e-10, _p(out), Tmax, _stream()), 'delta_cmvn_batch')
        return out, feat_len


def create_transform(audio_config, device='cuda'):
    ''' same contract as the reference (audio.py:115-133): pops feat_type / feat_dim / delta_order /
    delta_window_size / apply_cmvn, the rest are fbank kwargs; returns (Sequential, output dim) '''
    try:                                    # the whole-batch form of the same chain (collate uses it when present)
        batch_form = BatchFeatureTransform(audio_config, device=device)
    except NotImplementedError:
        batch_form = None
    feat_type = audio_config.pop("feat_type")
    feat_dim = audio_config.pop("feat_dim")

    delta_order = audio_config.pop("delta_order", 0)
    delta_window_size = audio_config.pop("delta_window_size", 2)
    apply_cmvn = audio_config.pop("apply_cmvn")

    transforms = [ExtractAudioFeature(feat_type, feat_dim, device=device, **audio_config)]
    if delta_order >= 1:
        transforms.append(Delta(delta_order, delta_window_size))
    if apply_cmvn:
        transforms.append(CMVN())
    transforms.append(Postprocess())

    seq = nn.Sequential(*transforms)
    seq.batch = batch_form                  # plain attribute: not a submodule, not part of any state_dict
    return seq, feat_dim * (delta_order + 1)
