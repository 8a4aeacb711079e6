"""CPU: the native FLAC decoder (csrc/flac.cpp, what replaces torchaudio.load on LibriSpeech's .flac
files - src/audio.py:102) against an independent ENCODER written here from the format specification
(RFC 9639 / xiph.org/flac/format.html): every subframe type, Rice and Rice2 partitions with escape
codes, wasted bits, all stereo decorrelation modes, fixed and explicit block sizes, CRC and MD5 checks."""
import hashlib
import importlib
import struct

import numpy as np
import pytest

from conftest import PKG_NAME


class BitWriter:
    def __init__(self):
        self.bits = []

    def write(self, value, n):
        for i in range(n - 1, -1, -1):
            self.bits.append((value >> i) & 1)

    def write_signed(self, value, n):
        self.write(value & ((1 << n) - 1), n)

    def unary(self, q):
        self.bits.extend([0] * q + [1])

    def rice(self, v, k):
        u = (v << 1) if v >= 0 else ((-v << 1) - 1)
        self.unary(u >> k)
        if k:
            self.write(u & ((1 << k) - 1), k)

    def pad(self):
        while len(self.bits) % 8:
            self.bits.append(0)

    def to_bytes(self):
        assert len(self.bits) % 8 == 0
        out = bytearray()
        for i in range(0, len(self.bits), 8):
            b = 0
            for bit in self.bits[i:i + 8]:
                b = (b << 1) | bit
            out.append(b)
        return bytes(out)


def crc(data, poly, width):
    c, top = 0, 1 << (width - 1)
    for byte in data:
        c ^= byte << (width - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) if c & top else (c << 1)
            c &= (1 << width) - 1
    return c


def residual(w, res, blocksize, order, porder, method, escape_part=None):
    w.write(method, 2)
    w.write(porder, 4)
    pbits = 4 if method == 0 else 5
    n_parts, pos = 1 << porder, 0
    for part in range(n_parts):
        cnt = (blocksize >> porder) - (order if part == 0 else 0) if porder else blocksize - order
        chunk = res[pos:pos + cnt]
        pos += cnt
        if escape_part == part:
            w.write((1 << pbits) - 1, pbits)
            nb = max(1, max((int(abs(v)).bit_length() + 1 for v in chunk), default=1))
            w.write(nb, 5)
            for v in chunk:
                w.write_signed(int(v), nb)
        else:
            mean = np.mean(np.abs(chunk)) if len(chunk) else 0
            k = min((1 << pbits) - 2, max(0, int(np.log2(mean + 1))))
            w.write(k, pbits)
            for v in chunk:
                w.rice(int(v), k)


FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def subframe(w, x, bps, kind, **kw):
    x = [int(v) for v in x]
    n = len(x)
    wasted = kw.get("wasted", 0)
    w.write(0, 1)
    if kind == "constant":
        w.write(0, 6)
    elif kind == "verbatim":
        w.write(1, 6)
    elif kind == "fixed":
        w.write(8 + kw["order"], 6)
    else:
        w.write(32 + kw["order"] - 1, 6)
    if wasted:
        w.write(1, 1)
        w.unary(wasted - 1)
        x = [v >> wasted for v in x]
        bps -= wasted
    else:
        w.write(0, 1)
    if kind == "constant":
        w.write_signed(x[0], bps)
    elif kind == "verbatim":
        for v in x:
            w.write_signed(v, bps)
    else:
        order = kw["order"]
        for v in x[:order]:
            w.write_signed(v, bps)
        if kind == "fixed":
            coefs, shift = FIXED[order], 0
        else:
            coefs, shift, prec = kw["coefs"], kw["shift"], kw["prec"]
            w.write(prec - 1, 4)
            w.write_signed(shift, 5)
            for c in coefs:
                w.write_signed(c, prec)
        res = []
        for i in range(order, n):
            pred = sum(c * x[i - 1 - j] for j, c in enumerate(coefs)) >> shift
            res.append(x[i] - pred)
        residual(w, res, n, order, kw.get("porder", 0), kw.get("method", 0), kw.get("escape_part"))


def frame(index, channels, bps, sr_code, ch_code, specs, blocksize_code=None):
    """channels: list of per-channel int arrays as they are CODED (after stereo decorrelation)"""
    n = len(channels[0])
    w = BitWriter()
    w.write(0x3FFE, 14); w.write(0, 1); w.write(0, 1)           # sync, reserved, fixed block size
    codes = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12}
    bs_code = blocksize_code if blocksize_code is not None else codes.get(n, 6 if n <= 256 else 7)
    w.write(bs_code, 4); w.write(sr_code, 4); w.write(ch_code, 4)
    w.write({8: 1, 12: 2, 16: 4, 20: 5, 24: 6}[bps], 3); w.write(0, 1)
    assert index < 128
    w.write(index, 8)                                           # UTF-8 coded frame number (1 byte)
    if bs_code == 6:
        w.write(n - 1, 8)
    elif bs_code == 7:
        w.write(n - 1, 16)
    hdr = w.to_bytes()
    w.write(crc(hdr, 0x07, 8), 8)
    for c, (x, spec) in enumerate(zip(channels, specs)):
        side = (ch_code == 8 and c == 1) or (ch_code == 9 and c == 0) or (ch_code == 10 and c == 1)
        subframe(w, x, bps + (1 if side else 0), **spec)
    w.pad()
    body = w.to_bytes()
    return body + struct.pack(">H", crc(body, 0x8005, 16))


def make_flac(path, pcm, bps, sr, frames, with_md5=True):
    """pcm: int array [n, channels]; frames: list of (start, length, ch_code, specs)"""
    n, nch = pcm.shape
    nbytes = (bps + 7) // 8
    raw = b"".join(int(v).to_bytes(nbytes, "little", signed=True) for v in pcm.reshape(-1))
    md5 = hashlib.md5(raw).digest() if with_md5 else bytes(16)
    si = BitWriter()
    maxb = max(f[1] for f in frames)
    si.write(maxb, 16); si.write(maxb, 16); si.write(0, 24); si.write(0, 24)
    si.write(sr, 20); si.write(nch - 1, 3); si.write(bps - 1, 5); si.write(n, 36)
    body = b"fLaC" + bytes([0x80 | 0]) + (34).to_bytes(3, "big") + si.to_bytes() + md5
    sr_code = {8000: 4, 16000: 5, 22050: 6, 44100: 9, 48000: 10}[sr]
    for idx, (start, length, ch_code, specs) in enumerate(frames):
        blk = pcm[start:start + length].astype(np.int64)
        if ch_code < 8:
            chans = [blk[:, c] for c in range(nch)]
        elif ch_code == 8:
            chans = [blk[:, 0], blk[:, 0] - blk[:, 1]]
        elif ch_code == 9:
            chans = [blk[:, 0] - blk[:, 1], blk[:, 1]]
        else:
            chans = [(blk[:, 0] + blk[:, 1]) >> 1, blk[:, 0] - blk[:, 1]]
        body += frame(idx, chans, bps, sr_code, ch_code, specs)
    open(path, "wb").write(body)


def _signal(n, nch, bps, seed):
    rng = np.random.RandomState(seed)
    t = np.arange(n)
    amp = (1 << (bps - 1)) * 0.3
    x = np.stack([amp * np.sin(2 * np.pi * (0.01 + 0.003 * c) * t + c) + amp * 0.02 * rng.randn(n) for c in range(nch)], 1)
    return np.round(x).astype(np.int64)


@pytest.fixture(scope="module")
def audio():
    importlib.import_module(PKG_NAME + ".build").build(verbose=False)
    return importlib.import_module(PKG_NAME + ".src.audio")


def test_flac_mono_16bit_all_subframe_types(audio, tmp_path):
    n = 4096 + 1152 + 256 + 100 + 17
    pcm = _signal(n, 1, 16, 1)
    pcm[4096:4096 + 1152] = -1234                                  # a constant block
    pcm[4096 + 1152:4096 + 1152 + 256] &= ~3                       # two wasted bits
    lpc = dict(kind="lpc", order=3, coefs=[1900, -1100, 210], shift=10, prec=12, porder=2, method=1, escape_part=1)
    frames = [(0, 4096, 0, [dict(kind="fixed", order=2, porder=4, method=0)]),
              (4096, 1152, 0, [dict(kind="constant")]),
              (5248, 256, 0, [dict(kind="fixed", order=4, porder=1, wasted=2)]),
              (5504, 100, 0, [lpc]),                               # explicit 8-bit block size
              (5604, 17, 0, [dict(kind="verbatim")])]
    path = tmp_path / "a.flac"
    make_flac(path, pcm, 16, 16000, frames)
    x, sr = audio.load_wav(str(path))
    assert sr == 16000 and tuple(x.shape) == (1, n)
    assert np.array_equal(np.round(x.numpy()[0] * 32768).astype(np.int64), pcm[:, 0])


@pytest.mark.parametrize("ch_code", [1, 8, 9, 10])
def test_flac_stereo_decorrelation_modes(audio, tmp_path, ch_code):
    n = 576 + 300
    pcm = _signal(n, 2, 16, 2 + ch_code)
    spec = [dict(kind="fixed", order=1, porder=0), dict(kind="lpc", order=2, coefs=[30, -14], shift=4, prec=7, porder=2)]
    frames = [(0, 576, ch_code, spec), (576, 300, ch_code, [dict(kind="fixed", order=3), dict(kind="verbatim")])]
    path = tmp_path / "s.flac"
    make_flac(path, pcm, 16, 44100, frames)
    x, sr = audio.load_wav(str(path))
    assert sr == 44100 and tuple(x.shape) == (2, n)
    assert np.array_equal(np.round(x.numpy().T * 32768).astype(np.int64), pcm)


def test_flac_24bit_and_corruption_is_detected(audio, tmp_path):
    n = 512
    pcm = _signal(n, 1, 24, 7)
    path = tmp_path / "h.flac"
    make_flac(path, pcm, 24, 48000, [(0, 512, 0, [dict(kind="fixed", order=2, porder=3, method=1)])])
    x, sr = audio.load_wav(str(path))
    assert sr == 48000 and np.array_equal(np.round(x.numpy()[0].astype(np.float64) * (1 << 23)).astype(np.int64), pcm[:, 0])
    data = bytearray(open(path, "rb").read())
    data[-40] ^= 0x10                                              # flip one bit inside the frame
    bad = tmp_path / "bad.flac"
    open(bad, "wb").write(bytes(data))
    with pytest.raises(Exception):
        audio.load_wav(str(bad))
    # a wrong STREAMINFO MD5 is caught even when every frame CRC is fine
    data = bytearray(open(path, "rb").read())
    data[8 + 18] ^= 0xFF
    open(bad, "wb").write(bytes(data))
    with pytest.raises(ValueError):
        audio.load_wav(str(bad))


def test_flac_decode_reports_required_capacity_instead_of_truncating(audio, tmp_path):
    """a stream that holds more samples than the caller's buffer (possible when STREAMINFO carries no sample count
    and the size was guessed, e.g. long CONSTANT / silent stretches): ASRK_EWORKSPACE + the count needed, and the
    loader retries - never a silently shortened waveform"""
    import ctypes
    n = 4096 * 3
    pcm = _signal(n, 1, 16, 5)
    pcm[:] = 77                                                    # constant blocks: a few bytes per 4096 samples
    frames = [(i * 4096, 4096, 0, [dict(kind="constant")]) for i in range(3)]
    path = tmp_path / "c.flac"
    make_flac(path, pcm, 16, 16000, frames)
    lib = importlib.import_module(PKG_NAME + "._lib").load()
    buf = np.empty((100, 1), dtype=np.int32)
    got = ctypes.c_int64(0)
    rc = lib.asrk_flac_decode_i32(str(path).encode(), buf.ctypes.data_as(ctypes.c_void_p), 100, ctypes.byref(got))
    assert rc == -3 and got.value == n and np.all(buf == 77)
    # the loader's own guess (file size * 8) is far below n when total_samples is unknown: patch it to 0
    raw = bytearray(open(path, "rb").read())
    # STREAMINFO: 4 (magic) + 4 (block header) + 10 bytes in: 36-bit total_samples field -> zero it
    raw[4 + 4 + 13] &= 0xF0
    raw[4 + 4 + 14:4 + 4 + 18] = b"\x00\x00\x00\x00"
    p2 = tmp_path / "c0.flac"
    open(p2, "wb").write(bytes(raw))
    x, sr = audio.load_flac(str(p2), verify_md5=True)
    assert tuple(x.shape) == (1, n) and np.all(np.round(x.numpy() * 32768) == 77)
