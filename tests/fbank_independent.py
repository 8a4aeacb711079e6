"""A second, independent implementation of the Kaldi-compliant log-mel filterbank the reference calls at
src/audio.py:104-108 (torchaudio.compliance.kaldi.fbank) plus closed-form known answers.  It shares NO
code with oracle/fbank_oracle.py or the product: scipy.signal / scipy.fft primitives, per-filter loops
and literal constants computed by hand from the published Kaldi definitions

    mel(f)      = 1127 ln(1 + f / 700)
    window      = povey: hann(N, symmetric)^0.85
    frame i     = x[i*shift : i*shift + N]   (snip_edges), minus its mean, pre-emphasised with 0.97 and
                  the first sample replicated, windowed, zero-padded to 512, |rFFT|^2
    filter m    = triangle in the MEL domain between mel_lo + m d, + (m+1) d, + (m+2) d,
                  d = (mel_hi - mel_lo) / (M + 1), evaluated at the FFT bin centres k * sr / 512, k < 256
    output      = ln(max(energy, FLT_EPSILON))

so that tests can pin BOTH the oracle and the HIP pipeline on something outside this repo's own
restatement (torchaudio itself is not installable in the image).
"""
import math

import numpy as np
import scipy.fft
import scipy.signal

FLT_EPS = 1.1920928955078125e-07
LOG_FLOOR = -15.942385152878742        # ln(FLT_EPSILON), hand-computed
MEL_LOW_20HZ = 31.7485783415           # 1127 ln(1 + 20/700)
MEL_HIGH_8KHZ = 2840.0377117384        # 1127 ln(1 + 8000/700)
# centre frequencies (Hz) of filters 0, 1, M/2, M-1 for 16 kHz audio: 700 (exp((mel_lo + (m+1) d)/1127) - 1)
CENTRES_HZ = {40: {0: 65.116020, 1: 113.059061, 20: 1880.021197, 39: 7486.993653},
              80: {0: 42.493792, 1: 65.690321, 40: 1841.593199, 79: 7736.434175},
              23: {0: 98.773432, 1: 186.165271, 11: 1802.798434, 22: 7142.023471}}


def scipy_fbank(x, sr, num_mel_bins, frame_length_ms=25.0, frame_shift_ms=10.0, preemph=0.97):
    x = np.asarray(x, dtype=np.float64)
    N = int(sr * frame_length_ms / 1000.0)
    S = int(sr * frame_shift_ms / 1000.0)
    P = 1
    while P < N:
        P *= 2
    n_frames = 0 if len(x) < N else 1 + (len(x) - N) // S
    out = np.empty((n_frames, num_mel_bins))
    window = scipy.signal.get_window("hann", N, fftbins=False) ** 0.85
    # triangular filters, one python loop per filter and bin (deliberately naive)
    mel_lo = 1127.0 * math.log(1.0 + 20.0 / 700.0)
    mel_hi = 1127.0 * math.log(1.0 + (sr / 2.0) / 700.0)
    d = (mel_hi - mel_lo) / (num_mel_bins + 1)
    fb = np.zeros((num_mel_bins, P // 2 + 1))
    for m in range(num_mel_bins):
        left, centre, right = mel_lo + m * d, mel_lo + (m + 1) * d, mel_lo + (m + 2) * d
        for k in range(P // 2):                      # the Nyquist bin carries no weight
            mk = 1127.0 * math.log(1.0 + (k * sr / P) / 700.0)
            if left < mk < right:
                fb[m, k] = (mk - left) / (centre - left) if mk <= centre else (right - mk) / (right - centre)
    for i in range(n_frames):
        fr = x[i * S:i * S + N].copy()
        fr -= fr.sum() / N
        # y[n] = fr[n] - 0.97 fr[n-1] with fr[-1] := fr[0]: an FIR filter whose state starts at fr[0]
        zi = scipy.signal.lfiltic([1.0, -preemph], [1.0], y=[], x=[fr[0]])
        fr, _ = scipy.signal.lfilter([1.0, -preemph], [1.0], fr, zi=zi)
        spec = scipy.fft.rfft(fr * window, n=P)
        out[i] = np.log(np.maximum(fb @ (spec.real ** 2 + spec.imag ** 2), FLT_EPS))
    return out


def tone(freq_hz, n, sr, amp=0.25):
    return amp * np.cos(2.0 * math.pi * freq_hz * np.arange(n) / sr)


def tone_frame_energy(freq_hz, sr, amp=0.25, N=400, preemph=0.97):
    """Closed form for the total in-band energy of ONE frame of a pure tone whose period divides the
    frame (zero mean): by Parseval the one-sided power sum of the 512-point transform of the windowed,
    pre-emphasised frame is 512/2 * sum_n (g[n] w[n])^2 (the k = 0 and k = 256 terms are negligible for a
    mid-band tone), and the mel triangles - linear in mel, adjacent ones overlapping - sum to exactly 1
    on every FFT bin between the first and the last centre.  So sum_m exp(logmel[m]) must equal it."""
    n = np.arange(N)
    f = amp * np.cos(2.0 * math.pi * freq_hz * n / sr)
    g = f - preemph * np.concatenate([f[:1], f[:-1]])
    w = (0.5 - 0.5 * np.cos(2.0 * math.pi * n / (N - 1))) ** 0.85
    return 256.0 * float(np.sum((g * w) ** 2))
