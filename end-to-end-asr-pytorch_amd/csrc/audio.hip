// Audio front end on the device (reference: src/audio.py:7-133; the fbank itself is
// torchaudio.compliance.kaldi.fbank, third-party — algorithm restated in oracle/fbank_oracle.py).
//
//   frames   : snip_edges framing -> per-frame DC removal -> pre-emphasis (replicate pad) -> povey
//              window                                  (one wave per frame, coalesced sample rows)
//   spectrum : the 400-sample frames times a precomputed [win, 2*NB] cos|sin basis with the exact-
//              f32 MFMA GEMM (asrk_gemm_f32) — a 512-point real DFT as a dense contraction is
//              0.4 MFLOP/frame, i.e. 21 GFLOP for a whole cfg3 batch: free on the matrix cores and
//              without an LDS butterfly network; then |X|^2                     (power kernel)
//   mel      : power [m, NB] x melT [NB, nmel] (GEMM) -> log(max(., eps))       (log kernel)
//   delta    : 9-tap / 5-tap cross-correlation along time with ZERO padding (src/audio.py:48-54)
//   cmvn     : per (channel, feature) row over time, unbiased std, eps added to std (audio.py:24-27)
// All of these are HBM-streaming kernels apart from the two GEMMs.
#include "common.h"

namespace {

// one wave per frame; frames: [m, ldf] (ldf >= win, multiple of 4; pad columns zeroed)
__global__ __launch_bounds__(256) void fbank_frames_kernel(const float *__restrict__ wavef,
                                                           const float *__restrict__ window,
                                                           float *__restrict__ frames, int64_t n_samples,
                                                           int m, int win, int shift, int ldf,
                                                           float preemph, int remove_dc) {
    const int f = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (f >= m) return;
    const float *x = wavef + (int64_t)f * shift;
    float s = 0.f;
    for (int j = lane; j < win; j += 64) s += x[j];
    const float mean = remove_dc ? wave_sum(s) / (float)win : 0.f;
    float *o = frames + (size_t)f * ldf;
    for (int j = lane; j < ldf; j += 64) {
        float v = 0.f;
        if (j < win) {
            const float cur = x[j] - mean;
            const float prev = x[j > 0 ? j - 1 : 0] - mean;   // replicate-padded first sample
            v = (cur - preemph * prev) * window[j];
        }
        o[j] = v;
    }
}

// The same framing for a whole padded batch: wave [B, ld_wave] (int16 PCM, scaled by `scale` on load - the
// exact torchaudio.load conversion x / 32768 - or f32), frame f of utterance b goes to row frame_off[b] + f.
// grid (ceil(max_m / 4), B); the per-frame arithmetic is the per-utterance kernel's, operation for operation.
template <typename SampleT>
__global__ __launch_bounds__(256) void fbank_frames_batch_kernel(const SampleT *__restrict__ wavef, int64_t ld_wave,
                                                                 const int64_t *__restrict__ frame_off,
                                                                 const float *__restrict__ window,
                                                                 float *__restrict__ frames, int win, int shift,
                                                                 int ldf, float scale, float preemph, int remove_dc) {
    const int b = blockIdx.y;
    const int f = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t off = frame_off[b];
    if (f >= (int)(frame_off[b + 1] - off)) return;
    const SampleT *x = wavef + (int64_t)b * ld_wave + (int64_t)f * shift;
    float s = 0.f;
    for (int j = lane; j < win; j += 64) s += (float)x[j] * scale;
    const float mean = remove_dc ? wave_sum(s) / (float)win : 0.f;
    float *o = frames + (size_t)(off + f) * ldf;
    for (int j = lane; j < ldf; j += 64) {
        float v = 0.f;
        if (j < win) {
            const float cur = (float)x[j] * scale - mean;
            const float prev = (float)x[j > 0 ? j - 1 : 0] * scale - mean;
            v = (cur - preemph * prev) * window[j];
        }
        o[j] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Fused log-mel filterbank (round 6): PCM -> log-mel energies in ONE kernel, one wave per frame.
//
// Rounds 3-5 ran the 512-point DFT of every frame as a dense [frames x 512] x [512 x 514] f32 GEMM against a cos|sin
// basis (26.9 GFLOP per 32 x 1600-frame batch) between a framing kernel, a power kernel, a second GEMM against the
// dense mel matrix and a log kernel: 263 MB of operands written and re-read for 33 MB of algorithmic bytes (0.4 % of
// the HBM roofline).  Here a wave
//   1. reads the frame's `win` samples straight from the padded PCM batch (the framing arithmetic of
//      fbank_frames_batch_kernel, operation for operation: DC removal, pre-emphasis, povey window), packs the
//      zero-padded real sequence v[0..N) as N/2 complex points z[m] = v[2m] + i v[2m+1] in its LDS tile,
//   2. runs an N/2-point complex FFT in LDS (Stockham autosort, radix 4 with a radix-2 stage when log2(N/2) is odd;
//      twiddles from a table computed in float64 on the host),
//   3. unpacks the real spectrum X[k], k = 0..N/2, forms |X[k]|^2,
//   4. applies the triangular mel weights (bin m only touches its own [first, last) range of FFT bins; the weights
//      come from the same dense table the GEMM used) and stores log(max(E, eps)).
// The only global traffic is 2 B (int16) or 4 B per sample in and 4 B per mel energy out.
template <typename SampleT, int LOG2N>
__global__ __launch_bounds__(256) void fbank_logmel_batch_kernel(
    const SampleT *__restrict__ wavef, int64_t ld_wave, const int64_t *__restrict__ frame_off,
    const float *__restrict__ window, const float *__restrict__ tw_fft, const float *__restrict__ tw_unpack,
    const float *__restrict__ melT, const int *__restrict__ mel_range, int nmel, int ld_mel, float *__restrict__ out,
    int win, int shift, float scale, float preemph, int remove_dc, float eps) {
    constexpr int N = 1 << LOG2N, M = N / 2, LOG2M = LOG2N - 1;     // M complex points
    constexpr int PER = M / 64;                                      // complex points per lane (M >= 64)
    __shared__ float2 s_z[4][2][M + 1];                              // per wave: ping-pong tiles (+1: X[M] / |X|^2)
    __shared__ float2 s_tw[M];                                       // e^{-2 pi i k / M}
    const int b = blockIdx.y, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int f = blockIdx.x * 4 + wave;
    for (int i = threadIdx.x; i < M; i += 256) s_tw[i] = make_float2(tw_fft[2 * i], tw_fft[2 * i + 1]);
    __syncthreads();
    const int64_t off = frame_off[b];
    if (f >= (int)(frame_off[b + 1] - off)) return;
    const SampleT *x = wavef + (int64_t)b * ld_wave + (int64_t)f * shift;
    float2 *za = s_z[wave][0], *zb = s_z[wave][1];
    // ---- 1. framing (fbank_frames_batch_kernel's arithmetic)
    float sm = 0.f;
    for (int j = lane; j < win; j += 64) sm += (float)x[j] * scale;
    const float mean = remove_dc ? wave_sum(sm) / (float)win : 0.f;
    for (int m = lane; m < M; m += 64) {
        float v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = 2 * m + h;
            float r = 0.f;
            if (j < win) {
                const float cur = (float)x[j] * scale - mean;
                const float prev = (float)x[j > 0 ? j - 1 : 0] * scale - mean;
                r = (cur - preemph * prev) * window[j];
            }
            v[h] = r;
        }
        za[m] = make_float2(v[0], v[1]);
    }
    __builtin_amdgcn_wave_barrier();
    // ---- 2. M-point complex FFT, Stockham autosort: stage with sub-transform length Ns reads src[j + t * M / R],
    // multiplies by e^{-2 pi i t k / (R Ns)} (k = j mod Ns) and writes dst[(j - k) * R + k + t * Ns]
    auto cmul = [](float2 a, float2 w) { return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); };
    float2 *src = za, *dst = zb;
    int Ns = 1;
    if (LOG2M & 1) {   // one radix-2 stage first
        for (int j = lane; j < M / 2; j += 64) {
            const float2 a0 = src[j], a1 = src[j + M / 2];      // Ns = 1: twiddle 1
            dst[2 * j] = make_float2(a0.x + a1.x, a0.y + a1.y);
            dst[2 * j + 1] = make_float2(a0.x - a1.x, a0.y - a1.y);
        }
        __builtin_amdgcn_wave_barrier();
        float2 *t = src; src = dst; dst = t;
        Ns = 2;
    }
#pragma unroll
    for (; Ns < M; Ns *= 4) {
        for (int j = lane; j < M / 4; j += 64) {
            const int k = j & (Ns - 1);
            const int ti = k * (M / (4 * Ns));                   // index of e^{-2 pi i k / (4 Ns)} in the M-table
            const float2 a0 = src[j];
            const float2 a1 = cmul(src[j + M / 4], s_tw[ti]);
            const float2 a2 = cmul(src[j + 2 * (M / 4)], s_tw[2 * ti]);
            const float2 a3 = cmul(src[j + 3 * (M / 4)], s_tw[3 * ti]);
            const float2 s02 = make_float2(a0.x + a2.x, a0.y + a2.y), d02 = make_float2(a0.x - a2.x, a0.y - a2.y);
            const float2 s13 = make_float2(a1.x + a3.x, a1.y + a3.y), d13 = make_float2(a1.x - a3.x, a1.y - a3.y);
            const int o = (j - k) * 4 + k;
            dst[o] = make_float2(s02.x + s13.x, s02.y + s13.y);
            dst[o + Ns] = make_float2(d02.x + d13.y, d02.y - d13.x);          // d02 - i d13
            dst[o + 2 * Ns] = make_float2(s02.x - s13.x, s02.y - s13.y);
            dst[o + 3 * Ns] = make_float2(d02.x - d13.y, d02.y + d13.x);      // d02 + i d13
        }
        __builtin_amdgcn_wave_barrier();
        float2 *t = src; src = dst; dst = t;
    }
    // ---- 3. real spectrum from the packed transform: X[k] = (Z[k] + conj Z[M-k]) / 2 - i/2 e^{-2 pi i k / N} (Z[k] - conj Z[M-k])
    float *pw = reinterpret_cast<float *>(dst);                  // |X[k]|^2, k = 0..M
    for (int k = lane; k <= M; k += 64) {
        const float2 zk = src[k & (M - 1)], zm = src[(M - k) & (M - 1)];
        const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
        const float2 o = make_float2(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));   // (Z[k] - conj Z[M-k]) / 2
        const float2 w = make_float2(tw_unpack[2 * k], tw_unpack[2 * k + 1]);       // e^{-2 pi i k / N}
        // -i * w * o
        const float2 wo = cmul(o, w);
        const float re = e.x + wo.y, im = e.y - wo.x;
        pw[k] = re * re + im * im;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- 4. mel weights + log
    for (int m = lane; m < nmel; m += 64) {
        const int k0 = mel_range[2 * m], k1 = mel_range[2 * m + 1];
        float acc = 0.f;
        for (int k = k0; k < k1; ++k) acc += pw[k] * melT[(size_t)k * ld_mel + m];
        out[(size_t)(off + f) * nmel + m] = logf(fmaxf(acc, eps));
    }
}

// Delta (src/audio.py:33-80) -> CMVN (src/audio.py:7-30) -> Postprocess (src/audio.py:83-89) -> zero padding
// of the batch (pad_sequence, src/data.py:39) in ONE pass over a batch of frame-major features
// mel [total frames, D]: workgroup = (utterance, channel c, block of 64 features), lane = feature, the four
// waves stride over time.  y[c,d,t] = sum_j filt[c,j] mel[t + j - half, d] (zero outside the utterance) is
// recomputed from L2 in each of the three sweeps (mean, unbiased variance, normalised store) - 9 taps of a
// 16-MB tensor - instead of being materialised as [C, D, T] and transposed back.
// (round 6: 16 waves stride over time and every wave keeps 8 frames' loads in flight - clamped addresses, masked values,
// no predicated loads - instead of 4 waves with one dependent round trip per frame: 746 -> ~60 us for 32 x 1600 x 80.)
constexpr int DC_WAVES = 16, DC_UNR = 8;
// v where c holds, +0 elsewhere, without a select on a loaded value (the compiler would sink the load into a branch)
__device__ __forceinline__ float dc_mask(float v, bool c) { return __uint_as_float(__float_as_uint(v) & (0u - (unsigned)c)); }
template <int LT>
__global__ __launch_bounds__(64 * DC_WAVES) void delta_cmvn_batch_kernel(const float *__restrict__ mel,
                                                                        const int64_t *__restrict__ frame_off, int D,
                                                                        const float *__restrict__ filt, int C, int L,
                                                                        int apply_cmvn, float eps,
                                                                        float *__restrict__ out, int Tmax) {
    __shared__ float red[DC_WAVES][64];
    const int b = blockIdx.z, c = blockIdx.y, d = blockIdx.x * 64 + (threadIdx.x & 63);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t off = frame_off[b];
    const int m = (int)(frame_off[b + 1] - off), half = (L - 1) / 2;
    const bool live = d < D;
    const float *x = mel + (size_t)off * D + (live ? d : 0);
    float fl[LT];
#pragma unroll
    for (int j = 0; j < LT; ++j) fl[j] = j < L ? filt[c * L + j] : 0.f;
    // y[c,d,t] for t = t0, t0 + DC_WAVES, ...: DC_UNR frames x LT taps of independent loads
    auto y8 = [&](int t0, float (&y)[DC_UNR]) {
        float v[DC_UNR][LT];
#pragma unroll
        for (int u = 0; u < DC_UNR; ++u)
#pragma unroll
            for (int j = 0; j < LT; ++j) {
                const int tt = t0 + u * DC_WAVES + j - half;
                v[u][j] = x[(size_t)min(max(tt, 0), max(m - 1, 0)) * D];
            }
#pragma unroll
        for (int u = 0; u < DC_UNR; ++u) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < LT; ++j) {
                const int tt = t0 + u * DC_WAVES + j - half;
                acc += fl[j] * dc_mask(v[u][j], j < L && tt >= 0 && tt < m);
            }
            y[u] = acc;
        }
    };
    float mean = 0.f, inv = 1.f;
    if (m <= 0) {   // an utterance without a frame: only padding to write (and nothing behind `x` to read)
        if (live)
            for (int t = wave; t < Tmax; t += DC_WAVES)
                out[(size_t)b * Tmax * C * D + (size_t)t * C * D + (size_t)c * D + d] = 0.f;
        return;
    }
    if (apply_cmvn) {
        float s = 0.f;
        for (int t0 = wave; t0 < m; t0 += DC_WAVES * DC_UNR) {
            float y[DC_UNR];
            y8(t0, y);
#pragma unroll
            for (int u = 0; u < DC_UNR; ++u) s += (t0 + u * DC_WAVES < m) ? y[u] : 0.f;
        }
        red[wave][lane] = s;
        __syncthreads();
        s = 0.f;
#pragma unroll
        for (int w = 0; w < DC_WAVES; ++w) s += red[w][lane];
        mean = s / (float)m;
        __syncthreads();
        float v = 0.f;
        for (int t0 = wave; t0 < m; t0 += DC_WAVES * DC_UNR) {
            float y[DC_UNR];
            y8(t0, y);
#pragma unroll
            for (int u = 0; u < DC_UNR; ++u) {
                const float q = y[u] - mean;
                v += (t0 + u * DC_WAVES < m) ? q * q : 0.f;
            }
        }
        red[wave][lane] = v;
        __syncthreads();
        v = 0.f;
#pragma unroll
        for (int w = 0; w < DC_WAVES; ++w) v += red[w][lane];
        const float var = v / (float)(m - 1);
        inv = 1.f / (eps + sqrtf(var));                       // m == 1 -> nan, as torch.std
    }
    if (!live) return;
    float *o = out + (size_t)b * Tmax * C * D + (size_t)c * D + d;
    for (int t0 = wave; t0 < Tmax; t0 += DC_WAVES * DC_UNR) {
        float y[DC_UNR];
        y8(t0, y);
#pragma unroll
        for (int u = 0; u < DC_UNR; ++u) {
            const int t = t0 + u * DC_WAVES;
            if (t < Tmax) o[(size_t)t * C * D] = t < m ? (y[u] - mean) * inv : 0.f;
        }
    }
}

// spec [m, 2*nb] = [re | im] -> power [m, nb]
__global__ void power_kernel(const float *__restrict__ spec, float *__restrict__ power, int64_t m,
                             int nb) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * nb) return;
    const int64_t r = i / nb;
    const int c = (int)(i - r * nb);
    const float re = spec[r * 2 * nb + c], im = spec[r * 2 * nb + nb + c];
    power[i] = re * re + im * im;
}

__global__ void log_floor_kernel(float *__restrict__ x, int64_t n, float eps) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = logf(fmaxf(x[i], eps));
}

// x [D, T] -> y [C, D, T]: y[c,d,t] = sum_j filt[c,j] * x[d, t + j - half]   (zero padded)
__global__ void delta_kernel(const float *__restrict__ x, const float *__restrict__ filt,
                             float *__restrict__ y, int C, int D, int T, int L) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)C * D * T) return;
    const int t = (int)(i % T);
    const int d = (int)((i / T) % D);
    const int c = (int)(i / ((int64_t)T * D));
    const int half = (L - 1) / 2;
    const float *xr = x + (size_t)d * T;
    float acc = 0.f;
    for (int j = 0; j < L; ++j) {
        const int tt = t + j - half;
        if (tt >= 0 && tt < T) acc += filt[c * L + j] * xr[tt];
    }
    y[i] = acc;
}

// rows [R, T]: y = (x - mean) / (eps + std_unbiased); one wave per row
__global__ __launch_bounds__(256) void cmvn_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                   int R, int T, float eps) {
    const int r = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= R) return;
    const float *xr = x + (size_t)r * T;
    float s = 0.f;
    for (int t = lane; t < T; t += 64) s += xr[t];
    const float mean = wave_sum(s) / (float)T;
    float v = 0.f;
    for (int t = lane; t < T; t += 64) {
        const float d = xr[t] - mean;
        v += d * d;
    }
    const float var = wave_sum(v) / (float)(T - 1);   // T == 1 -> nan, as torch.std
    const float inv = 1.f / (eps + sqrtf(var));
    float *yr = y + (size_t)r * T;
    for (int t = lane; t < T; t += 64) yr[t] = (xr[t] - mean) * inv;
}

// [R, Ccols] -> [Ccols, R] tiled transpose (Postprocess / [m,D] <-> [D,m])
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ x,
                                                        float *__restrict__ y, int R, int Cc) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int r = by + k, c = bx + tx;
        tile[k][tx] = (r < R && c < Cc) ? x[(size_t)r * Cc + c] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = bx + k, r = by + tx;
        if (c < Cc && r < R) y[(size_t)c * R + r] = tile[tx][k];
    }
}

inline unsigned nblk(int64_t n, int bs) { return (unsigned)asrk_div_up64(n, bs); }

}  // namespace

extern "C" int asrk_fbank_frames_f32(const float *wave, int64_t n_samples, const float *window,
                                     float *frames, int m, int win, int shift, int ldf,
                                     float preemph, int remove_dc, void *stream) {
    if (m < 0 || win <= 0 || shift <= 0 || ldf < win) return ASRK_EINVAL;
    if (m == 0) return ASRK_OK;
    if (!wave || !window || !frames) return ASRK_EINVAL;
    if ((int64_t)(m - 1) * shift + win > n_samples) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_FBANK, s);
    hipLaunchKernelGGL(fbank_frames_kernel, dim3(asrk_div_up(m, 4)), dim3(256), 0, s, wave, window,
                       frames, n_samples, m, win, shift, ldf, preemph, remove_dc);
    asrk_prof_end_(PROF_FBANK, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_power_spectrum_f32(const float *spec, float *power, int64_t m, int nb,
                                       void *stream) {
    if (m < 0 || nb <= 0) return ASRK_EINVAL;
    if (m == 0) return ASRK_OK;
    if (!spec || !power) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(power_kernel, dim3(nblk(m * nb, 256)), dim3(256), 0, s, spec, power, m, nb);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_log_floor_f32(float *x, int64_t n, float eps, void *stream) {
    if (n < 0) return ASRK_EINVAL;
    if (n == 0) return ASRK_OK;
    if (!x) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(log_floor_kernel, dim3(nblk(n, 256)), dim3(256), 0, s, x, n, eps);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_delta_f32(const float *x, const float *filters, float *y, int C, int D, int T,
                              int L, void *stream) {
    if (C <= 0 || D <= 0 || T < 0 || L <= 0 || (L & 1) == 0) return ASRK_EINVAL;
    if (T == 0) return ASRK_OK;
    if (!x || !filters || !y) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(delta_kernel, dim3(nblk((int64_t)C * D * T, 256)), dim3(256), 0, s, x, filters, y,
                       C, D, T, L);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_cmvn_f32(const float *x, float *y, int rows, int T, float eps, void *stream) {
    if (rows < 0 || T <= 0) return ASRK_EINVAL;
    if (rows == 0) return ASRK_OK;
    if (!x || !y) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cmvn_kernel, dim3(asrk_div_up(rows, 4)), dim3(256), 0, s, x, y, rows, T, eps);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_transpose_f32(const float *x, float *y, int rows, int cols, void *stream) {
    if (rows < 0 || cols < 0) return ASRK_EINVAL;
    if (rows == 0 || cols == 0) return ASRK_OK;
    if (!x || !y) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(transpose_kernel, dim3(asrk_div_up(cols, 32), asrk_div_up(rows, 32)), dim3(256), 0,
                       s, x, y, rows, cols);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_fbank_frames_batch_f32(const void *wave, int sample_bytes, int64_t ld_wave,
                                           const int64_t *n_samples_host, const int64_t *frame_off, int B,
                                           int max_m, const float *window, float *frames, int win, int shift,
                                           int ldf, float scale, float preemph, int remove_dc, void *stream) {
    if (B < 0 || max_m < 0 || win <= 0 || shift <= 0 || ldf < win || (sample_bytes != 2 && sample_bytes != 4))
        return ASRK_EINVAL;
    if (B == 0 || max_m == 0) return ASRK_OK;
    if (!wave || !frame_off || !window || !frames) return ASRK_EINVAL;
    // the frames of every utterance must lie inside its row of the padded batch
    if ((int64_t)(max_m - 1) * shift + win > ld_wave) return ASRK_EINVAL;
    if (n_samples_host)
        for (int b = 0; b < B; ++b)
            if (n_samples_host[b] > ld_wave) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_FBANK, s);
    const dim3 grid(asrk_div_up(max_m, 4), B);
    if (sample_bytes == 2)
        hipLaunchKernelGGL(fbank_frames_batch_kernel<int16_t>, grid, dim3(256), 0, s,
                           reinterpret_cast<const int16_t *>(wave), ld_wave, frame_off, window, frames, win, shift,
                           ldf, scale, preemph, remove_dc);
    else
        hipLaunchKernelGGL(fbank_frames_batch_kernel<float>, grid, dim3(256), 0, s,
                           reinterpret_cast<const float *>(wave), ld_wave, frame_off, window, frames, win, shift, ldf,
                           scale, preemph, remove_dc);
    asrk_prof_end_(PROF_FBANK, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_fbank_logmel_batch_f32(const void *wave, int sample_bytes, int64_t ld_wave,
                                           const int64_t *n_samples_host, const int64_t *frame_off, int B, int max_m,
                                           const float *window, const float *tw_fft, const float *tw_unpack,
                                           const float *melT, const int *mel_range, int nmel, int ld_mel, float *mel,
                                           int win, int shift, int log2n, float scale, float preemph, int remove_dc,
                                           float eps, void *stream) {
    if (B < 0 || max_m < 0 || win <= 0 || shift <= 0 || nmel <= 0 || ld_mel < nmel ||
        (sample_bytes != 2 && sample_bytes != 4))
        return ASRK_EINVAL;
    if (log2n < 8 || log2n > 10 || win > (1 << log2n)) return ASRK_ESHAPE;
    if (B == 0 || max_m == 0) return ASRK_OK;
    if (!wave || !frame_off || !window || !tw_fft || !tw_unpack || !melT || !mel_range || !mel) return ASRK_EINVAL;
    if ((int64_t)(max_m - 1) * shift + win > ld_wave) return ASRK_EINVAL;
    if (n_samples_host)
        for (int b = 0; b < B; ++b)
            if (n_samples_host[b] > ld_wave) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_FBANK, s);
    const dim3 grid(asrk_div_up(max_m, 4), B);
#define ASRK_FBANK_CASE(L2_)                                                                                       \
    case L2_:                                                                                                      \
        if (sample_bytes == 2)                                                                                     \
            hipLaunchKernelGGL((fbank_logmel_batch_kernel<int16_t, L2_>), grid, dim3(256), 0, s,                   \
                               reinterpret_cast<const int16_t *>(wave), ld_wave, frame_off, window, tw_fft,        \
                               tw_unpack, melT, mel_range, nmel, ld_mel, mel, win, shift, scale, preemph,          \
                               remove_dc, eps);                                                                    \
        else                                                                                                       \
            hipLaunchKernelGGL((fbank_logmel_batch_kernel<float, L2_>), grid, dim3(256), 0, s,                     \
                               reinterpret_cast<const float *>(wave), ld_wave, frame_off, window, tw_fft,          \
                               tw_unpack, melT, mel_range, nmel, ld_mel, mel, win, shift, scale, preemph,          \
                               remove_dc, eps);                                                                    \
        break;
    switch (log2n) {
        ASRK_FBANK_CASE(8) ASRK_FBANK_CASE(9) ASRK_FBANK_CASE(10)
    }
#undef ASRK_FBANK_CASE
    asrk_prof_end_(PROF_FBANK, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_delta_cmvn_batch_f32(const float *mel, const int64_t *frame_off, int B, int D,
                                         const float *filters, int C, int L, int apply_cmvn, float eps, float *out,
                                         int Tmax, void *stream) {
    if (B < 0 || D <= 0 || C <= 0 || L <= 0 || L > 16 || (L & 1) == 0 || Tmax < 0) return ASRK_EINVAL;
    if (B == 0 || Tmax == 0) return ASRK_OK;
    if (!mel || !frame_off || !filters || !out) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_FBANK, s);
    const dim3 grid(asrk_div_up(D, 64), C, B), block(64 * DC_WAVES);
#define ASRK_DC_CASE(LT_)                                                                                          \
    hipLaunchKernelGGL(delta_cmvn_batch_kernel<LT_>, grid, block, 0, s, mel, frame_off, D, filters, C, L, apply_cmvn, \
                       eps, out, Tmax)
    if (L == 1) ASRK_DC_CASE(1);
    else if (L <= 5) ASRK_DC_CASE(5);
    else if (L <= 9) ASRK_DC_CASE(9);
    else ASRK_DC_CASE(16);
#undef ASRK_DC_CASE
    asrk_prof_end_(PROF_FBANK, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}
