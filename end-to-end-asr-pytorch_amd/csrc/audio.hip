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

// Delta (src/audio.py:33-80) -> CMVN (src/audio.py:7-30) -> Postprocess (src/audio.py:83-89) -> zero padding
// of the batch (pad_sequence, src/data.py:39) in ONE pass over a batch of frame-major features
// mel [total frames, D]: workgroup = (utterance, channel c, block of 64 features), lane = feature, the four
// waves stride over time.  y[c,d,t] = sum_j filt[c,j] mel[t + j - half, d] (zero outside the utterance) is
// recomputed from L2 in each of the three sweeps (mean, unbiased variance, normalised store) - 9 taps of a
// 16-MB tensor - instead of being materialised as [C, D, T] and transposed back.
__global__ __launch_bounds__(256) void delta_cmvn_batch_kernel(const float *__restrict__ mel,
                                                               const int64_t *__restrict__ frame_off, int D,
                                                               const float *__restrict__ filt, int C, int L,
                                                               int apply_cmvn, float eps, float *__restrict__ out,
                                                               int Tmax) {
    __shared__ float red[4][64];
    const int b = blockIdx.z, c = blockIdx.y, d = blockIdx.x * 64 + (threadIdx.x & 63);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int64_t off = frame_off[b];
    const int m = (int)(frame_off[b + 1] - off), half = (L - 1) / 2;
    const bool live = d < D;
    const float *x = mel + (size_t)off * D + (live ? d : 0);
    float fl[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) fl[j] = j < L ? filt[c * L + j] : 0.f;
    auto y_at = [&](int t) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int tt = t + j - half;
            if (j < L && tt >= 0 && tt < m) acc += fl[j] * x[(size_t)tt * D];
        }
        return acc;
    };
    float mean = 0.f, inv = 1.f;
    if (apply_cmvn) {
        float s = 0.f;
        if (live) for (int t = wave; t < m; t += 4) s += y_at(t);
        red[wave][lane] = s;
        __syncthreads();
        mean = (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) / (float)m;
        __syncthreads();
        float v = 0.f;
        if (live) for (int t = wave; t < m; t += 4) { const float q = y_at(t) - mean; v += q * q; }
        red[wave][lane] = v;
        __syncthreads();
        const float var = (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) / (float)(m - 1);
        inv = 1.f / (eps + sqrtf(var));                       // m == 1 -> nan, as torch.std
    }
    if (!live) return;
    float *o = out + (size_t)b * Tmax * C * D + (size_t)c * D + d;
    for (int t = wave; t < Tmax; t += 4) o[(size_t)t * C * D] = t < m ? (y_at(t) - mean) * inv : 0.f;
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

extern "C" int asrk_delta_cmvn_batch_f32(const float *mel, const int64_t *frame_off, int B, int D,
                                         const float *filters, int C, int L, int apply_cmvn, float eps, float *out,
                                         int Tmax, void *stream) {
    if (B < 0 || D <= 0 || C <= 0 || L <= 0 || L > 16 || (L & 1) == 0 || Tmax < 0) return ASRK_EINVAL;
    if (B == 0 || Tmax == 0) return ASRK_OK;
    if (!mel || !frame_off || !filters || !out) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_FBANK, s);
    hipLaunchKernelGGL(delta_cmvn_batch_kernel, dim3(asrk_div_up(D, 64), C, B), dim3(256), 0, s, mel, frame_off, D,
                       filters, C, L, apply_cmvn, eps, out, Tmax);
    asrk_prof_end_(PROF_FBANK, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}
