// Per-frame regularisers of the encoder / decoder stacks (src/module.py:116-119,135-138;
// src/asr.py:36,162,219,226): LayerNorm over the feature axis and inverted dropout.
// Both are one-pass HBM streaming kernels (wave64 per row / 16-B lanes).  Dropout keeps NO mask in
// memory: the keep-decision of element i is bit (Philox4x32-10(counter = i/4, key = seed))[i%4],
// so backward re-derives the same mask from (seed, offset) instead of reading 1 byte/element.
#include "common.h"
#include "knobs.h"
#include <algorithm>

namespace {

static inline bool al16h(const void *p) { return (((uintptr_t)p) & 15) == 0; }

// ---------------------------------------------------------------- LayerNorm
// one wave per row; biased variance, eps inside the sqrt (torch.nn.LayerNorm semantics)
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float *__restrict__ x,
                                                     const float *__restrict__ w,
                                                     const float *__restrict__ b,
                                                     float *__restrict__ y,
                                                     float *__restrict__ mean_out,
                                                     float *__restrict__ rstd_out, int rows, int cols,
                                                     float eps) {
    const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *xr = x + (size_t)row * cols;
    float *yr = y + (size_t)row * cols;
    float s = 0.f;
    for (int i = lane; i < cols; i += 64) s += xr[i];
    const float mean = wave_sum(s) / (float)cols;
    float v = 0.f;
    for (int i = lane; i < cols; i += 64) {
        const float d = xr[i] - mean;
        v += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)cols + eps);
    for (int i = lane; i < cols; i += 64) yr[i] = (xr[i] - mean) * rstd * w[i] + b[i];
    if (lane == 0) {
        mean_out[row] = mean;
        rstd_out[row] = rstd;
    }
}

// dx_r = rstd * (g*w - mean_c(g*w) - xhat * mean_c(g*w*xhat))
__global__ __launch_bounds__(256) void ln_bwd_data_kernel(const float *__restrict__ x,
                                                          const float *__restrict__ w,
                                                          const float *__restrict__ dy,
                                                          const float *__restrict__ mean_in,
                                                          const float *__restrict__ rstd_in,
                                                          float *__restrict__ dx, int rows, int cols) {
    const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *xr = x + (size_t)row * cols;
    const float *gr = dy + (size_t)row * cols;
    float *dr = dx + (size_t)row * cols;
    const float mean = mean_in[row], rstd = rstd_in[row];
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < cols; i += 64) {
        const float gw = gr[i] * w[i];
        s1 += gw;
        s2 += gw * (xr[i] - mean) * rstd;
    }
    s1 = wave_sum(s1) / (float)cols;
    s2 = wave_sum(s2) / (float)cols;
    for (int i = lane; i < cols; i += 64) {
        const float xh = (xr[i] - mean) * rstd;
        dr[i] = rstd * (gr[i] * w[i] - s1 - xh * s2);
    }
}

// dw_c = sum_r dy*xhat, db_c = sum_r dy : 64 columns x 4 row-lanes per block, row chunks on
// grid.y, atomics combine the chunks (outputs pre-zeroed by the launcher)
__global__ __launch_bounds__(256) void ln_bwd_param_kernel(const float *__restrict__ x,
                                                           const float *__restrict__ dy,
                                                           const float *__restrict__ mean_in,
                                                           const float *__restrict__ rstd_in,
                                                           float *__restrict__ dw,
                                                           float *__restrict__ db, int rows, int cols,
                                                           int rows_per_chunk) {
    __shared__ float pw[4][64], pb[4][64];
    const int l = threadIdx.x & 63;
    const int c = blockIdx.x * 64 + l;
    const int rl = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = blockIdx.y * rows_per_chunk;
    const int r1 = min(rows, r0 + rows_per_chunk);
    float sw = 0.f, sb = 0.f;
    if (c < cols) {
        for (int r = r0 + rl; r < r1; r += 4) {
            const float g = dy[(size_t)r * cols + c];
            sw += g * (x[(size_t)r * cols + c] - mean_in[r]) * rstd_in[r];
            sb += g;
        }
    }
    pw[rl][l] = sw;
    pb[rl][l] = sb;
    __syncthreads();
    if (rl == 0 && c < cols) {
        unsafeAtomicAdd(dw + c, pw[0][l] + pw[1][l] + pw[2][l] + pw[3][l]);
        unsafeAtomicAdd(db + c, pb[0][l] + pb[1][l] + pb[2][l] + pb[3][l]);
    }
}

// ---------------------------------------------------------------- dropout (Philox4x32-10)
__device__ __forceinline__ void philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3,
                                             uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
}

__device__ __forceinline__ void philox4x32_10(uint64_t ctr, uint64_t offset, uint64_t seed,
                                              uint32_t out[4]) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32);
    uint32_t c2 = (uint32_t)offset, c3 = (uint32_t)(offset >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// y[i] = keep(i) ? x[i] * scale : 0, keep(i) <=> (u32 >> 8) >= thresh24  (thresh24 = round(p*2^24))
template <bool VEC>
__global__ __launch_bounds__(256) void dropout_kernel(const float *__restrict__ x,
                                                      float *__restrict__ y, int64_t n,
                                                      uint32_t thresh24, float scale, uint64_t seed,
                                                      uint64_t offset) {
    const int64_t nq = (n + 3) >> 2;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq;
         q += (int64_t)gridDim.x * blockDim.x) {
        uint32_t r[4];
        philox4x32_10((uint64_t)q, offset, seed, r);
        if (VEC) {
            f32x4 v = reinterpret_cast<const f32x4 *>(x)[q];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ((r[j] >> 8) >= thresh24) ? v[j] * scale : 0.f;
            reinterpret_cast<f32x4 *>(y)[q] = v;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t i = q * 4 + j;
                if (i < n) y[i] = ((r[j] >> 8) >= thresh24) ? x[i] * scale : 0.f;
            }
        }
    }
}

}  // namespace

extern "C" int asrk_layer_norm_fwd_f32(const float *x, const float *weight, const float *bias,
                                       float *y, float *mean, float *rstd, int rows, int cols,
                                       float eps, void *stream) {
    if (rows < 0 || cols <= 0 || !(eps >= 0.f)) return ASRK_EINVAL;
    if (rows == 0) return ASRK_OK;
    if (!x || !weight || !bias || !y || !mean || !rstd) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_ROWOPS, s);
    hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)asrk_div_up(rows, 4)), dim3(256), 0, s, x,
                       weight, bias, y, mean, rstd, rows, cols, eps);
    asrk_prof_end_(PROF_ROWOPS, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_layer_norm_bwd_f32(const float *x, const float *weight, const float *dy,
                                       const float *mean, const float *rstd, float *dx, float *dweight,
                                       float *dbias, int rows, int cols, void *stream) {
    if (rows < 0 || cols <= 0) return ASRK_EINVAL;
    if (!dweight != !dbias) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (dweight) {
        ASRK_HIP(hipMemsetAsync(dweight, 0, sizeof(float) * cols, s));
        ASRK_HIP(hipMemsetAsync(dbias, 0, sizeof(float) * cols, s));
    }
    if (rows == 0) return ASRK_OK;
    if (!x || !weight || !dy || !mean || !rstd) return ASRK_EINVAL;
    asrk_prof_begin_(PROF_ROWOPS, s);
    if (dx)
        hipLaunchKernelGGL(ln_bwd_data_kernel, dim3((unsigned)asrk_div_up(rows, 4)), dim3(256), 0, s,
                           x, weight, dy, mean, rstd, dx, rows, cols);
    if (dweight) {
        // ASRK_DETERMINISTIC: one row chunk - a column's fixed-order block sum is the only value added to dw / db
        const int chunks = asrk_knobs_().get(asrk_knobs_().deterministic, 0)
                               ? 1 : std::max(1, std::min(asrk_div_up(rows, 64), 256));
        const int rpc = asrk_div_up(asrk_div_up(rows, chunks), 4) * 4;
        hipLaunchKernelGGL(ln_bwd_param_kernel,
                           dim3((unsigned)asrk_div_up(cols, 64), (unsigned)asrk_div_up(rows, rpc)),
                           dim3(256), 0, s, x, dy, mean, rstd, dweight, dbias, rows, cols, rpc);
    }
    asrk_prof_end_(PROF_ROWOPS, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_dropout_f32(const float *x, float *y, int64_t n, float p, uint64_t seed,
                                uint64_t offset, void *stream) {
    if (n < 0 || !(p >= 0.f) || !(p < 1.f)) return ASRK_EINVAL;
    if (n == 0) return ASRK_OK;
    if (!x || !y) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t thresh24 = (uint32_t)llrintf(p * 16777216.0f);
    const float scale = 1.0f / (1.0f - p);
    const bool vec = al16h(x) && al16h(y) && (n % 4 == 0);
    const int64_t nq = (n + 3) / 4;
    const unsigned grid = (unsigned)std::min<int64_t>(8192, asrk_div_up64(nq, 256));
    asrk_prof_begin_(PROF_ROWOPS, s);
    if (vec)
        hipLaunchKernelGGL((dropout_kernel<true>), dim3(grid), dim3(256), 0, s, x, y, n, thresh24,
                           scale, (uint64_t)seed, (uint64_t)offset);
    else
        hipLaunchKernelGGL((dropout_kernel<false>), dim3(grid), dim3(256), 0, s, x, y, n, thresh24,
                           scale, (uint64_t)seed, (uint64_t)offset);
    asrk_prof_end_(PROF_ROWOPS, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}
