// GRU cell (one time step) — the `module: 'GRU'` option of the encoder layers, the decoder and the
// RNN-LM (reference: src/module.py:112-113, src/asr.py:175-176, src/lm.py:20-21 -> torch.nn.GRU).
// torch gate order (r, z, n):
//   r = sigmoid(gi_r + gh_r),  z = sigmoid(gi_z + gh_z),  n = tanh(gi_n + r * gh_n),
//   h' = (1 - z) * n + z * h            with gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh (two GEMMs).
// The step GEMMs are the skinny weight-streaming kernels of gemm.hip; this file is the elementwise
// gate math of ONE step.  Whole GRU layers (encoder, RNN-LM training) run in the persistent recurrence kernels of
// lstm_rec.hip in their GRU mode (asrk_gru_rec_{fwd,bwd}_f32); these cell kernels serve the single decoder / LM
// decode steps and hidden sizes that are not a multiple of 4.
#include "common.h"

namespace {

// gi/gh rows of 3H (row strides ldi/ldh); writes r, z, n over gi (gh keeps gh_n for the backward)
__global__ void gru_cell_fwd_kernel(float *__restrict__ gi, const float *__restrict__ gh, int64_t ldi,
                                    int64_t ldh, const float *__restrict__ h_prev, int64_t ldp,
                                    float *__restrict__ h_new, int64_t ldn, int B, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, u = i - b * H;
    float *a = gi + (size_t)b * ldi + u;
    const float *g = gh + (size_t)b * ldh + u;
    const float r = sigmoidf_acc(a[0] + g[0]);
    const float z = sigmoidf_acc(a[H] + g[H]);
    const float n = tanhf(a[2 * H] + r * g[2 * H]);
    const float hp = h_prev ? h_prev[(size_t)b * ldp + u] : 0.f;
    a[0] = r; a[H] = z; a[2 * H] = n;
    h_new[(size_t)b * ldn + u] = (1.f - z) * n + z * hp;
}

// gi holds (r, z, n), gh holds (.., .., gh_n); dh (+ dh2, contiguous [B,H]) = gradient w.r.t. h'.  Writes dgi over gi, dgh over
// gh and dh_prev = dh * z (the part of dL/dh that does not go through gh; the caller adds dgh W_hh).
__global__ void gru_cell_bwd_kernel(float *__restrict__ gi, float *__restrict__ gh, int64_t ldi,
                                    int64_t ldh, const float *__restrict__ h_prev, int64_t ldp,
                                    const float *__restrict__ dh, int64_t ldd,
                                    const float *__restrict__ dh2,
                                    float *__restrict__ dh_prev, int64_t ldo, int B, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, u = i - b * H;
    float *a = gi + (size_t)b * ldi + u;
    float *g = gh + (size_t)b * ldh + u;
    const float r = a[0], z = a[H], n = a[2 * H], ghn = g[2 * H];
    const float hp = h_prev ? h_prev[(size_t)b * ldp + u] : 0.f;
    const float d = (dh ? dh[(size_t)b * ldd + u] : 0.f) + (dh2 ? dh2[(size_t)b * H + u] : 0.f);
    const float dn = d * (1.f - z) * (1.f - n * n);       // through tanh
    const float dz = d * (hp - n) * z * (1.f - z);
    const float dr = dn * ghn * r * (1.f - r);
    a[0] = dr; a[H] = dz; a[2 * H] = dn;
    g[0] = dr; g[H] = dz; g[2 * H] = dn * r;
    dh_prev[(size_t)b * ldo + u] = d * z;
}

inline unsigned blocks_for(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

}  // namespace

extern "C" int asrk_gru_cell_fwd_f32(float *gi, const float *gh, int64_t ldi, int64_t ldh,
                                     const float *h_prev, int64_t ldp, float *h_new, int64_t ldn, int B,
                                     int H, void *stream) {
    if (B < 0 || H <= 0 || ldi < 3 * H || ldh < 3 * H || ldn < H || (h_prev && ldp < H)) return ASRK_EINVAL;
    if (B == 0) return ASRK_OK;
    if (!gi || !gh || !h_new) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_CELL, s);
    hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3(blocks_for((int64_t)B * H, 256)), dim3(256), 0, s, gi, gh,
                       ldi, ldh, h_prev, ldp, h_new, ldn, B, H);
    asrk_prof_end_(PROF_CELL, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_gru_cell_bwd_f32(float *gi, float *gh, int64_t ldi, int64_t ldh, const float *h_prev,
                                     int64_t ldp, const float *dh, int64_t ldd, const float *dh2,
                                     float *dh_prev, int64_t ldo, int B, int H, void *stream) {
    if (B < 0 || H <= 0 || ldi < 3 * H || ldh < 3 * H || ldd < H || ldo < H || (h_prev && ldp < H))
        return ASRK_EINVAL;
    if (B == 0) return ASRK_OK;
    if (!gi || !gh || (!dh && !dh2) || !dh_prev) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_CELL, s);
    hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3(blocks_for((int64_t)B * H, 256)), dim3(256), 0, s, gi, gh,
                       ldi, ldh, h_prev, ldp, dh, ldd, dh2, dh_prev, ldo, B, H);
    asrk_prof_end_(PROF_CELL, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}
