// Attention-decoder step kernels for gfx950 (reference: src/module.py:179-258 BaseAttention /
// ScaleDotAttention / LocationAwareAttention, src/asr.py:277-313 Attention.forward, src/asr.py:218
// the 1-step decoder LSTM, src/asr.py:103-109,134,142 the embedding look-ups).
//
// One decode step touches only ~60 MB (K 7.7 MB + V 52 MB at cfg3) and ~0.1 GFLOP: these are
// HBM/L2-streaming kernels, not MFMA work.  Layout: lanes run along the contiguous feature axis
// (A / Dv) so every global access is a coalesced row segment; per-frame reductions are wave64
// shuffles; the small per-batch state (conv output, energies) lives in LDS.
//
//  loc_conv          c[b,t,k]   = sum_{n,j} prev_att[b,n,t+j-ks] * Wc[k,n,j]      (zero padded)
//  loc_energy        e[bn,t]    = we . tanh(key[bn,t,:] + q[bn,:] + tanh(Wp c[b,t,:])) + be
//  dot_energy        e[bn,t]    = key[bn,t,:] . q[bn,:]
//  masked softmax    attn[bn,:] = softmax(e / temperature) over t < len[b]   (0 beyond)
//  context           ctx[bn,:]  = sum_t attn[bn,t] * value[bn,t,:]
// and the matching backward kernels (weight / key gradients are ACCUMULATED in place into
// caller-owned buffers so a 64-step decode adds no per-step gradient tensors).
#include "common.h"
#include "knobs.h"

namespace {

// ------------------------------------------------------------------ location conv
// grid (B, ceil(T/64)), block 256: wave w computes channels k = w, w+4, ... for 64 frames
__global__ __launch_bounds__(256) void loc_conv_fwd_kernel(const float *__restrict__ prev,
                                                           const float *__restrict__ Wc,
                                                           float *__restrict__ c, int N, int T,
                                                           int K, int ks) {
    extern __shared__ float sp[];  // [N][64 + 2ks]
    const int b = blockIdx.x, t0 = blockIdx.y * 64;
    const int KW = 2 * ks + 1, span = 64 + 2 * ks;
    for (int i = threadIdx.x; i < N * span; i += 256) {
        const int n = i / span, o = i - n * span;
        const int t = t0 + o - ks;
        sp[i] = (t >= 0 && t < T) ? prev[((size_t)b * N + n) * T + t] : 0.f;
    }
    __syncthreads();
    const int tl = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (t0 + tl >= T) return;
    for (int k = w; k < K; k += 4) {
        float acc = 0.f;
        for (int n = 0; n < N; ++n) {
            const float *wr = Wc + ((size_t)k * N + n) * KW;
            const float *pr = sp + n * span + tl;
            for (int j = 0; j < KW; ++j) acc += pr[j] * wr[j];
        }
        c[((size_t)b * T + t0 + tl) * K + k] = acc;
    }
}

// dprev[b,n,s] = sum_{k,j} dc[b, s-j+ks, k] * Wc[k,n,j];  grid (B, ceil(T/64)), block 256.
// The (k, j) reduction is split over the 4 waves by channel (k = w, w+4, ...) and combined through
// LDS: with the usual single head, looping heads over waves left three of them idle (135 us per
// call at cfg3: K=10, 201 taps).  The dc window is stored channel-major so the 64 lanes of a wave
// read consecutive LDS words (frame-major with K=10 was a 4-way bank conflict); the filter taps are
// wave-uniform.
__global__ __launch_bounds__(256) void loc_conv_bwd_data_kernel(const float *__restrict__ dc,
                                                                const float *__restrict__ Wc,
                                                                float *__restrict__ dprev, int N,
                                                                int T, int K, int ks) {
    extern __shared__ float sd[];  // [K][64 + 2ks] window of dc, then [4][N][64] partial sums
    const int b = blockIdx.x, s0 = blockIdx.y * 64;
    const int KW = 2 * ks + 1, span = 64 + 2 * ks;
    float *part = sd + K * span;
    for (int i = threadIdx.x; i < span * K; i += 256) {
        const int o = i / K, k = i - o * K;
        const int t = s0 + o - ks;
        sd[k * span + o] = (t >= 0 && t < T) ? dc[((size_t)b * T + t) * K + k] : 0.f;
    }
    __syncthreads();
    const int sl = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int n = 0; n < N; ++n) {
        float acc = 0.f;
        for (int k = w; k < K; k += 4) {
            const float *wr = Wc + ((size_t)k * N + n) * KW;
            // t = s - j + ks  ->  window column (sl + 2ks - j)
            const float *col = sd + k * span + sl + 2 * ks;
            for (int j = 0; j < KW; ++j) acc += col[-j] * wr[j];
        }
        part[(w * N + n) * 64 + sl] = acc;
    }
    __syncthreads();
    if (s0 + sl >= T) return;
    for (int n = w; n < N; n += 4)
        dprev[((size_t)b * N + n) * T + s0 + sl] =
            (part[n * 64 + sl] + part[(N + n) * 64 + sl]) + (part[(2 * N + n) * 64 + sl] + part[(3 * N + n) * 64 + sl]);
}

// dWc[k,n,j] += sum_{b,t} dc[b,t,k] * prev[b,n,t+j-ks];  grid (K*N, B), block 256 (threads over j)
__global__ __launch_bounds__(256) void loc_conv_bwd_weight_kernel(const float *__restrict__ dc,
                                                                  const float *__restrict__ prev,
                                                                  float *__restrict__ dWc, int N, int T,
                                                                  int K, int ks) {
    const int k = blockIdx.x / N, n = blockIdx.x % N, b = blockIdx.y;
    const int KW = 2 * ks + 1;
    const float *pr = prev + ((size_t)b * N + n) * T;
    const float *dr = dc + (size_t)b * T * K + k;
    for (int j = threadIdx.x; j < KW; j += 256) {
        float acc = 0.f;
        const int lo = max(0, ks - j), hi = min(T, T + ks - j);
        for (int t = lo; t < hi; ++t) acc += dr[(size_t)t * K] * pr[t + j - ks];
        unsafeAtomicAdd(dWc + ((size_t)k * N + n) * KW + j, acc);
    }
}

// ------------------------------------------------------------------ energy + masked softmax
// LOC: location-aware (module.py:245-256); else scaled-dot (module.py:204-212).
// grid (B*N, TC): the frames of one (batch, head) row are split over TC workgroups so a decode
// step fills the chip (B*N = 32 workgroups alone left 7/8 of the CUs idle: 1.3 ms per call).
struct EnergyArgs {
    const float *key, *q;        // [BN,T,A], [BN,A]
    const float *c, *Wp;         // [B,T,K], [A,K]     (LOC)
    const float *we, *be;        // [A], [1]           (LOC)
    const int64_t *lens;         // [B]
    float *e;                    // [BN,T] scaled, masked energies (scratch)
    int N, T, A, K, tpb;         // tpb = frames per workgroup
    float inv_temp;
};

template <bool LOC>
__global__ __launch_bounds__(256) void energy_fwd_kernel(EnergyArgs p) {
    extern __shared__ float sm[];
    const int bn = blockIdx.x, b = bn / p.N;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, A = p.A, K = p.K;
    const int t0 = blockIdx.y * p.tpb, t1 = min(T, t0 + p.tpb);
    float *s_q = sm;                        // [A]
    float *s_we = s_q + A;                  // [A]   (LOC)
    float *s_wp = s_we + (LOC ? A : 0);     // [A*K] (LOC)
    float *s_c = s_wp + (LOC ? A * K : 0);  // [tpb*K] (LOC)
    for (int i = tid; i < A; i += 256) {
        s_q[i] = p.q[(size_t)bn * A + i];
        if (LOC) s_we[i] = p.we[i];
    }
    if (LOC) {
        for (int i = tid; i < A * K; i += 256) s_wp[i] = p.Wp[i];
        for (int i = tid; i < (t1 - t0) * K; i += 256) s_c[i] = p.c[((size_t)b * T + t0) * K + i];
    }
    __syncthreads();
    const int len = min((int)p.lens[b], T);
    const float be = LOC ? p.be[0] : 0.f;
    for (int t = t0 + wave; t < t1; t += 4) {
        const float *kr = p.key + ((size_t)bn * T + t) * A;
        float part = 0.f;
        for (int a = lane; a < A; a += 64) {
            if (LOC) {
                float u = 0.f;
                for (int k = 0; k < K; ++k) u += s_wp[a * K + k] * s_c[(t - t0) * K + k];
                part += s_we[a] * tanhf(kr[a] + s_q[a] + tanhf(u));
            } else {
                part += kr[a] * s_q[a];
            }
        }
        part = wave_sum(part);
        if (lane == 0) p.e[(size_t)bn * T + t] = (t < len) ? (part + be) * p.inv_temp : -INFINITY;
    }
}

// attn[bn,:] = softmax(e[bn,:]) with -inf entries -> 0 ; grid (BN), block 256
__global__ __launch_bounds__(256) void masked_softmax_kernel(const float *__restrict__ e,
                                                             float *__restrict__ attn, int T) {
    __shared__ float s_red[8];
    const int bn = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float *er = e + (size_t)bn * T;
    float m = -INFINITY;
    for (int t = tid; t < T; t += 256) m = fmaxf(m, er[t]);
    m = wave_max(m);
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    float sum = 0.f;
    for (int t = tid; t < T; t += 256) sum += expf(er[t] - m);   // exp(-inf) = 0 for masked frames
    sum = wave_sum(sum);
    if (lane == 0) s_red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.f / (s_red[4] + s_red[5] + s_red[6] + s_red[7]);
    for (int t = tid; t < T; t += 256) attn[(size_t)bn * T + t] = expf(er[t] - m) * inv;
}

struct EnergyBwdArgs {
    const float *key, *q, *c, *Wp, *we;
    const int64_t *lens;
    const float *attn, *dattn;   // [BN,T]
    float *dkey_acc;             // [BN,T,A]  += (accumulated over decode steps)
    float *dq;                   // [BN,A]    += (zeroed by the launcher)
    float *dc;                   // [B,T,K]   += over heads (zeroed by the launcher)   (LOC)
    float *dWp_acc, *dwe_acc, *dbe_acc;  // += (LOC)
    int N, T, A, K, tpb;
    float inv_temp;
};

template <bool LOC>
__global__ __launch_bounds__(256) void energy_bwd_kernel(EnergyBwdArgs p) {
    extern __shared__ float sm[];
    const int bn = blockIdx.x, b = bn / p.N;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, A = p.A, K = p.K;
    const int t0 = blockIdx.y * p.tpb, t1 = min(T, t0 + p.tpb);
    float *s_q = sm;                            // [A]
    float *s_dq = s_q + A;                      // [4][A] per-wave partial dq
    float *s_we = s_dq + 4 * A;                 // [A]
    float *s_dwe = s_we + (LOC ? A : 0);        // [4][A]
    float *s_wp = s_dwe + (LOC ? 4 * A : 0);    // [A*K]
    float *s_c = s_wp + (LOC ? A * K : 0);      // [tpb*K]
    float *s_dwp = s_c + (LOC ? p.tpb * K : 0); // [4][A*K] per-wave partial dWp (no atomics)
    __shared__ float s_red[4];
    for (int i = tid; i < A; i += 256) {
        s_q[i] = p.q[(size_t)bn * A + i];
        if (LOC) s_we[i] = p.we[i];
    }
    for (int i = tid; i < 4 * A; i += 256) {
        s_dq[i] = 0.f;
        if (LOC) s_dwe[i] = 0.f;
    }
    if (LOC) {
        for (int i = tid; i < 4 * A * K; i += 256) s_dwp[i] = 0.f;
        for (int i = tid; i < A * K; i += 256) s_wp[i] = p.Wp[i];
        for (int i = tid; i < (t1 - t0) * K; i += 256) s_c[i] = p.c[((size_t)b * T + t0) * K + i];
    }
    // softmax backward needs the full-row dot product sum_t attn*dattn (cheap: 2T reads)
    float dot = 0.f;
    for (int t = tid; t < T; t += 256) dot += p.attn[(size_t)bn * T + t] * p.dattn[(size_t)bn * T + t];
    dot = wave_sum(dot);
    if (lane == 0) s_red[wave] = dot;
    __syncthreads();
    dot = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    const int len = min((int)p.lens[b], T);
    float dbe = 0.f;
    float *wdq = s_dq + wave * A, *wdwe = s_dwe + wave * A, *wdwp = s_dwp + wave * A * K;
    for (int t = t0 + wave; t < min(t1, len); t += 4) {
        const float de = p.attn[(size_t)bn * T + t] * (p.dattn[(size_t)bn * T + t] - dot) * p.inv_temp;
        dbe += de;   // every lane holds the same value; lane 0's copy is used below
        const float *kr = p.key + ((size_t)bn * T + t) * A;
        float *dkr = p.dkey_acc + ((size_t)bn * T + t) * A;
        float dck[16];  // K <= 16 partial dc for this frame (LOC)
#pragma unroll
        for (int k = 0; k < 16; ++k) dck[k] = 0.f;
        for (int a = lane; a < A; a += 64) {
            if (LOC) {
                const float *cr = s_c + (t - t0) * K;
                float u = 0.f;
                for (int k = 0; k < K; ++k) u += s_wp[a * K + k] * cr[k];
                const float loc = tanhf(u);
                const float z = tanhf(kr[a] + s_q[a] + loc);
                const float dz = de * s_we[a] * (1.f - z * z);
                dkr[a] += dz;
                wdq[a] += dz;
                wdwe[a] += de * z;
                const float du = dz * (1.f - loc * loc);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if (k < K) {
                        dck[k] += du * s_wp[a * K + k];
                        wdwp[a * K + k] += du * cr[k];
                    }
                }
            } else {
                dkr[a] += de * s_q[a];
                wdq[a] += de * kr[a];
            }
        }
        if (LOC) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k < K) {
                    const float v = wave_sum(dck[k]);
                    if (lane == 0) unsafeAtomicAdd(p.dc + ((size_t)b * T + t) * K + k, v);
                }
            }
        }
    }
    __syncthreads();
    for (int a = tid; a < A; a += 256) {
        unsafeAtomicAdd(p.dq + (size_t)bn * A + a, s_dq[a] + s_dq[A + a] + s_dq[2 * A + a] + s_dq[3 * A + a]);
        if (LOC)
            unsafeAtomicAdd(p.dwe_acc + a, s_dwe[a] + s_dwe[A + a] + s_dwe[2 * A + a] + s_dwe[3 * A + a]);
    }
    if (LOC) {
        const int AK = A * K;
        for (int i = tid; i < AK; i += 256)
            unsafeAtomicAdd(p.dWp_acc + i, s_dwp[i] + s_dwp[AK + i] + s_dwp[2 * AK + i] + s_dwp[3 * AK + i]);
        if (lane == 0) unsafeAtomicAdd(p.dbe_acc, dbe);
    }
}

// ------------------------------------------------------------------ context = attn . value
// grid (BN, ceil(Dv/256)): thread d accumulates over t (coalesced along d)
__global__ __launch_bounds__(256) void context_fwd_kernel(const float *__restrict__ attn,
                                                          const float *__restrict__ value,
                                                          float *__restrict__ ctx, int T, int Dv,
                                                          int64_t ctx_stride) {
    extern __shared__ float sa[];  // [T]
    const int bn = blockIdx.x, d = blockIdx.y * 256 + threadIdx.x;
    for (int t = threadIdx.x; t < T; t += 256) sa[t] = attn[(size_t)bn * T + t];
    __syncthreads();
    if (d >= Dv) return;
    const float *v = value + (size_t)bn * T * Dv + d;
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc += sa[t] * v[(size_t)t * Dv];
    ctx[(size_t)bn * ctx_stride + d] = acc;
}

// dattn[bn,t] = dctx[bn,:] . value[bn,t,:] ; grid (BN, ceil(T/4)), one wave per frame
__global__ __launch_bounds__(256) void context_bwd_attn_kernel(const float *__restrict__ dctx,
                                                               const float *__restrict__ value,
                                                               float *__restrict__ dattn, int T, int Dv,
                                                               int64_t dctx_stride) {
    const int bn = blockIdx.x, t = blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= T) return;
    const float *v = value + ((size_t)bn * T + t) * Dv;
    const float *g = dctx + (size_t)bn * dctx_stride;
    float acc = 0.f;
    for (int d = lane; d < Dv; d += 64) acc += g[d] * v[d];
    acc = wave_sum(acc);
    if (lane == 0) dattn[(size_t)bn * T + t] = acc;
}

// ------------------------------------------------------------------ LSTM cell (one step)
// gates [B,4H] pre-activations (i,f,g,o) -> activated in place; c_prev -> c, h
__global__ void lstm_cell_fwd_kernel(float *__restrict__ gates, const float *__restrict__ c_prev,
                                     float *__restrict__ c, float *__restrict__ h, int B, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, u = i - b * H;
    float *g = gates + (size_t)b * 4 * H + u;
    const float gi = sigmoidf_acc(g[0]), gf = sigmoidf_acc(g[H]);
    const float gg = tanhf(g[2 * H]), go = sigmoidf_acc(g[3 * H]);
    const float cn = gf * c_prev[i] + gi * gg;
    g[0] = gi; g[H] = gf; g[2 * H] = gg; g[3 * H] = go;
    c[i] = cn;
    h[i] = go * tanhf(cn);
}

// gates: activated (in) -> pre-activation grads (out, in place); dc_prev = dc_total * f
__global__ void lstm_cell_bwd_kernel(float *__restrict__ gates, const float *__restrict__ c_prev,
                                     const float *__restrict__ c, const float *__restrict__ dh,
                                     const float *__restrict__ dc_in, float *__restrict__ dc_prev,
                                     int B, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, u = i - b * H;
    float *g = gates + (size_t)b * 4 * H + u;
    const float gi = g[0], gf = g[H], gg = g[2 * H], go = g[3 * H];
    const float tc = tanhf(c[i]);
    const float dhv = dh ? dh[i] : 0.f;
    const float dct = dhv * go * (1.f - tc * tc) + (dc_in ? dc_in[i] : 0.f);
    g[0] = dct * gg * gi * (1.f - gi);
    g[H] = dct * c_prev[i] * gf * (1.f - gf);
    g[2 * H] = dct * gi * (1.f - gg * gg);
    g[3 * H] = dhv * tc * go * (1.f - go);
    dc_prev[i] = dct * gf;
}

// ------------------------------------------------------------------ embedding
__global__ void embedding_fwd_kernel(const int64_t *__restrict__ idx, const float *__restrict__ W,
                                     float *__restrict__ out, int64_t n, int D, int V) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * D) return;
    const int64_t r = i / D;
    const int d = (int)(i - r * D);
    const int64_t v = idx[r];
    out[i] = (v >= 0 && v < V) ? W[(size_t)v * D + d] : 0.f;
}

__global__ void embedding_bwd_kernel(const int64_t *__restrict__ idx, const float *__restrict__ dout,
                                     float *__restrict__ dW, int64_t n, int D, int V) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * D) return;
    const int64_t r = i / D;
    const int d = (int)(i - r * D);
    const int64_t v = idx[r];
    if (v >= 0 && v < V) unsafeAtomicAdd(dW + (size_t)v * D + d, dout[i]);
}

// deterministic form (ASRK_DETERMINISTIC): thread d walks the tokens in order, so repeated ids add in a fixed order
__global__ void embedding_bwd_ordered_kernel(const int64_t *__restrict__ idx, const float *__restrict__ dout,
                                             float *__restrict__ dW, int64_t n, int D, int V) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    for (int64_t r = 0; r < n; ++r) {
        const int64_t v = idx[r];
        if (v >= 0 && v < V) dW[(size_t)v * D + d] += dout[r * D + d];
    }
}

inline unsigned blocks_for(int64_t n, int bs) { return (unsigned)asrk_div_up64(n, bs); }

}  // namespace

extern "C" int asrk_loc_conv_fwd_f32(const float *prev_att, const float *Wc, float *c, int B, int N,
                                     int T, int K, int ks, void *stream) {
    if (B < 0 || N <= 0 || T <= 0 || K <= 0 || ks < 0) return ASRK_EINVAL;
    if (B == 0) return ASRK_OK;
    if (!prev_att || !Wc || !c) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)N * (64 + 2 * ks) * sizeof(float);
    if (lds > 60 * 1024) return ASRK_ESHAPE;
    asrk_prof_begin_(PROF_ATTN, s);
    hipLaunchKernelGGL(loc_conv_fwd_kernel, dim3(B, asrk_div_up(T, 64)), dim3(256), lds, s, prev_att, Wc,
                       c, N, T, K, ks);
    asrk_prof_end_(PROF_ATTN, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_loc_conv_bwd_f32(const float *dc, const float *prev_att, const float *Wc,
                                     float *dprev_att, float *dWc_acc, int B, int N, int T, int K, int ks,
                                     void *stream) {
    if (B < 0 || N <= 0 || T <= 0 || K <= 0 || ks < 0) return ASRK_EINVAL;
    if (B == 0) return ASRK_OK;
    if (!dc || !prev_att || !Wc) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = ((size_t)(64 + 2 * ks) * K + (size_t)4 * N * 64) * sizeof(float);
    if (lds > 60 * 1024) return ASRK_ESHAPE;
    asrk_prof_begin_(PROF_ATTN, s);
    if (dprev_att)
        hipLaunchKernelGGL(loc_conv_bwd_data_kernel, dim3(B, asrk_div_up(T, 64)), dim3(256), lds, s, dc, Wc,
                           dprev_att, N, T, K, ks);
    if (dWc_acc)
        hipLaunchKernelGGL(loc_conv_bwd_weight_kernel, dim3(K * N, B), dim3(256), 0, s, dc, prev_att,
                           dWc_acc, N, T, K, ks);
    asrk_prof_end_(PROF_ATTN, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

// frames per workgroup: enough workgroups to cover the chip a few times, >= 8 frames each
static inline int energy_tpb(int BN, int T) {
    int tc = asrk_div_up(1024, BN > 0 ? BN : 1);
    if (tc > asrk_div_up(T, 8)) tc = asrk_div_up(T, 8);
    if (tc < 1) tc = 1;
    return asrk_div_up(T, tc);
}

extern "C" int asrk_attn_energy_fwd_f32(int loc, const float *key, const float *q, const float *c,
                                        const float *Wp, const float *we, const float *be,
                                        const int64_t *lens, float *attn, float *e_scratch, int B,
                                        int N, int T, int A, int K, float temperature, void *stream) {
    if (B < 0 || N <= 0 || T <= 0 || A <= 0 || temperature == 0.f) return ASRK_EINVAL;
    if (B == 0) return ASRK_OK;
    if (!key || !q || !lens || !attn || !e_scratch) return ASRK_EINVAL;
    if (loc && (!c || !Wp || !we || !be || K <= 0)) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int tpb = energy_tpb(B * N, T);
    EnergyArgs a{key, q, c, Wp, we, be, lens, e_scratch, N, T, A, loc ? K : 0, tpb, 1.f / temperature};
    const size_t lds = (size_t)(A + (loc ? A + A * K + tpb * K : 0)) * sizeof(float);
    if (lds > 150 * 1024) return ASRK_ESHAPE;
    const dim3 grid(B * N, asrk_div_up(T, tpb));
    asrk_prof_begin_(PROF_ATTN, s);
    if (loc) {
        ASRK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(energy_fwd_kernel<true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(energy_fwd_kernel<true>, grid, dim3(256), lds, s, a);
    } else {
        ASRK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(energy_fwd_kernel<false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(energy_fwd_kernel<false>, grid, dim3(256), lds, s, a);
    }
    hipLaunchKernelGGL(masked_softmax_kernel, dim3(B * N), dim3(256), 0, s, e_scratch, attn, T);
    asrk_prof_end_(PROF_ATTN, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_attn_energy_bwd_f32(int loc, const float *key, const float *q, const float *c,
                                        const float *Wp, const float *we, const int64_t *lens,
                                        const float *attn, const float *dattn, float *dkey_acc,
                                        float *dq, float *dc, float *dWp_acc, float *dwe_acc,
                                        float *dbe_acc, int B, int N, int T, int A, int K,
                                        float temperature, void *stream) {
    if (B < 0 || N <= 0 || T <= 0 || A <= 0 || temperature == 0.f) return ASRK_EINVAL;
    if (B == 0) return ASRK_OK;
    if (!key || !q || !lens || !attn || !dattn || !dkey_acc || !dq) return ASRK_EINVAL;
    if (loc && K > 16) return ASRK_ESHAPE;
    if (loc && (!c || !Wp || !we || !dc || !dWp_acc || !dwe_acc || !dbe_acc || K <= 0))
        return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int tpb = energy_tpb(B * N, T);
    size_t lds = 0;
    for (;;) {  // shrink-proof: the per-wave dWp partials dominate LDS, tpb only adds tpb*K
        lds = (size_t)(A + 4 * A + (loc ? A + 4 * A + A * K + tpb * K + 4 * A * K : 0)) * sizeof(float);
        if (lds <= 150 * 1024 || tpb <= 8) break;
        tpb = (tpb + 1) / 2;
    }
    if (lds > 150 * 1024) return ASRK_ESHAPE;
    EnergyBwdArgs a{key, q, c, Wp, we, lens, attn, dattn, dkey_acc, dq, dc, dWp_acc, dwe_acc, dbe_acc,
                    N, T, A, loc ? K : 0, tpb, 1.f / temperature};
    const dim3 grid(B * N, asrk_div_up(T, tpb));
    asrk_prof_begin_(PROF_ATTN, s);
    ASRK_HIP(hipMemsetAsync(dq, 0, (size_t)B * N * A * sizeof(float), s));
    if (loc) {
        ASRK_HIP(hipMemsetAsync(dc, 0, (size_t)B * T * K * sizeof(float), s));
        ASRK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(energy_bwd_kernel<true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(energy_bwd_kernel<true>, grid, dim3(256), lds, s, a);
    } else {
        ASRK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(energy_bwd_kernel<false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(energy_bwd_kernel<false>, grid, dim3(256), lds, s, a);
    }
    asrk_prof_end_(PROF_ATTN, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_attn_context_fwd_f32(const float *attn, const float *value, float *ctx, int BN,
                                         int T, int Dv, int64_t ctx_stride, void *stream) {
    if (BN < 0 || T <= 0 || Dv <= 0 || ctx_stride < Dv) return ASRK_EINVAL;
    if (BN == 0) return ASRK_OK;
    if (!attn || !value || !ctx) return ASRK_EINVAL;
    if ((size_t)T * 4 > 60 * 1024) return ASRK_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_ATTN, s);
    hipLaunchKernelGGL(context_fwd_kernel, dim3(BN, asrk_div_up(Dv, 256)), dim3(256), (size_t)T * 4, s,
                       attn, value, ctx, T, Dv, ctx_stride);
    asrk_prof_end_(PROF_ATTN, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_attn_context_bwd_f32(const float *dctx, const float *value, float *dattn, int BN,
                                         int T, int Dv, int64_t dctx_stride, void *stream) {
    if (BN < 0 || T <= 0 || Dv <= 0 || dctx_stride < Dv) return ASRK_EINVAL;
    if (BN == 0) return ASRK_OK;
    if (!dctx || !value || !dattn) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_ATTN, s);
    hipLaunchKernelGGL(context_bwd_attn_kernel, dim3(BN, asrk_div_up(T, 4)), dim3(256), 0, s, dctx, value,
                       dattn, T, Dv, dctx_stride);
    asrk_prof_end_(PROF_ATTN, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_lstm_cell_fwd_f32(float *gates, const float *c_prev, float *c, float *h, int B, int H,
                                      void *stream) {
    if (B < 0 || H <= 0) return ASRK_EINVAL;
    if (B == 0) return ASRK_OK;
    if (!gates || !c_prev || !c || !h) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_CELL, s);
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(blocks_for((int64_t)B * H, 256)), dim3(256), 0, s, gates,
                       c_prev, c, h, B, H);
    asrk_prof_end_(PROF_CELL, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_lstm_cell_bwd_f32(float *gates, const float *c_prev, const float *c, const float *dh,
                                      const float *dc_in, float *dc_prev, int B, int H, void *stream) {
    if (B < 0 || H <= 0) return ASRK_EINVAL;
    if (B == 0) return ASRK_OK;
    if (!gates || !c_prev || !c || !dc_prev) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_CELL, s);
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(blocks_for((int64_t)B * H, 256)), dim3(256), 0, s, gates,
                       c_prev, c, dh, dc_in, dc_prev, B, H);
    asrk_prof_end_(PROF_CELL, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_embedding_fwd_f32(const int64_t *idx, const float *W, float *out, int64_t n, int D,
                                      int V, void *stream) {
    if (n < 0 || D <= 0 || V <= 0) return ASRK_EINVAL;
    if (n == 0) return ASRK_OK;
    if (!idx || !W || !out) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(blocks_for(n * D, 256)), dim3(256), 0, s, idx, W, out, n,
                       D, V);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_embedding_bwd_f32(const int64_t *idx, const float *dout, float *dW_acc, int64_t n,
                                      int D, int V, void *stream) {
    if (n < 0 || D <= 0 || V <= 0) return ASRK_EINVAL;
    if (n == 0) return ASRK_OK;
    if (!idx || !dout || !dW_acc) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (asrk_knobs_().get(asrk_knobs_().deterministic, 0))
        hipLaunchKernelGGL(embedding_bwd_ordered_kernel, dim3(blocks_for(D, 64)), dim3(64), 0, s, idx, dout, dW_acc,
                           n, D, V);
    else
        hipLaunchKernelGGL(embedding_bwd_kernel, dim3(blocks_for(n * D, 256)), dim3(256), 0, s, idx, dout,
                           dW_acc, n, D, V);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}
