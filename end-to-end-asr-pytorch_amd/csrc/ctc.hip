// CTC loss forward (alpha) and backward (beta + gradient) for gfx950.
//
// Replaces torch.nn.CTCLoss(blank=0, zero_infinity=False) as called at
// bin/train_asr.py:49,123-124 (ATen ctc_loss / ctc_loss_backward): log-space alpha/beta over the
// blank-extended label sequence (S = 2L+1), nll_b = -logsumexp(alpha[T_b-1][S-1], alpha[..][S-2]),
// grad = (exp(lp) - exp(logsum_{s:ext[s]=c}(alpha+beta) + nll - lp)) * scale.
//
// One workgroup per utterance; lanes own contiguous runs of lattice states; the previous lattice
// row lives in LDS (double-buffered), the log-prob gathers for step t+1 are issued before the
// log-sum-exp of step t (only 2L+1 of the V log-probs per frame are ever read: 3.3 MB of the
// 128 MB tensor at cfg3).  The dense gradient write is a separate streaming pass (HBM-bound:
// read lp + write grad once), followed by a sparse fix-up of the <= L+1 label columns per frame.
#include "common.h"

namespace {

constexpr int CTC_THREADS = 64;
constexpr int CTC_MAX_SPL = 16;  // states per lane -> S <= 1024 (L <= 511)

__device__ __forceinline__ float lse3(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c));
    if (m == -INFINITY) return -INFINITY;
    return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

struct CtcArgs {
    const float *lp;
    int64_t st, sb;
    int T, B, V;
    const int64_t *targets;
    int64_t tgt_stride;
    int Lmax;
    const int64_t *in_len, *tg_len;
    int blank;
    float *alpha;  // [B, T, S]
    float *beta;   // [B, T, S]
    float *nll;    // [B]
};

// ext[s]: blank for even s, target[(s-1)/2] for odd s
__device__ __forceinline__ int ext_label(const int64_t *tgt, int s, int blank) {
    return (s & 1) ? (int)tgt[s >> 1] : blank;
}

template <bool BACKWARD>
__global__ __launch_bounds__(CTC_THREADS) void ctc_lattice_kernel(CtcArgs p) {
    extern __shared__ float srow[];  // [2][Smax]
    const int b = blockIdx.x, lane = threadIdx.x;
    const int Smax = 2 * p.Lmax + 1;
    const int spl = (Smax + CTC_THREADS - 1) / CTC_THREADS;
    int Tb = (int)p.in_len[b];
    int tl = (int)p.tg_len[b];
    if (Tb > p.T) Tb = p.T;
    if (tl > p.Lmax) tl = p.Lmax;
    const int S = 2 * tl + 1;
    const int64_t *tgt = p.targets + (int64_t)b * p.tgt_stride;
    float *out = (BACKWARD ? p.beta : p.alpha) + (size_t)b * p.T * Smax;
    const float *lpb = p.lp + (int64_t)b * p.sb;

    // per-lane static state info
    int lab[CTC_MAX_SPL];
    bool skip_ok[CTC_MAX_SPL];  // may take the s-2 (fwd) / s+2 (bwd) transition
#pragma unroll
    for (int j = 0; j < CTC_MAX_SPL; ++j) {
        lab[j] = p.blank;
        skip_ok[j] = false;
        if (j < spl) {
            const int s = lane * spl + j;
            if (s < S) {
                lab[j] = ext_label(tgt, s, p.blank);
                if (!BACKWARD) {
                    if (s >= 2 && (s & 1)) skip_ok[j] = lab[j] != ext_label(tgt, s - 2, p.blank);
                } else {
                    if (s + 2 < S && (s & 1)) skip_ok[j] = lab[j] != ext_label(tgt, s + 2, p.blank);
                }
            }
        }
    }

    if (Tb <= 0) {
        if (!BACKWARD && lane == 0) p.nll[b] = (tl == 0) ? 0.f : INFINITY;
        return;
    }

    const int t_first = BACKWARD ? Tb - 1 : 0;
    const int dt = BACKWARD ? -1 : 1;

    // gather log-probs of the first frame
    float lpc[CTC_MAX_SPL];
#pragma unroll
    for (int j = 0; j < CTC_MAX_SPL; ++j) {
        lpc[j] = 0.f;
        if (j < spl && lane * spl + j < S) lpc[j] = lpb[(int64_t)t_first * p.st + lab[j]];
    }

    float cur[CTC_MAX_SPL];
#pragma unroll
    for (int j = 0; j < CTC_MAX_SPL; ++j) {
        cur[j] = -INFINITY;
        if (j < spl) {
            const int s = lane * spl + j;
            if (s < S) {
                const bool init = BACKWARD ? (s >= S - 2) : (s <= 1);
                if (init) cur[j] = lpc[j];
            }
        }
    }

    for (int step = 0; step < Tb; ++step) {
        const int t = t_first + dt * step;
        float *row = srow + (step & 1) * Smax;
        // publish row t
#pragma unroll
        for (int j = 0; j < CTC_MAX_SPL; ++j) {
            if (j < spl) {
                const int s = lane * spl + j;
                if (s < Smax) {
                    row[s] = cur[j];
                    out[(size_t)t * Smax + s] = cur[j];
                }
            }
        }
        if (step + 1 == Tb) break;
        const int tn = t + dt;
        // prefetch the gathers of the next frame before waiting on the LDS row
#pragma unroll
        for (int j = 0; j < CTC_MAX_SPL; ++j) {
            if (j < spl && lane * spl + j < S) lpc[j] = lpb[(int64_t)tn * p.st + lab[j]];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CTC_MAX_SPL; ++j) {
            if (j < spl) {
                const int s = lane * spl + j;
                if (s < S) {
                    float a0 = row[s], a1 = -INFINITY, a2 = -INFINITY;
                    if (!BACKWARD) {
                        if (s >= 1) a1 = row[s - 1];
                        if (skip_ok[j]) a2 = row[s - 2];
                    } else {
                        if (s + 1 < S) a1 = row[s + 1];
                        if (skip_ok[j]) a2 = row[s + 2];
                    }
                    cur[j] = lse3(a0, a1, a2) + lpc[j];
                } else {
                    cur[j] = -INFINITY;
                }
            }
        }
    }

    // rows beyond the utterance length are never read by the gradient kernels
    if (!BACKWARD) {
        __syncthreads();
        if (lane == 0) {
            const float *row = srow + ((Tb - 1) & 1) * Smax;
            const float l1 = row[S - 1];
            const float l2 = S >= 2 ? row[S - 2] : -INFINITY;
            p.nll[b] = -log_add(l1, l2);
        }
    }
}

// dense part: grad = exp(lp) * scale (t < T_b) or 0
__global__ __launch_bounds__(256) void ctc_grad_dense_kernel(const float *__restrict__ lp,
                                                             int64_t st, int64_t sb, int T, int B,
                                                             int V, const int64_t *in_len,
                                                             const float *gscale,
                                                             float *__restrict__ grad, int64_t gst,
                                                             int64_t gsb) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);  // row = t*B + b
    const int lane = threadIdx.x & 63;
    if (row >= T * B) return;
    const int t = row / B, b = row - t * B;
    const float *x = lp + (int64_t)t * st + (int64_t)b * sb;
    float *g = grad + (int64_t)t * gst + (int64_t)b * gsb;
    const bool live = t < (int)in_len[b];
    const float sc = gscale[b];
    const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g)) & 15) == 0 &&
                     (V % 4 == 0);
    if (vec) {
        const int nv = V >> 2;
        for (int i = lane; i < nv; i += 64) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            if (live) {
                const f32x4 v = reinterpret_cast<const f32x4 *>(x)[i];
                o[0] = expf(v[0]) * sc; o[1] = expf(v[1]) * sc;
                o[2] = expf(v[2]) * sc; o[3] = expf(v[3]) * sc;
            }
            reinterpret_cast<f32x4 *>(g)[i] = o;
        }
    } else {
        for (int i = lane; i < V; i += 64) g[i] = live ? expf(x[i]) * sc : 0.f;
    }
}

// sparse part: for the label columns, subtract exp(lcab + nll - lp) * scale
struct CtcFixArgs {
    const float *lp;
    int64_t st, sb;
    int T, B;
    const int64_t *targets;
    int64_t tgt_stride;
    int Lmax;
    const int64_t *in_len, *tg_len;
    int blank;
    const float *alpha, *beta, *nll, *gscale;
    float *grad;
    int64_t gst, gsb;
    int t_per_block;
};

__global__ __launch_bounds__(256) void ctc_grad_fix_kernel(CtcFixArgs p) {
    extern __shared__ int sm_i[];
    const int b = blockIdx.x;
    const int Smax = 2 * p.Lmax + 1;
    int *ext = sm_i;                 // [Smax]
    int *nxt = sm_i + Smax;          // [Smax] next state with the same label (or -1)
    int *leader = sm_i + 2 * Smax;   // [Smax] 1 if first occurrence
    float *ab = reinterpret_cast<float *>(sm_i + 3 * Smax);  // [Smax]
    int Tb = (int)p.in_len[b];
    int tl = (int)p.tg_len[b];
    if (Tb > p.T) Tb = p.T;
    if (tl > p.Lmax) tl = p.Lmax;
    const int S = 2 * tl + 1;
    const int64_t *tgt = p.targets + (int64_t)b * p.tgt_stride;
    for (int s = threadIdx.x; s < S; s += blockDim.x) ext[s] = ext_label(tgt, s, p.blank);
    __syncthreads();
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        const int c = ext[s];
        int n = -1;
        for (int s2 = s + 1; s2 < S; ++s2)
            if (ext[s2] == c) { n = s2; break; }
        nxt[s] = n;
        int first = 1;
        for (int s2 = 0; s2 < s; ++s2)
            if (ext[s2] == c) { first = 0; break; }
        leader[s] = first;
    }
    __syncthreads();
    const float nll = p.nll[b], sc = p.gscale[b];
    const int t0 = blockIdx.y * p.t_per_block;
    const int t1 = min(Tb, t0 + p.t_per_block);
    const float *al = p.alpha + (size_t)b * p.T * Smax;
    const float *be = p.beta + (size_t)b * p.T * Smax;
    for (int t = t0; t < t1; ++t) {
        for (int s = threadIdx.x; s < S; s += blockDim.x)
            ab[s] = al[(size_t)t * Smax + s] + be[(size_t)t * Smax + s];
        __syncthreads();
        for (int s = threadIdx.x; s < S; s += blockDim.x) {
            if (leader[s]) {
                float acc = ab[s];
                for (int n = nxt[s]; n >= 0; n = nxt[n]) acc = log_add(acc, ab[n]);
                const int c = ext[s];
                const float x = p.lp[(int64_t)t * p.st + (int64_t)b * p.sb + c];
                float *g = p.grad + (int64_t)t * p.gst + (int64_t)b * p.gsb + c;
                *g -= expf(acc + nll - x) * sc;
            }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int asrk_ctc_loss_fwd_f32(const float *lp, int64_t stride_t, int64_t stride_b, int T,
                                     int B, int V, const int64_t *targets, int64_t tgt_stride,
                                     int Lmax, const int64_t *input_lengths,
                                     const int64_t *target_lengths, int blank, float *alpha,
                                     float *nll, void *stream) {
    if (T < 0 || B < 0 || V <= 0 || Lmax < 0 || blank < 0 || blank >= V) return ASRK_EINVAL;
    if (B == 0) return ASRK_OK;
    if (!lp || !input_lengths || !target_lengths || !alpha || !nll) return ASRK_EINVAL;
    if (Lmax > 0 && !targets) return ASRK_EINVAL;
    const int Smax = 2 * Lmax + 1;
    if (Smax > CTC_THREADS * CTC_MAX_SPL) return ASRK_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    CtcArgs a{lp, stride_t, stride_b, T, B, V, targets, tgt_stride, Lmax, input_lengths,
              target_lengths, blank, alpha, nullptr, nll};
    asrk_prof_begin_(PROF_CTC, s);
    hipLaunchKernelGGL((ctc_lattice_kernel<false>), dim3(B), dim3(CTC_THREADS),
                       2 * Smax * sizeof(float), s, a);
    asrk_prof_end_(PROF_CTC, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_ctc_loss_bwd_f32(const float *lp, int64_t stride_t, int64_t stride_b, int T,
                                     int B, int V, const int64_t *targets, int64_t tgt_stride,
                                     int Lmax, const int64_t *input_lengths,
                                     const int64_t *target_lengths, int blank, const float *alpha,
                                     float *beta, const float *nll, const float *gscale,
                                     float *grad, int64_t g_stride_t, int64_t g_stride_b,
                                     void *stream) {
    if (T < 0 || B < 0 || V <= 0 || Lmax < 0 || blank < 0 || blank >= V) return ASRK_EINVAL;
    if (B == 0 || T == 0) return ASRK_OK;
    if (!lp || !input_lengths || !target_lengths || !alpha || !beta || !nll || !gscale || !grad)
        return ASRK_EINVAL;
    if (Lmax > 0 && !targets) return ASRK_EINVAL;
    const int Smax = 2 * Lmax + 1;
    if (Smax > CTC_THREADS * CTC_MAX_SPL) return ASRK_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_CTC, s);
    CtcArgs a{lp, stride_t, stride_b, T, B, V, targets, tgt_stride, Lmax, input_lengths,
              target_lengths, blank, const_cast<float *>(alpha), beta, const_cast<float *>(nll)};
    hipLaunchKernelGGL((ctc_lattice_kernel<true>), dim3(B), dim3(CTC_THREADS),
                       2 * Smax * sizeof(float), s, a);
    hipLaunchKernelGGL(ctc_grad_dense_kernel, dim3(asrk_div_up(T * B, 4)), dim3(256), 0, s, lp,
                       stride_t, stride_b, T, B, V, input_lengths, gscale, grad, g_stride_t,
                       g_stride_b);
    int tchunks = asrk_div_up(1024, B);
    if (tchunks > T) tchunks = T;
    if (tchunks < 1) tchunks = 1;
    const int tpb = asrk_div_up(T, tchunks);
    tchunks = asrk_div_up(T, tpb);
    CtcFixArgs f{lp, stride_t, stride_b, T, B, targets, tgt_stride, Lmax, input_lengths,
                 target_lengths, blank, alpha, beta, nll, gscale, grad, g_stride_t, g_stride_b,
                 tpb};
    hipLaunchKernelGGL(ctc_grad_fix_kernel, dim3(B, tchunks), dim3(256),
                       (size_t)Smax * (3 * sizeof(int) + sizeof(float)), s, f);
    asrk_prof_end_(PROF_CTC, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

// ---------------------------------------------------------------------------------------------
// CTC prefix scoring for joint CTC-attention beam search (reference: src/ctc.py:76-116,
// CTCPrefixScore.cheap_compute — numpy on the host, one hypothesis at a time).  Here ALL
// (hypothesis, candidate) pairs of a beam step run in one launch: one lane per pair walks the
// T' frames with its two running log-probabilities in registers.
namespace {

__device__ __forceinline__ float lae(float a, float b) {   // np.logaddexp for finite inputs
    const float m = fmaxf(a, b);
    return m + log1pf(expf(-fabsf(a - b)));
}

struct PrefixArgs {
    const float *x;        // [T, V] log-probs of the utterance
    const float *r_prev;   // [n, T, 2] previous state of each hypothesis (0 = non-blank, 1 = blank)
    const int *plen;       // [n] prefix length |g|
    const int *last;       // [n] last token of the prefix (ignored when plen == 0)
    const int *cand;       // [n, C]
    float *psi;            // [n, C]
    float *r_out;          // [n, C, T, 2]
    int n, C, T, V, blank, eos;
    float logzero;
};

__global__ void ctc_prefix_kernel(PrefixArgs p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n * p.C) return;
    const int h = i / p.C;
    const int c = p.cand[i];
    const int T = p.T, V = p.V;
    const float *rp = p.r_prev + (size_t)h * T * 2;
    float *ro = p.r_out + (size_t)i * T * 2;
    const int plen = p.plen[h];
    const bool same = plen > 0 && c == p.last[h];   // phi uses only the blank path (ctc.py:97-99)
    const int start = plen > 1 ? plen : 1;
    for (int t = 0; t < T; ++t) {
        ro[2 * t] = p.logzero;
        ro[2 * t + 1] = p.logzero;
    }
    if (plen == 0) ro[0] = p.x[c];                  // r[0,0] = x[0, c] if g = <sos>
    float rn = (start - 1 < T) ? ro[2 * (start - 1)] : p.logzero;       // r[t-1, 0]
    float rb = (start - 1 < T) ? ro[2 * (start - 1) + 1] : p.logzero;   // r[t-1, 1]
    float psi = rn;
    for (int t = start; t < T; ++t) {
        const float phi = same ? rp[2 * (t - 1) + 1] : lae(rp[2 * (t - 1)], rp[2 * (t - 1) + 1]);
        const float xc = p.x[(size_t)t * V + c], xb = p.x[(size_t)t * V + p.blank];
        const float nn = lae(rn, phi) + xc;
        const float nb = lae(rb, rn) + xb;
        psi = lae(psi, phi + xc);
        rn = nn;
        rb = nb;
        ro[2 * t] = rn;
        ro[2 * t + 1] = rb;
    }
    if (c == p.eos) psi = lae(rp[2 * (T - 1)], rp[2 * (T - 1) + 1]);   // P(<eos>) = P(g)
    p.psi[i] = psi;
}

}  // namespace

extern "C" int asrk_ctc_prefix_score_f32(const float *x, const float *r_prev, const int *prefix_len,
                                         const int *last_char, const int *candidates, float *psi,
                                         float *r_out, int n, int C, int T, int V, int blank, int eos,
                                         float logzero, void *stream) {
    if (n < 0 || C < 0 || T <= 0 || V <= 0) return ASRK_EINVAL;
    if (n == 0 || C == 0) return ASRK_OK;
    if (!x || !r_prev || !prefix_len || !last_char || !candidates || !psi || !r_out) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    PrefixArgs a{x, r_prev, prefix_len, last_char, candidates, psi, r_out, n, C, T, V, blank, eos, logzero};
    asrk_prof_begin_(PROF_CTC, s);
    hipLaunchKernelGGL(ctc_prefix_kernel, dim3(asrk_div_up(n * C, 64)), dim3(64), 0, s, a);
    asrk_prof_end_(PROF_CTC, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}
