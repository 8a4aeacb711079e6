// CTC loss forward (alpha) and backward (beta + gradient) for gfx950.
//
// Replaces torch.nn.CTCLoss(blank=0, zero_infinity=False) as called at
// bin/train_asr.py:49,123-124 (ATen ctc_loss / ctc_loss_backward): log-space alpha/beta over the
// blank-extended label sequence (S = 2L+1), nll_b = -logsumexp(alpha[T_b-1][S-1], alpha[..][S-2]),
// grad = (exp(lp) - exp(logsum_{s:ext[s]=c}(alpha+beta) + nll - lp)) * scale.
//
// Only 2L+1 of the V log-probs per frame are ever read by the lattices (3.3 MB of the 128 MB tensor
// at cfg3): a gather pre-pass compacts them to [B,T,S]; then alpha and beta run CONCURRENTLY in one
// launch (one wave per utterance and direction, previous lattice row in LDS, the next 8 frames'
// gathered rows already in registers, so no memory latency sits on the T-step dependent chain).
// The dense gradient write is a separate streaming pass (HBM-bound: read lp + write grad once),
// followed by a sparse overwrite of the <= L+1 label columns per frame.
#include "common.h"

namespace {

constexpr int CTC_THREADS = 64;
constexpr int CTC_MAX_SPL = 32;  // states per lane -> S <= 2048 (L <= 1023)

template <bool FAST>
__device__ __forceinline__ float lse3(float a, float b, float c) {
    const float m = fmaxf(a, fmaxf(b, c));
    if (m == -INFINITY) return -INFINITY;
    // long lattices (T > 512 or more than 256 states) use the library functions: the ~1e-7 per-step
    // error of the hardware approximations accumulates to ~1e-3 in alpha + beta after ~1000 frames
    if (!FAST) return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
    // hardware exp2/log2 (v_exp_f32 / v_log_f32, ~1 ulp): the library expf/logf are ~15 instructions
    // each and this is the dependent chain of the lattice (one wave per lattice: the step was
    // ALU-bound at ~2k cycles).  The largest term is exp(0) = 1 exactly, the others only matter
    // when they are close to it, so the absolute error per step stays ~1e-7.
    return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
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
    float *beta;   // [B, T, S]  (nullptr: alpha only)
    float *lpg;    // [B, T, S]  gathered log-probs lp[t, b, ext_b[s]]
    float *nll;    // [B]
};

// ext[s]: blank for even s, target[(s-1)/2] for odd s
__device__ __forceinline__ int ext_label(const int64_t *tgt, int s, int blank) {
    return (s & 1) ? (int)tgt[s >> 1] : blank;
}

// K0: lpg[b][t][s] = lp[t, b, ext_b[s]] for t < T_b, s < S_b.  Fully parallel scattered gather; every
// later CTC kernel reads these compact rows (coalesced, L2-resident) instead of the [T,B,V] tensor.
constexpr int GATHER_ROWS = 8;
__global__ __launch_bounds__(256) void ctc_gather_kernel(CtcArgs p) {
    const int b = blockIdx.x;
    const int Smax = 2 * p.Lmax + 1;
    int Tb = min((int)p.in_len[b], p.T);
    int tl = min((int)p.tg_len[b], p.Lmax);
    const int S = 2 * tl + 1;
    const int64_t *tgt = p.targets + (int64_t)b * p.tgt_stride;
    const int t0 = blockIdx.y * GATHER_ROWS;
    const float *lpb = p.lp + (int64_t)b * p.sb;
    float *out = p.lpg + (size_t)b * p.T * Smax;
    for (int idx = threadIdx.x; idx < GATHER_ROWS * Smax; idx += 256) {
        const int r = idx / Smax, sidx = idx - r * Smax;
        const int t = t0 + r;
        if (t < Tb && sidx < S)
            // a label outside [0,V) never reads out of bounds; the lattice kernel reports it (NaN loss)
            out[(size_t)t * Smax + sidx] =
                lpb[(int64_t)t * p.st + min(max(ext_label(tgt, sidx, p.blank), 0), p.V - 1)];
    }
}

// K1: alpha (blockIdx.y == 0) and beta (blockIdx.y == 1) lattices of utterance blockIdx.x, one wave
// each.  The gathered log-prob rows of the next PF frames are kept in registers (ring refilled PF
// steps ahead), so a step is: publish row -> barrier -> 3-way log-sum-exp; no memory latency on
// the dependent chain.
constexpr int CTC_PF = 8;
template <int SPL, bool FAST>
__global__ __launch_bounds__(CTC_THREADS) void ctc_lattice_kernel(CtcArgs p) {
    extern __shared__ float srow[];  // [2][Smax]
    const int b = blockIdx.x, lane = threadIdx.x;
    const bool bw = blockIdx.y == 1;
    const int Smax = 2 * p.Lmax + 1;
    const int spl = (Smax + CTC_THREADS - 1) / CTC_THREADS;   // <= SPL
    int Tb = min((int)p.in_len[b], p.T);
    int tl = min((int)p.tg_len[b], p.Lmax);
    const int S = 2 * tl + 1;
    const int64_t *tgt = p.targets + (int64_t)b * p.tgt_stride;
    float *out = (bw ? p.beta : p.alpha) + (size_t)b * p.T * Smax;
    const float *g = p.lpg + (size_t)b * p.T * Smax;

    bool skip_ok[SPL];  // may take the s-2 (alpha) / s+2 (beta) transition
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        skip_ok[j] = false;
        const int s = lane * spl + j;
        if (j < spl && s < S && (s & 1)) {
            const int lab = ext_label(tgt, s, p.blank);
            if (!bw) {
                if (s >= 2) skip_ok[j] = lab != ext_label(tgt, s - 2, p.blank);
            } else {
                if (s + 2 < S) skip_ok[j] = lab != ext_label(tgt, s + 2, p.blank);
            }
        }
    }

    if (Tb <= 0) {
        if (!bw && lane == 0) p.nll[b] = (tl == 0) ? 0.f : INFINITY;
        return;
    }
    {   // torch.nn.CTCLoss rejects targets outside [0,V); here the utterance's loss becomes NaN (no
        // host sync), which the solver's NaN-gradient guard (src/solver.py:85-89) then skips
        bool bad = false;
        for (int i = lane; i < tl; i += CTC_THREADS) bad |= tgt[i] < 0 || tgt[i] >= p.V;
        if (__any(bad)) {
            if (!bw && lane == 0) p.nll[b] = __builtin_nanf("");
            return;
        }
    }
    const int t_first = bw ? Tb - 1 : 0;
    const int dt = bw ? -1 : 1;

    float cur[SPL], pre[CTC_PF][SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        cur[j] = -INFINITY;
        const int s = lane * spl + j;
        if (j < spl && s < S) {
            const bool init = bw ? (s >= S - 2) : (s <= 1);
            if (init) cur[j] = g[(size_t)t_first * Smax + s];
        }
    }
#pragma unroll
    for (int d = 0; d < CTC_PF; ++d)
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            pre[d][j] = 0.f;
            const int s = lane * spl + j;
            if (j < spl && s < S && 1 + d < Tb) pre[d][j] = g[(size_t)(t_first + dt * (1 + d)) * Smax + s];
        }

    for (int base = 0; base < Tb; base += CTC_PF) {
#pragma unroll
        for (int d = 0; d < CTC_PF; ++d) {
            const int step = base + d;
            if (step >= Tb) break;
            const int t = t_first + dt * step;
            float *row = srow + (step & 1) * Smax;
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
                const int s = lane * spl + j;
                if (j < spl && s < Smax) {
                    row[s] = cur[j];
                    out[(size_t)t * Smax + s] = cur[j];
                }
            }
            if (step + 1 == Tb) break;
            float lpn[SPL];
            const int sp = step + 1 + CTC_PF;
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
                lpn[j] = pre[d][j];
                const int s = lane * spl + j;
                if (j < spl && s < S && sp < Tb) pre[d][j] = g[(size_t)(t_first + dt * sp) * Smax + s];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
                const int s = lane * spl + j;
                if (j < spl) {
                    if (s < S) {
                        float a0 = row[s], a1 = -INFINITY, a2 = -INFINITY;
                        if (!bw) {
                            if (s >= 1) a1 = row[s - 1];
                            if (skip_ok[j]) a2 = row[s - 2];
                        } else {
                            if (s + 1 < S) a1 = row[s + 1];
                            if (skip_ok[j]) a2 = row[s + 2];
                        }
                        cur[j] = lse3<FAST>(a0, a1, a2) + lpn[j];
                    } else {
                        cur[j] = -INFINITY;
                    }
                }
            }
        }
    }

    // rows beyond the utterance length are never read by the gradient kernels
    if (!bw) {
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
    const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // row = t*B + b
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

// sparse part: the label columns are OVERWRITTEN with (exp(lp) - exp(lcab + nll - lp)) * scale
// (runs after the dense pass; no read-modify-write of the gradient)
struct CtcFixArgs {
    int T, B;
    const int64_t *targets;
    int64_t tgt_stride;
    int Lmax;
    const int64_t *in_len, *tg_len;
    int blank;
    const float *alpha, *beta, *lpg, *nll, *gscale;
    float *grad;
    int64_t gst, gsb;
    int t_per_block;
};

__global__ __launch_bounds__(256) void ctc_grad_fix_kernel(CtcFixArgs p) {
    extern __shared__ int sm_i[];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Smax = 2 * p.Lmax + 1;
    int *ext = sm_i;                 // [Smax]
    int *nxt = sm_i + Smax;          // [Smax] next state with the same label (or -1)
    int *leader = sm_i + 2 * Smax;   // [Smax] 1 if first occurrence
    float *ab = reinterpret_cast<float *>(sm_i + 3 * Smax) + wave * Smax;  // [4 waves][Smax]
    int Tb = min((int)p.in_len[b], p.T);
    int tl = min((int)p.tg_len[b], p.Lmax);
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
    const float *lg = p.lpg + (size_t)b * p.T * Smax;
    for (int tb = t0; tb < t1; tb += 4) {       // 4 frames per iteration, one per wave
        const int t = tb + wave;
        const bool live = t < t1;
        if (live)
            for (int s = lane; s < S; s += 64) ab[s] = al[(size_t)t * Smax + s] + be[(size_t)t * Smax + s];
        __syncthreads();
        if (live) {
            // the blank sits on every even state (tl+1 of them): the whole wave reduces that column
            // as max + sum-of-exp instead of lane 0 walking a tl-long log_add chain per frame
            float m = -INFINITY;
            for (int s = lane; s < S; s += 64)
                if (ext[s] == p.blank) m = fmaxf(m, ab[s]);   // even states (+ a label equal to blank)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            float sum = 0.f;
            if (m > -INFINITY)
                for (int s = lane; s < S; s += 64)
                    if (ext[s] == p.blank) sum += expf(ab[s] - m);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
            if (lane == 0) {
                const float acc = m > -INFINITY ? m + logf(sum) : -INFINITY;
                const float x = lg[(size_t)t * Smax];
                p.grad[(int64_t)t * p.gst + (int64_t)b * p.gsb + p.blank] =
                    (expf(x) - expf(acc + nll - x)) * sc;
            }
            for (int s = 1 + 2 * lane; s < S; s += 128) {   // label states (odd), first occurrences
                if (leader[s] && ext[s] != p.blank) {
                    float acc = ab[s];
                    for (int n = nxt[s]; n >= 0; n = nxt[n]) acc = log_add(acc, ab[n]);
                    const float x = lg[(size_t)t * Smax + s];
                    p.grad[(int64_t)t * p.gst + (int64_t)b * p.gsb + ext[s]] =
                        (expf(x) - expf(acc + nll - x)) * sc;
                }
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
                                     float *beta, float *lpg, float *nll, void *stream) {
    if (T < 0 || B < 0 || V <= 0 || Lmax < 0 || blank < 0 || blank >= V) return ASRK_EINVAL;
    if (B == 0) return ASRK_OK;
    if (!lp || !input_lengths || !target_lengths || !alpha || !lpg || !nll) return ASRK_EINVAL;
    if (Lmax > 0 && !targets) return ASRK_EINVAL;
    const int Smax = 2 * Lmax + 1;
    if (Smax > CTC_THREADS * CTC_MAX_SPL) return ASRK_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    CtcArgs a{lp, stride_t, stride_b, T, B, V, targets, tgt_stride, Lmax, input_lengths,
              target_lengths, blank, alpha, beta, lpg, nll};
    asrk_prof_begin_(PROF_CTC, s);
    if (T > 0)
        hipLaunchKernelGGL(ctc_gather_kernel, dim3(B, asrk_div_up(T, GATHER_ROWS)), dim3(256), 0, s, a);
    const dim3 grid(B, beta ? 2 : 1);
    if (Smax <= CTC_THREADS * 4 && T <= 512)
        hipLaunchKernelGGL((ctc_lattice_kernel<4, true>), grid, dim3(CTC_THREADS), 2 * Smax * sizeof(float), s, a);
    else if (Smax <= CTC_THREADS * 4)
        hipLaunchKernelGGL((ctc_lattice_kernel<4, false>), grid, dim3(CTC_THREADS), 2 * Smax * sizeof(float), s, a);
    else if (Smax <= CTC_THREADS * 16)
        hipLaunchKernelGGL((ctc_lattice_kernel<16, false>), grid, dim3(CTC_THREADS), 2 * Smax * sizeof(float), s, a);
    else   // character-level transcripts of the longest LibriSpeech utterances (L up to 1023)
        hipLaunchKernelGGL((ctc_lattice_kernel<CTC_MAX_SPL, false>), grid, dim3(CTC_THREADS),
                           2 * Smax * sizeof(float), s, a);
    asrk_prof_end_(PROF_CTC, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_ctc_loss_bwd_f32(const float *lp, int64_t stride_t, int64_t stride_b, int T,
                                     int B, int V, const int64_t *targets, int64_t tgt_stride,
                                     int Lmax, const int64_t *input_lengths,
                                     const int64_t *target_lengths, int blank, const float *alpha,
                                     const float *beta, const float *lpg, const float *nll,
                                     const float *gscale, float *grad, int64_t g_stride_t,
                                     int64_t g_stride_b, void *stream) {
    if (T < 0 || B < 0 || V <= 0 || Lmax < 0 || blank < 0 || blank >= V) return ASRK_EINVAL;
    if (B == 0 || T == 0) return ASRK_OK;
    if (!lp || !input_lengths || !target_lengths || !alpha || !beta || !lpg || !nll || !gscale || !grad)
        return ASRK_EINVAL;
    if (Lmax > 0 && !targets) return ASRK_EINVAL;
    const int Smax = 2 * Lmax + 1;
    if (Smax > CTC_THREADS * CTC_MAX_SPL) return ASRK_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    // dense gradient: every log-prob read, every gradient element written (the lattice itself is a few MB)
    asrk_prof_work_(PROF_CTC, 8.0 * (double)T * (double)B * (double)V);
    asrk_prof_begin_(PROF_CTC, s);
    hipLaunchKernelGGL(ctc_grad_dense_kernel, dim3(asrk_div_up(T * B, 4)), dim3(256), 0, s, lp,
                       stride_t, stride_b, T, B, V, input_lengths, gscale, grad, g_stride_t,
                       g_stride_b);
    int tchunks = asrk_div_up(512, B);
    if (tchunks > asrk_div_up(T, 4)) tchunks = asrk_div_up(T, 4);
    if (tchunks < 1) tchunks = 1;
    const int tpb = asrk_div_up(asrk_div_up(T, tchunks), 4) * 4;
    tchunks = asrk_div_up(T, tpb);
    CtcFixArgs f{T, B, targets, tgt_stride, Lmax, input_lengths, target_lengths, blank, alpha, beta,
                 lpg, nll, gscale, grad, g_stride_t, g_stride_b, tpb};
    hipLaunchKernelGGL(ctc_grad_fix_kernel, dim3(B, tchunks), dim3(256),
                       (size_t)Smax * (3 * sizeof(int) + 4 * sizeof(float)), s, f);
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
    const int *row_mem;    // optional [n]: hypothesis h scores against utterance row_mem[h] of x [U, T, V]
    const int *mem_len;    // optional [U]: frames of every utterance (<= T, the common stride)
};

__global__ void ctc_prefix_kernel(PrefixArgs p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n * p.C) return;
    const int h = i / p.C;
    const int c = p.cand[i];
    const int TS = p.T, V = p.V;                       // TS: stride of the state arrays
    const int u = p.row_mem ? p.row_mem[h] : 0;
    const int T = p.mem_len ? min(p.mem_len[u], TS) : TS;   // frames of this hypothesis' utterance
    const float *px = p.x + (size_t)u * TS * V;
    const float *rp = p.r_prev + (size_t)h * TS * 2;
    float *ro = p.r_out + (size_t)i * TS * 2;
    const int plen = p.plen[h];
    const bool same = plen > 0 && c == p.last[h];   // phi uses only the blank path (ctc.py:97-99)
    const int start = plen > 1 ? plen : 1;
    for (int t = 0; t < TS; ++t) {
        ro[2 * t] = p.logzero;
        ro[2 * t + 1] = p.logzero;
    }
    if (plen == 0) ro[0] = px[c];                   // r[0,0] = x[0, c] if g = <sos>
    float rn = (start - 1 < T) ? ro[2 * (start - 1)] : p.logzero;       // r[t-1, 0]
    float rb = (start - 1 < T) ? ro[2 * (start - 1) + 1] : p.logzero;   // r[t-1, 1]
    float psi = rn;
    for (int t = start; t < T; ++t) {
        const float phi = same ? rp[2 * (t - 1) + 1] : lae(rp[2 * (t - 1)], rp[2 * (t - 1) + 1]);
        const float xc = px[(size_t)t * V + c], xb = px[(size_t)t * V + p.blank];
        const float nn = lae(rn, phi) + xc;
        const float nb = lae(rb, rn) + xb;
        psi = lae(psi, phi + xc);
        rn = nn;
        rb = nb;
        ro[2 * t] = rn;
        ro[2 * t + 1] = rb;
    }
    // P(<eos>) = P(g); an utterance with no frames left after the encoder's time reduction has no path at all
    if (c == p.eos) psi = T > 0 ? lae(rp[2 * (T - 1)], rp[2 * (T - 1) + 1]) : p.logzero;
    p.psi[i] = psi;
}

// One 64-thread workgroup per hypothesis, one lane per candidate.  Everything the T'-step chain reads is
// staged in LDS first by all lanes in parallel (the candidate's log-prob column x[t, c] - a stride-V
// gather -, phi[t] and the blank terms, which only depend on the hypothesis), so the dependent chain is
// 2 LDS reads + 3 log-add-exps per frame instead of 4 global loads (206 -> ~25 us per beam step at
// T' = 200); the new states leave through LDS as coalesced rows.
__device__ __forceinline__ float lae_fast(float a, float b) {
    const float m = fmaxf(a, b);
    return m + __logf(1.f + __expf(-fabsf(a - b)));
}

__global__ __launch_bounds__(64) void ctc_prefix_lds_kernel(PrefixArgs p) {
    extern __shared__ float sm[];
    const int h = blockIdx.x, lane = threadIdx.x;
    const int TS = p.T, V = p.V, C = p.C;              // TS: stride of the state arrays (LDS images included)
    const int u = p.row_mem ? p.row_mem[h] : 0;
    const int T = p.mem_len ? min(p.mem_len[u], TS) : TS;   // frames of this hypothesis' utterance
    const float *px = p.x + (size_t)u * TS * V;
    float *s_phi = sm;                 // [TS] logaddexp(r_prev[t,0], r_prev[t,1])
    float *s_rb = s_phi + TS;          // [TS] r_prev[t,1]
    float *s_xb = s_rb + TS;           // [TS] x[t, blank]
    float *s_xc = s_xb + TS;           // [TS][C]
    float *s_ro = s_xc + (size_t)TS * C;   // [C][2 TS]
    const float *rp = p.r_prev + (size_t)h * TS * 2;
    for (int t = lane; t < T; t += 64) {
        const float a = rp[2 * t], b = rp[2 * t + 1];
        s_phi[t] = lae_fast(a, b);
        s_rb[t] = b;
        s_xb[t] = px[(size_t)t * V + p.blank];
    }
    for (int i = lane; i < T * C; i += 64) {
        const int t = i / C, c = i - t * C;
        s_xc[i] = px[(size_t)t * V + p.cand[(size_t)h * C + c]];
    }
    for (int i = lane; i < 2 * TS * C; i += 64) s_ro[i] = p.logzero;
    __syncthreads();
    const int plen = p.plen[h];
    if (lane < C) {
        const int c = p.cand[(size_t)h * C + lane];
        const bool same = plen > 0 && c == p.last[h];
        const int start = plen > 1 ? plen : 1;
        float *ro = s_ro + (size_t)lane * 2 * TS;
        if (plen == 0) ro[0] = s_xc[lane];
        float rn = (start - 1 < T) ? ro[2 * (start - 1)] : p.logzero;
        float rb = (start - 1 < T) ? ro[2 * (start - 1) + 1] : p.logzero;
        float psi = rn;
        for (int t = start; t < T; ++t) {
            const float phi = same ? s_rb[t - 1] : s_phi[t - 1];
            const float xc = s_xc[t * C + lane];
            const float nn = lae_fast(rn, phi) + xc;
            const float nb = lae_fast(rb, rn) + s_xb[t];
            psi = lae_fast(psi, phi + xc);
            rn = nn;
            rb = nb;
            ro[2 * t] = rn;
            ro[2 * t + 1] = rb;
        }
        if (c == p.eos) psi = T > 0 ? s_phi[T - 1] : p.logzero;
        p.psi[(size_t)h * C + lane] = psi;
    }
    __syncthreads();
    float *out = p.r_out + (size_t)h * C * 2 * TS;
    for (int i = lane; i < 2 * TS * C; i += 64) out[i] = s_ro[i];
}

}  // namespace

extern "C" int asrk_ctc_prefix_score_f32(const float *x, const float *r_prev, const int *prefix_len,
                                         const int *last_char, const int *candidates, float *psi,
                                         float *r_out, int n, int C, int T, int V, int blank, int eos,
                                         float logzero, void *stream) {
    return asrk_ctc_prefix_score_multi_f32(x, nullptr, nullptr, r_prev, prefix_len, last_char, candidates, psi,
                                           r_out, n, C, T, V, 1, blank, eos, logzero, stream);
}

extern "C" int asrk_ctc_prefix_score_multi_f32(const float *x, const int *row_mem, const int *mem_len,
                                               const float *r_prev, const int *prefix_len, const int *last_char,
                                               const int *candidates, float *psi, float *r_out, int n, int C,
                                               int T, int V, int U, int blank, int eos, float logzero,
                                               void *stream) {
    if (n < 0 || C < 0 || T <= 0 || V <= 0 || U <= 0) return ASRK_EINVAL;
    if ((row_mem == nullptr) != (mem_len == nullptr)) return ASRK_EINVAL;
    if (n == 0 || C == 0) return ASRK_OK;
    if (!x || !r_prev || !prefix_len || !last_char || !candidates || !psi || !r_out) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    PrefixArgs a{x, r_prev, prefix_len, last_char, candidates, psi, r_out, n, C, T, V, blank, eos, logzero,
                 row_mem, mem_len};
    asrk_prof_begin_(PROF_CTC, s);
    const size_t lds = ((size_t)3 * T + (size_t)3 * T * C) * sizeof(float);
    if (C <= 64 && lds <= 150 * 1024) {
        ASRK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ctc_prefix_lds_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(ctc_prefix_lds_kernel, dim3(n), dim3(64), lds, s, a);
    } else {
        hipLaunchKernelGGL(ctc_prefix_kernel, dim3(asrk_div_up(n * C, 64)), dim3(64), 0, s, a);
    }
    asrk_prof_end_(PROF_CTC, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}
