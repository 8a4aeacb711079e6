// Device-side bookkeeping of the joint CTC-attention(-LM) beam search (reference: src/decode.py:150-167 - the loop
// around Hypothesis.addTopk, 209-239): one decode position of EVERY utterance of a device batch without a read-back.
//
// Rows: utterance u owns the B row slots [u*B, (u+1)*B); a slot is alive or not.  At position t every live row holds a
// hypothesis of t labels.  From the position's top-B (score, label) of every row - and, with CTC, the candidates'
// prefix scores - one workgroup per utterance
//   * forms the B*B continuation records in the reference's order (row-major: hypothesis, then rank), drops <eos> and
//     labels the CTC scorer did not see, and ranks them by the AVERAGE score (sum of scores / length, float64 like the
//     Python floats of the reference) with ties in record order (a stable sort);
//   * writes the B best into the utterance's slots in rank order: parent row, label, score, candidate column, CTC
//     prefix probability, new score sum - the inputs of the next position's gathers - and the back-pointer history;
//   * logs finished hypotheses: a row whose top-B contains <eos> (its last occurrence, as the reference's loop leaves
//     it) once t >= min_len; and, when the utterance ends (no continuation left, the length limit, or beam 1 after its
//     first finished hypothesis), the surviving continuations.
// The host reads the history and the log once, after the last position.
#include "common.h"

namespace {

constexpr int MAXB = 32, MAXC = 48;

struct SelArgs {
    const float *topv;
    const int64_t *topi;          // [R,B]
    const float *psi;
    const int64_t *cand;          // [R,C] (C == 0: none)
    int U, B, C, t, R, fcap;
    const int *min_len, *max_len; // [U]
    int *alive;                   // [R]
    double *ssum;                 // [R]
    int *utt_done;                // [U]
    int64_t *prev_token, *parent, *col;   // [R]
    float *pctc;                  // [R]
    int *hist_tok;
    float *hist_sc;
    int *hist_par;                // [lmax][R], row t written
    int *fin_count;               // [U]
    int *fin_kind, *fin_t, *fin_row;      // [U][fcap]
    float *fin_term;
    double *fin_ssum;
    int *live_utts;               // [1]
};

__global__ __launch_bounds__(256) void beam_select_kernel(SelArgs p) {
    __shared__ double s_avg[MAXB * MAXB];
    __shared__ short s_col[MAXB * MAXB];
    __shared__ unsigned char s_valid[MAXB * MAXB];
    __shared__ double s_ssum[MAXB];
    __shared__ float s_term[MAXB];
    __shared__ int s_alive[MAXB], s_eos[MAXB];
    __shared__ int s_nvalid;
    __shared__ float s_tv[MAXB * MAXB];
    __shared__ int s_ti[MAXB * MAXB];
    __shared__ int s_cd[MAXB * MAXC];
    const int u = blockIdx.x, tid = threadIdx.x, B = p.B, C = p.C, R0 = u * B, NR = B * B;
    if (p.utt_done[u]) {
        for (int j = tid; j < B; j += 256) {
            p.alive[R0 + j] = 0;
            p.prev_token[R0 + j] = 0;
            p.parent[R0 + j] = R0;
            p.col[R0 + j] = 0;
            p.pctc[R0 + j] = 0.f;
        }
        return;
    }
    // the utterance's block of the position's results -> LDS, coalesced (the loops below walked global memory one
    // dependent load at a time: 37 us per launch, as long as a prefix-score launch)
    const int NC = B * C;
    for (int q = tid; q < NR; q += 256) {
        s_tv[q] = p.topv[(size_t)R0 * B + q];
        s_ti[q] = (int)p.topi[(size_t)R0 * B + q];
    }
    for (int q = tid; q < NC; q += 256) s_cd[q] = (int)p.cand[(size_t)R0 * C + q];
    if (tid == 0) s_nvalid = 0;
    for (int i = tid; i < B; i += 256) {
        s_ssum[i] = p.ssum[R0 + i];
        s_alive[i] = p.alive[R0 + i];
    }
    __syncthreads();
    for (int i = tid; i < B; i += 256) {
        int last = -1;
        if (s_alive[i])
            for (int k = 0; k < B; ++k)
                if (s_ti[i * B + k] == 1) last = k;
        s_eos[i] = last >= 0;
        s_term[i] = last >= 0 ? s_tv[i * B + last] : 0.f;
    }
    const double len = (double)(p.t + 1);
    for (int rec = tid; rec < NR; rec += 256) {
        const int i = rec / B;
        const int tok = s_ti[rec];
        bool valid = s_alive[i] && tok != 1;
        int c = 0;
        if (valid && C > 0) {
            for (; c < C; ++c)
                if (s_cd[i * C + c] == tok) break;
            if (c == C) {                 // a label the prefix scorer did not see: dropped
                valid = false;
                c = 0;
            }
        }
        s_avg[rec] = (s_ssum[i] + (double)s_tv[rec]) / len;
        s_col[rec] = (short)c;
        s_valid[rec] = valid ? 1 : 0;
        if (valid) atomicAdd(&s_nvalid, 1);
    }
    __syncthreads();
    const int nsurv = min(B, s_nvalid);
    for (int rec = tid; rec < NR; rec += 256) {
        if (!s_valid[rec]) continue;
        const double a = s_avg[rec];
        int rank = 0;
        for (int m = 0; m < NR; ++m)
            rank += (s_valid[m] && (s_avg[m] > a || (s_avg[m] == a && m < rec))) ? 1 : 0;
        if (rank < B) {
            const int i = rec / B, slot = R0 + rank;
            const size_t row = R0 + i;
            const float sc = s_tv[rec];
            const int64_t tok = s_ti[rec];
            p.prev_token[slot] = tok;
            p.parent[slot] = (int64_t)row;
            p.col[slot] = s_col[rec];
            p.pctc[slot] = C > 0 ? p.psi[row * C + s_col[rec]] : 0.f;
            p.ssum[slot] = s_ssum[i] + (double)sc;
            p.alive[slot] = 1;
            p.hist_tok[(size_t)p.t * p.R + slot] = (int)tok;
            p.hist_sc[(size_t)p.t * p.R + slot] = sc;
            p.hist_par[(size_t)p.t * p.R + slot] = (int)row;
        }
    }
    for (int j = nsurv + tid; j < B; j += 256) {
        p.alive[R0 + j] = 0;
        p.prev_token[R0 + j] = 0;
        p.parent[R0 + j] = R0;
        p.col[R0 + j] = 0;
        p.pctc[R0 + j] = 0.f;
        p.ssum[R0 + j] = 0.0;
    }
    __syncthreads();
    if (tid == 0) {
        int n = p.fin_count[u];
        bool stopped = false;
        const size_t f0 = (size_t)u * p.fcap;
        if (p.t >= p.min_len[u]) {
            for (int i = 0; i < B && !stopped; ++i) {
                if (!s_alive[i] || !s_eos[i]) continue;
                if (n < p.fcap) {
                    p.fin_kind[f0 + n] = 0;
                    p.fin_t[f0 + n] = p.t;
                    p.fin_row[f0 + n] = R0 + i;
                    p.fin_term[f0 + n] = s_term[i];
                    p.fin_ssum[f0 + n] = s_ssum[i] + (double)s_term[i];
                    ++n;
                }
                if (B == 1) stopped = true;          // beam 1 stops at its first finished hypothesis
            }
        }
        if (stopped || nsurv == 0 || p.t + 1 >= p.max_len[u]) {
            if (!stopped)
                for (int j = 0; j < nsurv; ++j) {
                    if (n >= p.fcap) break;
                    p.fin_kind[f0 + n] = 1;
                    p.fin_t[f0 + n] = p.t;
                    p.fin_row[f0 + n] = R0 + j;
                    p.fin_term[f0 + n] = 0.f;
                    p.fin_ssum[f0 + n] = p.ssum[R0 + j];
                    ++n;
                }
            p.utt_done[u] = 1;
            for (int j = 0; j < B; ++j) p.alive[R0 + j] = 0;
            atomicSub(p.live_utts, 1);
        }
        p.fin_count[u] = n;
    }
}

}  // namespace

extern "C" int asrk_beam_select_f32(const float *topv, const int64_t *topi, const float *psi, const int64_t *cand, int U,
                                    int B, int C, int t, int lmax, int fcap, const int *min_len, const int *max_len,
                                    int *alive, double *ssum, int *utt_done, int64_t *prev_token, int64_t *parent,
                                    int64_t *col, float *pctc, int *hist_tok, float *hist_sc, int *hist_par, int *fin_count,
                                    int *fin_kind, int *fin_t, int *fin_row, float *fin_term, double *fin_ssum,
                                    int *live_utts, void *stream) {
    if (U < 0 || B < 1 || C < 0 || t < 0 || t >= lmax || fcap < 1) return ASRK_EINVAL;
    if (B > MAXB || C > MAXC) return ASRK_ESHAPE;
    if (U == 0) return ASRK_OK;
    if (!topv || !topi || (C > 0 && (!psi || !cand)) || !min_len || !max_len || !alive || !ssum || !utt_done ||
        !prev_token || !parent || !col || !pctc || !hist_tok || !hist_sc || !hist_par || !fin_count || !fin_kind || !fin_t ||
        !fin_row || !fin_term || !fin_ssum || !live_utts)
        return ASRK_EINVAL;
    SelArgs a{topv, topi, psi, cand, U, B, C, t, U * B, fcap, min_len, max_len, alive, ssum, utt_done, prev_token, parent,
              col, pctc, hist_tok, hist_sc, hist_par, fin_count, fin_kind, fin_t, fin_row, fin_term, fin_ssum, live_utts};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(beam_select_kernel, dim3((unsigned)U), dim3(256), 0, s, a);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}


// ---- the survivors' states for the next position: up to 8 row gathers in ONE launch (decoder h / c, previous alignment,
// LM states per layer, CTC prefix state: seven index_select / advanced-indexing launches per position before).
// segment s: dst_s[i, :] = src_s[parent[i] * mul_s + (use_col_s ? col[i] : 0), :], rows of row_floats_s floats.
namespace {
constexpr int GR_MAXSEG = 8;
struct GatherArgs {
    const float *src[GR_MAXSEG];
    float *dst[GR_MAXSEG];
    int row_floats[GR_MAXSEG];
    int mul[GR_MAXSEG];
    int use_col[GR_MAXSEG];
    const int64_t *parent, *col;
    int n;
};
__global__ __launch_bounds__(256) void gather_rows_multi_kernel(GatherArgs p) {
    const int i = blockIdx.x, sg = blockIdx.y, tid = threadIdx.x;
    const int rf = p.row_floats[sg];
    const int64_t r = p.parent[i] * p.mul[sg] + (p.use_col[sg] ? p.col[i] : 0);
    const float *sp = p.src[sg] + r * rf;
    float *dp = p.dst[sg] + (int64_t)i * rf;
    if ((rf & 3) == 0 && ((reinterpret_cast<uintptr_t>(sp) | reinterpret_cast<uintptr_t>(dp)) & 15) == 0) {
        const f32x4 *s4 = reinterpret_cast<const f32x4 *>(sp);
        f32x4 *d4 = reinterpret_cast<f32x4 *>(dp);
        for (int e = tid; e < rf / 4; e += 256) d4[e] = s4[e];
    } else {
        for (int e = tid; e < rf; e += 256) dp[e] = sp[e];
    }
}
}  // namespace

extern "C" int asrk_gather_rows_multi_f32(int nseg, const float *const *src, float *const *dst, const int *row_floats,
                                          const int *mul, const int *use_col, const int64_t *parent, const int64_t *col,
                                          int n, void *stream) {
    if (nseg < 0 || nseg > GR_MAXSEG || n < 0) return ASRK_EINVAL;
    if (nseg == 0 || n == 0) return ASRK_OK;
    if (!src || !dst || !row_floats || !mul || !use_col || !parent) return ASRK_EINVAL;
    GatherArgs a{};
    for (int s = 0; s < nseg; ++s) {
        if (!src[s] || !dst[s] || row_floats[s] <= 0 || mul[s] <= 0 || (use_col[s] && !col)) return ASRK_EINVAL;
        a.src[s] = src[s]; a.dst[s] = dst[s]; a.row_floats[s] = row_floats[s]; a.mul[s] = mul[s]; a.use_col[s] = use_col[s];
    }
    a.parent = parent; a.col = col; a.n = n;
    hipLaunchKernelGGL(gather_rows_multi_kernel, dim3((unsigned)n, (unsigned)nseg), dim3(256), 0, (hipStream_t)stream, a);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}
