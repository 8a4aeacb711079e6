// Device prefix beam search for pure-CTC decoding (reference: CTCBeamDecoder.forward, src/ctc.py:241-352).
//
// One workgroup per utterance owns the whole search: the beam tables live in a caller workspace, the per-frame
// scratch (expanded entries, ranks, parent-pair tables) in LDS, and every frame is the sequence of data-parallel
// phases of prefix_beam.inc - candidate ranking, expansion, string-order rank sort, run de-duplication, score
// rank sort, rebuild.  Without an RNN-LM all frames run inside ONE launch; with LM fusion the host enqueues one
// launch per frame followed by the batched LM step for the rows the kernel marks (out_gidx), without ever
// reading anything back until the final hypotheses - the reference's loop does a vocabulary-sized Python sort
// and ~beam x (cand + 1) deep copies per frame on the host.  Integer / byte work plus float64 log-adds on
// <= 1024 entries: latency-bound, a single CU; utterances are the parallel axis (one launch per utterance, or
// one rank per utterance under torch.distributed: bin/test_asr.py).
#include "common.h"

#define PB_HD __device__
#define PB_FOR(i, n) for (int i = (int)threadIdx.x; i < (n); i += (int)blockDim.x)
#define PB_SYNC() __syncthreads()
#define PB_TID0 (threadIdx.x == 0)
// optional phase timeline (tools/ctc_beam_bench.py --timeline): thread 0 stamps the shader clock after every phase
__device__ unsigned long long *pb_dbg_ptr = nullptr;
__device__ int pb_dbg_frame = 0;
#define PB_STAMP(k)                                                                                     \
    do {                                                                                                \
        if (pb_dbg_ptr && threadIdx.x == 0 && pb_dbg_frame < 64) pb_dbg_ptr[pb_dbg_frame * 16 + (k)] = wall_clock64(); \
    } while (0)
__device__ __forceinline__ double pb_exp(double v) { return exp(v); }
__device__ __forceinline__ double pb_log1p(double v) { return log1p(v); }
struct PBState;
__device__ void pb_topw_device(const PBState &s, int m, int W);
#define PB_TOPW(s, m, W) pb_topw_device(s, m, W)
#include "prefix_beam.inc"

// sorted(B, reverse=True, key=score)[:W] (src/ctc.py:331-337) without the m x m rank count: every wave extracts the
// W best of its 1/8 of the survivors (keys in registers, W rounds of a wave arg-max, ties to the earlier position =
// the stable order), wave 0 merges the 8 x W.  Same total order as the rank count of the host build.
template <int CTRL>
__device__ __forceinline__ void pb_dpp_step_d(double &bk, int &bp) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, bk);
    const int lo = __builtin_amdgcn_mov_dpp((int)(u & 0xffffffffu), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(u >> 32), CTRL, 0xf, 0xf, true);
    const double ok = __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
    const int op = __builtin_amdgcn_mov_dpp(bp, CTRL, 0xf, 0xf, true);
    if (op != 0x7fffffff && (bp == 0x7fffffff || ok > bk || (ok == bk && op < bp))) { bk = ok; bp = op; }
}
__device__ __forceinline__ void pb_wave_best_d(double &bk, int &bp) {
    pb_dpp_step_d<0xB1>(bk, bp);
    pb_dpp_step_d<0x4E>(bk, bp);
    pb_dpp_step_d<0x141>(bk, bp);
    pb_dpp_step_d<0x140>(bk, bp);
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        const double ok = __shfl_xor(bk, o, 64);
        const int op = __shfl_xor(bp, o, 64);
        if (op != 0x7fffffff && (bp == 0x7fffffff || ok > bk || (ok == bk && op < bp))) { bk = ok; bp = op; }
    }
}
__device__ void pb_topw_device(const PBState &s, int m, int W) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (int)blockDim.x / 64;
    const int per = (m + nw - 1) / nw, a0 = wave * per, a1 = min(m, a0 + per);      // per <= 128 (m <= 1024)
    double k[2];
    int pos[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        pos[q] = a0 + lane + 64 * q;
        k[q] = pos[q] < a1 ? s.key[pos[q]] : 0.0;
        if (pos[q] >= a1 || !(k[q] == k[q])) pos[q] = 0x7fffffff;                    // out of range / NaN: never chosen
    }
    double pk = 0.0;
    int pp = -1;
    bool first = true;
    for (int c = 0; c < W; ++c) {
        double bk = 0.0;
        int bp = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bool eligible = pos[q] != 0x7fffffff && (first || k[q] < pk || (k[q] == pk && pos[q] > pp));
            if (eligible && (bp == 0x7fffffff || k[q] > bk || (k[q] == bk && pos[q] < bp))) { bk = k[q]; bp = pos[q]; }
        }
        pb_wave_best_d(bk, bp);
        if (lane == 0) { s.w_key[wave * W + c] = bk; s.w_pos[wave * W + c] = bp; }
        pk = bk; pp = bp; first = false;
        if (bp == 0x7fffffff) {
            for (int c2 = c + 1 + lane; c2 < W; c2 += 64) { s.w_key[wave * W + c2] = 0.0; s.w_pos[wave * W + c2] = 0x7fffffff; }
            break;
        }
    }
    __syncthreads();
    if (wave == 0) {
        const int n = nw * W;
        pk = 0.0; pp = -1; first = true;
        for (int c = 0; c < W && c < m; ++c) {
            double bk = 0.0;
            int bp = 0x7fffffff;
            for (int q = lane; q < n; q += 64) {
                const double kq = s.w_key[q];
                const int pq = s.w_pos[q];
                const bool eligible = pq != 0x7fffffff && (first || kq < pk || (kq == pk && pq > pp));
                if (eligible && (bp == 0x7fffffff || kq > bk || (kq == bk && pq < bp))) { bk = kq; bp = pq; }
            }
            pb_wave_best_d(bk, bp);
            if (lane == 0 && bp != 0x7fffffff) s.order[c] = s.m_list[bp];
            pk = bk; pp = bp; first = false;
        }
    }
    __syncthreads();
}

namespace {

constexpr int PB_THREADS = 512;

struct PBLaunch {
    PBState s;
    const float *ctc;            // [T][V] log-probs
    const float *lm;             // [W][V] or null
    const unsigned char *allowed;  // [V] 1 = in vocab_range
    float lw;
    int T, t0, t1;               // frames [t0, t1) of T
    int cur;                     // beam buffer holding the current beam at t0
    int lm_follows;              // an LM step is run after the (single) frame of this launch
    int init;                    // reset the beam to the single empty hypothesis first
};

// Candidate ranking of src/ctc.py:296-303 for one row: the C best symbols of ctc + lw*lm among the allowed ones,
// descending score, ties in vocab_range order (ascending id).  Two stages, no global re-reads, one workgroup
// barrier per row: every wave holds 1/8 of the vocabulary in registers (PB_RV per lane: V <= 8 * 64 * PB_RV) and
// extracts ITS C best by C rounds of a wave arg-max (round c takes the best element that sorts strictly after round
// c-1's winner: no "taken" flags); the global C best are among those 8 x C, which wave 0 merges the same way from
// LDS.  (First version: per-round scans of global memory, 1.4 ms per frame; second: workgroup arg-max with two
// barriers per round, ~100 us per row; this one ~10 us per row.)
constexpr int PB_RV = 32;
// wave-wide arg-max of (score, index) pairs, every lane ends with the winner: the four steps inside a 16-lane row
// are DPP moves (quad permutes, half-mirror, mirror: a few cycles each), only the two steps across rows go through
// the LDS crossbar (ds_bpermute, ~100 cycles each way) - the six-step shuffle butterfly cost ~1.5k cycles per round
template <int CTRL>
__device__ __forceinline__ void pb_dpp_step(float &bsc, int &bv) {
    const float osc = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, bsc), CTRL, 0xf, 0xf, true));
    const int ov = __builtin_amdgcn_mov_dpp(bv, CTRL, 0xf, 0xf, true);
    if (ov != 0x7fffffff && (bv == 0x7fffffff || osc > bsc || (osc == bsc && ov < bv))) { bsc = osc; bv = ov; }
}
__device__ __forceinline__ void pb_wave_best(float &bsc, int &bv) {
    pb_dpp_step<0xB1>(bsc, bv);        // quad_perm [1,0,3,2]
    pb_dpp_step<0x4E>(bsc, bv);        // quad_perm [2,3,0,1]
    pb_dpp_step<0x141>(bsc, bv);       // row_half_mirror
    pb_dpp_step<0x140>(bsc, bv);       // row_mirror
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        const float osc = __shfl_xor(bsc, o, 64);
        const int ov = __shfl_xor(bv, o, 64);
        if (ov != 0x7fffffff && (bv == 0x7fffffff || osc > bsc || (osc == bsc && ov < bv))) { bsc = osc; bv = ov; }
    }
}
// One WAVE per row, the whole row's masked scores in registers (V <= 64 * PB_RW): with an LM the eight waves rank
// eight rows at once.
constexpr int PB_RW = 80;
__device__ void pb_rank_wave(const PBState &s, int row, const float *x, const float *lmrow, float lw,
                             const unsigned char *allowed, int lane) {
    float r[PB_RW];
#pragma unroll
    for (int q = 0; q < PB_RW; ++q) {
        const int v = lane + 64 * q;
        const bool ok = v < s.V && allowed[v];
        r[q] = ok ? (lmrow ? x[v] + lw * lmrow[v] : x[v]) : __builtin_nanf("");
    }
    float psc = INFINITY;
    int pv = -1;
    for (int c = 0; c < s.C; ++c) {
        float bsc = -INFINITY;
        int bv = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < PB_RW; ++q) {
            const int v = lane + 64 * q;
            const float val = r[q];
            const bool eligible = val < psc || (val == psc && v > pv);
            if (eligible && (bv == 0x7fffffff || val > bsc)) { bsc = val; bv = v; }   // -inf scores rank too (after every finite one, in index order)
        }
        pb_wave_best(bsc, bv);
        if (lane == 0) s.cand[row * s.C + c] = bv == 0x7fffffff ? 0 : bv;
        psc = bsc; pv = bv;
    }
}
__device__ void pb_rank_row(const PBState &s, float *psc_l, int *pv_l, int row, const float *x, const float *lmrow,
                            float lw, const unsigned char *allowed) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = PB_THREADS / 64;
    const int per = (s.V + nw - 1) / nw, v0 = wave * per, v1 = min(s.V, v0 + per);
    float r[PB_RV];
#pragma unroll
    for (int q = 0; q < PB_RV; ++q) {
        const int v = v0 + lane + 64 * q;
        const bool ok = v < v1 && allowed[v];
        r[q] = ok ? (lmrow ? x[v] + lw * lmrow[v] : x[v]) : __builtin_nanf("");
    }
    float psc = INFINITY;
    int pv = -1;
    for (int c = 0; c < s.C; ++c) {                             // stage 1: this wave's C best
        float bsc = -INFINITY;
        int bv = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < PB_RV; ++q) {                       // ascending v: the first maximum has the smallest index
            const int v = v0 + lane + 64 * q;
            const float val = r[q];
            const bool eligible = val < psc || (val == psc && v > pv);   // NaN (masked / out of range): never
            if (eligible && (bv == 0x7fffffff || val > bsc)) { bsc = val; bv = v; }   // -inf scores rank too (after every finite one, in index order)
        }
        pb_wave_best(bsc, bv);
        if (lane == 0) { psc_l[wave * s.C + c] = bsc; pv_l[wave * s.C + c] = bv; }
        psc = bsc; pv = bv;
        if (bv == 0x7fffffff) {                                 // slice exhausted: pad the rest
            for (int c2 = c + 1 + lane; c2 < s.C; c2 += 64) { psc_l[wave * s.C + c2] = -INFINITY; pv_l[wave * s.C + c2] = 0x7fffffff; }
            break;
        }
    }
    __syncthreads();
    if (wave == 0) {                                            // stage 2: merge the nw x C survivors
        const int n = nw * s.C;
        psc = INFINITY; pv = -1;
        for (int c = 0; c < s.C; ++c) {
            float bsc = -INFINITY;
            int bv = 0x7fffffff;
            for (int q = lane; q < n; q += 64) {
                const float val = psc_l[q];
                const int v = pv_l[q];
                const bool eligible = v != 0x7fffffff && (val < psc || (val == psc && v > pv));
                if (eligible && (val > bsc || (val == bsc && v < bv))) { bsc = val; bv = v; }
            }
            pb_wave_best(bsc, bv);
            if (lane == 0) s.cand[row * s.C + c] = bv == 0x7fffffff ? 0 : bv;
            psc = bsc; pv = bv;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(PB_THREADS) void prefix_beam_kernel(PBLaunch p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pb_lds[];
    PBState s = p.s;
    // ---- carve the per-frame scratch out of LDS
    unsigned char *q = pb_lds;
    auto take = [&](size_t bytes) { unsigned char *r = q; q += (bytes + 15) & ~(size_t)15; return r; };
    s.e_pb = (double *)take(PB_MAX_ENTRIES * 8); s.e_pnb = (double *)take(PB_MAX_ENTRIES * 8);
    s.e_sc = (double *)take(PB_MAX_ENTRIES * 8); s.e_dig = (unsigned long long *)take(PB_MAX_ENTRIES * 8);
    s.key = (double *)take(PB_MAX_ENTRIES * 8);
    s.t_tail = (unsigned long long *)take(PB_PAIRS * 8);
    s.t_tkey = (unsigned long long *)take(PB_PAIRS * 8);
    s.e_lk = (unsigned long long *)take(PB_MAX_ENTRIES * 8);
    s.s_pb1 = (double *)take(PB_MAX_BEAM * 8); s.s_pnb1 = (double *)take(PB_MAX_BEAM * 8);
    s.s_same = (double *)take(PB_MAX_BEAM * 8); s.s_diff = (double *)take(PB_MAX_BEAM * 8);
    s.e_par = (int *)take(PB_MAX_ENTRIES * 4); s.e_tok = (int *)take(PB_MAX_ENTRIES * 4);
    s.sorted = (int *)take(PB_MAX_ENTRIES * 4); s.m_list = (int *)take(PB_MAX_ENTRIES * 4);
    s.order = (int *)take(PB_MAX_BEAM * 4); s.off = (int *)take(PB_MAX_BEAM * 4); s.fin = (int *)take(PB_MAX_BEAM * 4);
    s.r_len = (int *)take(PB_MAX_BEAM * 4); s.r_slen = (int *)take(PB_MAX_BEAM * 4); s.r_last = (int *)take(PB_MAX_BEAM * 4);
    s.t_lcp = (int *)take(PB_PAIRS * 4);
    s.cand = (int *)take((size_t)s.W * s.C * 4);
    s.scal = (int *)take(16);
    s.w_key = (double *)take((size_t)(PB_THREADS / 64) * PB_MAX_BEAM * 8);
    s.w_pos = (int *)take((size_t)(PB_THREADS / 64) * PB_MAX_BEAM * 4);
    float *psc_l = (float *)take((size_t)(PB_THREADS / 64) * s.C * 4);
    int *pv_l = (int *)take((size_t)(PB_THREADS / 64) * s.C * 4);
    s.bnd = take(PB_MAX_ENTRIES);
    s.t_dif = (signed char *)take(PB_PAIRS); s.t_pre = take(PB_PAIRS);

    int cur = p.cur;
    if (p.init) {
        // B = [CTCHypothesis()]: empty sequence, Pr- = 0, Pr+ = LOG_ZERO, updated_lm = True when an LM is fused
        if (threadIdx.x == 0) {
            const PBBeam &b = s.beam[cur];
            b.len[0] = 0; b.slen[0] = 0; b.pb[0] = 0.0; b.pnb[0] = PB_LOG_ZERO; b.upd[0] = 1;
            b.lcp[0] = 0; b.dif[0] = 0; b.pre[0] = 1; b.tail[0] = 0ull;
            s.nb[cur] = 1;
        }
        __syncthreads();
    }
    for (int t = p.t0; t < p.t1; ++t) {
        const float *x = p.ctc + (size_t)t * s.V;
        const int nb = s.nb[cur];
        PB_STAMP(0);
        // candidate ranking: per row with an LM, once (row 0) without
        const int rows = p.lm ? nb : 1;
        if (rows > 1 && s.V <= 64 * PB_RW) {                  // several rows (LM fusion): one wave per row, 8 at a time
            const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
            for (int i = wave; i < rows; i += PB_THREADS / 64)
                pb_rank_wave(s, i, x, p.lm + (size_t)i * s.V, p.lw, p.allowed, lane);
            __syncthreads();
        } else {
            for (int i = 0; i < rows; ++i)
                pb_rank_row(s, psc_l, pv_l, i, x, p.lm ? p.lm + (size_t)i * s.V : nullptr, p.lw, p.allowed);
        }
        if (!p.lm) {
            PB_FOR(z, (nb - 1) * s.C) s.cand[s.C + z] = s.cand[z % s.C];
            __syncthreads();
        }
        PB_STAMP(15);
        pb_frame(s, cur, x, p.lm, p.lw, t == p.T - 1, p.lm_follows);
        if (pb_dbg_ptr && threadIdx.x == 0) pb_dbg_frame += 1;
        cur ^= 1;
    }
}

size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

struct WsLayout {
    size_t tok[2], str[2], len[2], slen[2], pb[2], pnb[2], upd[2], lcp[2], dif[2], pre[2], tail[2], nb, outp, outl,
        outg, total;
};
WsLayout ws_layout(int W, int Lcap, int Scap) {
    WsLayout l;
    size_t o = 0;
    for (int k = 0; k < 2; ++k) {
        l.pb[k] = o; o += align16((size_t)W * 8);
        l.pnb[k] = o; o += align16((size_t)W * 8);
        l.tok[k] = o; o += align16((size_t)W * Lcap * 4);
        l.str[k] = o; o += align16((size_t)W * Scap);
        l.len[k] = o; o += align16((size_t)W * 4);
        l.slen[k] = o; o += align16((size_t)W * 4);
        l.upd[k] = o; o += align16((size_t)W * 4);
        l.tail[k] = o; o += align16((size_t)PB_PAIRS * 8);
        l.lcp[k] = o; o += align16((size_t)PB_PAIRS * 4);
        l.dif[k] = o; o += align16((size_t)PB_PAIRS);
        l.pre[k] = o; o += align16((size_t)PB_PAIRS);
    }
    l.nb = o; o += 16;
    l.outp = o; o += align16((size_t)W * 4);
    l.outl = o; o += align16((size_t)W * 4);
    l.outg = o; o += align16((size_t)W * 4);
    l.total = o;
    return l;
}

size_t lds_bytes(int W, int C, int V) {
    return (size_t)PB_MAX_ENTRIES * (8 * 6 + 4 * 4 + 1) + (size_t)PB_MAX_BEAM * (8 * 4 + 4 * 6) +
           (size_t)PB_PAIRS * (8 * 2 + 4 + 2) + (size_t)W * C * 4 + 16 + (size_t)(PB_THREADS / 64) * C * 8 + (size_t)(PB_THREADS / 64) * PB_MAX_BEAM * 12 + 40 * 16;
}

}  // namespace

// Lcap = T + 1 tokens per hypothesis (at most one symbol per frame), Scap = 5 * Lcap characters.
extern "C" size_t asrk_ctc_prefix_beam_ws_bytes(int beam, int T) {
    if (beam <= 0 || beam > PB_MAX_BEAM || T < 0) return 0;
    return ws_layout(beam, T + 1, 5 * (T + 1)).total;
}
constexpr int PB_MAX_V = (PB_THREADS / 64) * 64 * PB_RV;   // 16384: a row of masked scores lives in the workgroup's registers

// Offsets (bytes into the workspace) of what the caller reads back / feeds the LM with: the live-row count of
// beam buffer `buf` (int32), its lengths [beam] int32 and tokens [beam][T+1] int32, and the per-row LM
// bookkeeping of the last frame (parent row, last token, gather index; [beam] int32 each).
extern "C" int asrk_ctc_prefix_beam_ws_offsets(int beam, int T, int buf, int64_t *nb_off, int64_t *len_off,
                                               int64_t *tok_off, int64_t *parent_off, int64_t *last_off,
                                               int64_t *gidx_off) {
    if (beam <= 0 || beam > PB_MAX_BEAM || T < 0 || (buf != 0 && buf != 1)) return ASRK_EINVAL;
    const WsLayout l = ws_layout(beam, T + 1, 5 * (T + 1));
    if (nb_off) *nb_off = (int64_t)l.nb + 4 * buf;
    if (len_off) *len_off = (int64_t)l.len[buf];
    if (tok_off) *tok_off = (int64_t)l.tok[buf];
    if (parent_off) *parent_off = (int64_t)l.outp;
    if (last_off) *last_off = (int64_t)l.outl;
    if (gidx_off) *gidx_off = (int64_t)l.outg;
    return ASRK_OK;
}

// debug: device buffer of 64 x 16 uint64 receiving the phase stamps of the first 64 frames (NULL = off)
extern "C" int asrk_ctc_prefix_beam_set_debug_(void *buf) {
    unsigned long long *p = reinterpret_cast<unsigned long long *>(buf);
    int zero = 0;
    ASRK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(pb_dbg_ptr), &p, sizeof(p)));
    ASRK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(pb_dbg_frame), &zero, sizeof(zero)));
    return ASRK_OK;
}

extern "C" int asrk_ctc_prefix_beam_f32(const float *ctc, int T, int V, const unsigned char *allowed, int beam,
                                        int cand, const float *lm, float lm_weight, int t0, int t1, int cur_buf,
                                        int init, int lm_step_follows, void *ws, size_t ws_bytes, void *stream) {
    if (T <= 0 || V <= 0 || V > 99999 || beam <= 0 || beam > PB_MAX_BEAM || cand <= 0 || cand > V) return ASRK_EINVAL;
    if ((size_t)beam * (cand + 1) > PB_MAX_ENTRIES || V > PB_MAX_V) return ASRK_ESHAPE;
    if (t0 < 0 || t1 > T || t0 > t1 || (cur_buf != 0 && cur_buf != 1)) return ASRK_EINVAL;
    if (!ctc || !allowed || !ws) return ASRK_EINVAL;
    if (lm && t1 - t0 > 1) return ASRK_EINVAL;            // LM fusion: one frame per launch (an LM step in between)
    const int Lcap = T + 1, Scap = 5 * Lcap;
    const WsLayout l = ws_layout(beam, Lcap, Scap);
    if (ws_bytes < l.total) return ASRK_EWORKSPACE;
    if ((reinterpret_cast<uintptr_t>(ws) & 15) != 0) return ASRK_EINVAL;
    if (t0 == t1 && !init) return ASRK_OK;
    unsigned char *w = reinterpret_cast<unsigned char *>(ws);
    PBLaunch p{};
    PBState &s = p.s;
    s.W = beam; s.C = cand; s.V = V; s.Lcap = Lcap; s.Scap = Scap;
    for (int k = 0; k < 2; ++k) {
        s.beam[k].pb = (double *)(w + l.pb[k]); s.beam[k].pnb = (double *)(w + l.pnb[k]);
        s.beam[k].tok = (int *)(w + l.tok[k]); s.beam[k].str = w + l.str[k];
        s.beam[k].len = (int *)(w + l.len[k]); s.beam[k].slen = (int *)(w + l.slen[k]);
        s.beam[k].upd = (int *)(w + l.upd[k]);
        s.beam[k].lcp = (int *)(w + l.lcp[k]); s.beam[k].dif = (signed char *)(w + l.dif[k]);
        s.beam[k].pre = w + l.pre[k]; s.beam[k].tail = (unsigned long long *)(w + l.tail[k]);
    }
    s.nb = (int *)(w + l.nb);
    s.out_parent = (int *)(w + l.outp); s.out_last = (int *)(w + l.outl); s.out_gidx = (int *)(w + l.outg);
    p.ctc = ctc; p.lm = lm; p.allowed = allowed; p.lw = lm_weight;
    p.T = T; p.t0 = t0; p.t1 = t1; p.cur = cur_buf; p.lm_follows = lm_step_follows; p.init = init;
    const size_t lds = lds_bytes(beam, cand, V);
    // per call: the attribute belongs to the CURRENT device's copy of the kernel, and a latched flag would be wrong on
    // a second GPU or thread; the call is cheap next to this latency-bound launch
    ASRK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(prefix_beam_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    if (lds > 150 * 1024) return ASRK_ESHAPE;
    hipLaunchKernelGGL(prefix_beam_kernel, dim3(1), dim3(PB_THREADS), lds, (hipStream_t)stream, p);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}
