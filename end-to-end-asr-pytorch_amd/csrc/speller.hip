// The attention-decoder ("speller") loop of LAS as ONE C call per direction, for gfx950.
//
// Replaces the teacher-forced decode loop of the reference (src/asr.py:112-148 with tf_rate == 1):
// per step  Attention.forward (src/asr.py:277-313) -> LocationAwareAttention.forward
// (src/module.py:234-258) -> Decoder.forward (src/asr.py:214-221, a 1-step nn.LSTM), and the autograd
// graph PyTorch builds for it.  Single head, location-aware attention, single-layer LSTM decoder
// (config/libri/asr_example.yaml:47-54); other variants use the per-step kernels of attention.hip.
//
// A decode step is ~0.9 GFLOP and ~110 MB of L2/MALL-resident operands (K 7.7 MB, V 52 MB, decoder
// weights 50 MB at cfg3): bandwidth/latency work, not a GEMM.  What costs time in a step-by-step
// framework loop is the ~16 launches + autograd bookkeeping per step (330 us/step measured in round 1
// for ~110 us of kernels).  Here the host enqueues the whole loop from C++ with 4 kernels per forward
// step and 5 per backward step, no per-step allocation, no autograd nodes:
//
//   forward step t     F1  q_t   = tanh(h_t Wq^T + bq)                          skinny MFMA GEMM
//                      F2a c_t   = conv1d(attn_{t-1}, Wc);  e = we.tanh(key + q + tanh(c Wp^T)) + be
//                      F2b attn_t = softmax(e / T, masked);  ctx_t = attn_t . value
//                      F3  gates = eproj_t + ctx_t W_ih[:,E:]^T + h_t W_hh^T -> LSTM cell -> h_{t+1}, c_{t+1}
//   backward step t    B2  [dctx_t | dh_hh] = dG_t [W_ih[:,E:] | W_hh]            skinny MFMA GEMM
//                      B3  dattn = dctx_t . value (+ d att_seq + d attn from step t+1's conv)
//                      B4  softmax/energy backward: dkey +=, dconv, per-block partials of dq/dWp/dwe
//                      B5  conv backward (d attn_{t-1}, dWc partials), dq_pre = dq (1 - q^2)
//                      B6  dh_{t-1} = dq_pre Wq + dh_hh + dstates_{t-1} -> LSTM cell backward -> dG_{t-1}
// (eproj = embedded teacher tokens through W_ih[:, :E] for all steps at once, and every weight
// gradient, are whole-sequence GEMMs outside the loop: the teacher-forced inputs are known up front.)
//
// The skinny GEMMs (M = batch <= 64 rows) run on v_mfma_f32_16x16x4_f32 with the WEIGHT rows as the
// MFMA M dimension (16 rows per workgroup, K split over 8 waves, 16-byte global loads straight into
// the operand layout, no LDS staging: every weight byte is used once per step); the LSTM variants
// order a workgroup's 16 rows unit-major / gate-minor so one lane ends up with i,f,g,o of one cell and
// the cell update is the GEMM epilogue.
#include "common.h"
#include "knobs.h"
#include <cstdlib>

namespace {

__device__ __forceinline__ float tanh_fast(float x) {
    // 1 - 2/(e^{2x}+1) on the hardware exp/rcp (abs error ~2e-7; saturates correctly at +-inf)
    const float e = __expf(2.f * x);
    return 1.f - __fdividef(2.f, e + 1.f);
}
__device__ __forceinline__ float sigmoid_fast(float x) { return __fdividef(1.f, 1.f + __expf(-x)); }

// debug timeline (tools/speller_timeline.py): a launch given a stamp slot (16 uint64) records the shader clock at its
// phase boundaries from wave 0 of its first workgroup (slots 0..9) and the start / end of its LAST workgroup (12, 13)
unsigned long long *g_sp_dbg = nullptr;
int g_sp_dbg_slots = 0;
inline unsigned long long *sp_slot(int step, int kid) {
    const long i = (long)step * 16 + kid;
    return (g_sp_dbg && i < g_sp_dbg_slots) ? g_sp_dbg + i * 16 : nullptr;
}
#define SP_STAMP(ph)                                                                                     \
    do {                                                                                                 \
        if (p.stamps && threadIdx.x == 0) {                                                              \
            if (blockIdx.x == 0 && blockIdx.y == 0) p.stamps[ph] = __builtin_readcyclecounter();         \
            else if (blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1 && ((ph) == 0 || (ph) == 9)) \
                p.stamps[(ph) == 0 ? 12 : 13] = __builtin_readcyclecounter();                            \
        }                                                                                                \
    } while (0)

// ------------------------------------------------------------------------------------ skinny GEMM
constexpr int SK_THREADS = 512, SK_WAVES = 8, SK_CH = 32;   // 32 contraction indices per wave chunk

struct SkSeg {
    const float *x;   // [M, klen] rows at x + m*ldx
    const float *w;   // weight rows at w + r*ldw (klen contiguous floats each)
    long ldx, ldw;
    int klen;
    unsigned xbytes, wbytes;   // extents behind x / w (filled in by launch_skinny: buffer-load bounds)
};

enum { EPI_STORE = 0, EPI_TANH_BIAS = 1, EPI_LSTM_FWD = 2, EPI_LSTM_BWD = 3 };

struct SkArgs {
    SkSeg seg[3];
    int nseg, M, R, H;
    unsigned long long *stamps;   // debug timeline slot or nullptr
    // EPI_STORE / EPI_TANH_BIAS: out[m*ldo + r] = f(acc + bias[r])
    float *out;
    long ldo;
    const float *bias;
    // EPI_LSTM_FWD: pre-activation = acc + pre[m*4H + gate*H + u] + b0[..] + b1[..]
    const float *pre, *b0, *b1;
    const float *c_prev;
    float *c_new, *h_new, *gates;
    float *h_bm;      // optional batch-major copy of h_new: h_bm[m*h_bm_ld + u]
    long h_bm_ld;
    // EPI_LSTM_BWD: dh = acc + add0[m*ld0 + j] + add1[m*ld1 + j]; cell (m, j) of one step
    const float *add0, *add1;
    long ld0, ld1;
    float *dG;                       // activated gates in, pre-activation gradients out [M,4H]
    const float *bc_prev, *bc_new;   // [M,H]
    float *dc;                       // [M,H] in (if dc_valid) / out
    int dc_valid;
    // GRU cell in the same four-rows-per-unit layout (asrk_speller_t::cell): rows r, z, n_x, n_h; `c_prev` / `bc_prev`
    // then point at h_prev, c_new / bc_new are not used, and `dc` carries dh' * z (the direct path into h_prev)
    int gru;
};

template <int VEC>
__device__ __forceinline__ void load8(const float *row, int k, int klen, bool valid, float (&v)[8]) {
    if (VEC && valid && k + 8 <= klen) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(row + k);
        const f32x4 b = *reinterpret_cast<const f32x4 *>(row + k + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
        v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (valid && k + j < klen) ? row[k + j] : 0.f;
}

// grid = ceil(R/16) (LSTM_FWD: ceil(H/4)) workgroups of 512 threads.  Lane l of every wave owns
// weight-row slot (l & 15) and contraction sub-range 8*(l >> 4) of each 32-wide chunk; batch row
// (l & 15) + 16*mt of the x operand.  D[slot][m] comes back as 4 consecutive slots per lane.
// VEC: 0 = unaligned operands (scalar loads straight into the operand layout), 2 = aligned operands: 16-byte BUFFER loads,
// 8 rows x 128 contiguous bytes per wave instruction, parked in a wave-private LDS strip and read back as MFMA fragments.
// (Rounds 2-5 ran the staging path on global loads predicated with `cond ? *p : 0`; the compiler turned each into a
// branch with s_waitcnt vmcnt(0) at the join and re-read a debug mask from the kernel arguments in front of every load,
// so its "software pipelining" never had more than two loads in flight and loads, LDS staging and MFMAs ran one after
// the other - 22.8 / 20.6 us per call of B2 / F3 against 18.7 / 17.8 us for this form on the same box,
// profiles/r06_skinny_ab.log.  Here the loop has no branch: out-of-range rows / contraction indices are an
// out-of-bounds buffer offset, which reads 0 and moves no bytes.)
template <int MT, int EPI, int VEC>
__global__ __launch_bounds__(SK_THREADS) void skinny_kernel(SkArgs p) {
    __shared__ float red[SK_WAVES][MT][4][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = lane & 15, g = lane >> 4;
    SP_STAMP(0);
    int r;
    bool rvalid;
    if (EPI == EPI_LSTM_FWD) {
        const int u = blockIdx.x * 4 + (slot >> 2);
        r = (slot & 3) * p.H + u;
        rvalid = u < p.H;
    } else {
        r = blockIdx.x * 16 + slot;
        rvalid = r < p.R;
    }
    // two accumulator chains per M tile: a dependent 16x16x4 MFMA can only issue every 40 cycles, an
    // independent one every 32
    f32x4 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][0] = acc[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Epilogue inputs (biases, the hoisted input projection, cell state, saved gates) do not depend on the
    // contraction: the MT*64 epilogue threads request them NOW, so their memory round trip (2-3 us, on the
    // serial chain of the decode loop otherwise) overlaps the weight streaming.
    float e_in[EPI == EPI_LSTM_BWD ? 36 : 13];
    const bool e_thr = tid < MT * 64;
    const int e_m = (tid >> 6) * 16 + (lane & 15);
    if (e_thr && e_m < p.M) {
        if (EPI == EPI_STORE || EPI == EPI_TANH_BIAS) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = blockIdx.x * 16 + (lane >> 4) * 4 + q;
                e_in[q] = (rr < p.R && p.bias) ? p.bias[rr] : 0.f;
            }
        } else if (EPI == EPI_LSTM_FWD) {
            const int u = blockIdx.x * 4 + (lane >> 4), H = p.H;
            if (u < H) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const long col = (long)q * H + u;
                    e_in[q] = p.pre ? p.pre[(long)e_m * 4 * H + col] : 0.f;
                    e_in[4 + q] = p.b0 ? p.b0[col] : 0.f;
                    e_in[8 + q] = p.b1 ? p.b1[col] : 0.f;
                }
                e_in[12] = p.c_prev[(long)e_m * H + u];
            }
        } else {
            const int H = p.H;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = blockIdx.x * 16 + (lane >> 4) * 4 + q;
                if (j >= H) continue;
                const float *gr = p.dG + (long)e_m * 4 * H + j;
                e_in[q * 9 + 0] = p.add0 ? p.add0[(long)e_m * p.ld0 + j] : 0.f;
                e_in[q * 9 + 1] = p.add1 ? p.add1[(long)e_m * p.ld1 + j] : 0.f;
                e_in[q * 9 + 2] = gr[0];
                e_in[q * 9 + 3] = gr[H];
                e_in[q * 9 + 4] = gr[2 * (long)H];
                e_in[q * 9 + 5] = gr[3 * (long)H];
                e_in[q * 9 + 6] = p.bc_new[(long)e_m * H + j];
                e_in[q * 9 + 7] = p.dc_valid ? p.dc[(long)e_m * H + j] : 0.f;
                e_in[q * 9 + 8] = p.bc_prev[(long)e_m * H + j];
            }
        }
    }

    if (VEC == 2) {
        extern __shared__ __attribute__((aligned(16))) float stage_all[];
        constexpr int SROWS = 16 + MT * 16, SLD = 36, NX = 2 * MT;
        // chunk buffers in the register ring = chunks in flight (the cell-backward epilogue holds 36 prefetched inputs
        // per lane and its contraction is the attention width: two are enough there)
        constexpr int NB = (MT <= 2 && EPI != EPI_LSTM_BWD) ? 4 : 2;
        constexpr unsigned OOB = 0x80000000u;               // beyond every extent (launch_skinny keeps them < 2^31)
        float *st = stage_all + wave * SROWS * SLD;
        const int lrow = lane >> 3, lk = (lane & 7) * 4;    // coalesced-load coordinates: 8 rows x 128 B per wave load
        // chunks of 32 contraction indices, numbered across the segments; wave w takes chunks w, w + 8, ...
        const int n0 = (p.seg[0].klen + SK_CH - 1) / SK_CH;
        const int n1 = p.nseg > 1 ? (p.seg[1].klen + SK_CH - 1) / SK_CH : 0;
        const int n2 = p.nseg > 2 ? (p.seg[2].klen + SK_CH - 1) / SK_CH : 0;
        const int ntot = n0 + n1 + n2;
        const int mych = ntot > wave ? (ntot - wave + SK_WAVES - 1) / SK_WAVES : 0;
        // Per-segment constants live in SGPRs (readfirstlane pins them there: left as kernel-argument reads the
        // compiler sinks them, as loads, into per-lane control flow around every memory instruction).  The segment of
        // a chunk is a wave-uniform choice made on scalars; the buffer descriptor is built from the chosen pointer (a
        // select between descriptors is lowered to vector code and a waterfall loop).
        auto sg = [](unsigned v) __attribute__((always_inline)) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
        auto sg64 = [sg](const float *ptr) __attribute__((always_inline)) {
            const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
            return ((unsigned long long)sg((unsigned)(a >> 32)) << 32) | sg((unsigned)a);
        };
        const bool has1 = p.nseg > 1, has2 = p.nseg > 2;
        const unsigned long long wp0 = sg64(p.seg[0].w), wp1 = sg64(p.seg[1].w), wp2 = sg64(p.seg[2].w);
        const unsigned long long xp0 = sg64(p.seg[0].x), xp1 = sg64(p.seg[1].x), xp2 = sg64(p.seg[2].x);
        const unsigned wb0 = sg(p.seg[0].wbytes), wb1 = sg(has1 ? p.seg[1].wbytes : 0u), wb2 = sg(has2 ? p.seg[2].wbytes : 0u);
        const unsigned xb0 = sg(p.seg[0].xbytes), xb1 = sg(has1 ? p.seg[1].xbytes : 0u), xb2 = sg(has2 ? p.seg[2].xbytes : 0u);
        const unsigned ldw0 = sg((unsigned)p.seg[0].ldw * 4u), ldw1 = sg((unsigned)p.seg[1].ldw * 4u), ldw2 = sg((unsigned)p.seg[2].ldw * 4u);
        const unsigned ldx0 = sg((unsigned)p.seg[0].ldx * 4u), ldx1 = sg((unsigned)p.seg[1].ldx * 4u), ldx2 = sg((unsigned)p.seg[2].ldx * 4u);
        const unsigned kl0 = sg((unsigned)p.seg[0].klen * 4u), kl1 = sg(has1 ? (unsigned)p.seg[1].klen * 4u : 0u),
                       kl2 = sg(has2 ? (unsigned)p.seg[2].klen * 4u : 0u);
        const int Mrows = (int)sg((unsigned)p.M);
        // rows this lane fetches: weight slots lrow, lrow + 8; batch rows lrow + 8*h
        unsigned wrow[2];
        bool wok[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int sl = lrow + 8 * h;
            if (EPI == EPI_LSTM_FWD) {
                const int u = blockIdx.x * 4 + (sl >> 2);
                wrow[h] = (unsigned)((sl & 3) * p.H + u);
                wok[h] = u < p.H;
            } else {
                wrow[h] = (unsigned)(blockIdx.x * 16 + sl);
                wok[h] = (int)wrow[h] < p.R;
            }
        }
        // (a macro, not a lambda: selects between variables a closure holds by reference become loads from a
        // dynamically indexed closure object, which then lives in scratch memory together with everything it names)
#define SK_ISSUE(i_, gw_, gx_)                                                                          \
    do {                                                                                                \
        const int c_ = wave + (i_) * SK_WAVES;               /* uniform: segment selection is scalar work */ \
        const bool s1_ = c_ >= n0, s2_ = c_ >= n0 + n1;                                                 \
        const int base_ = s2_ ? n0 + n1 : (s1_ ? n0 : 0);                                               \
        const __amdgpu_buffer_rsrc_t rw_ = __builtin_amdgcn_make_buffer_rsrc(                           \
            reinterpret_cast<void *>(s2_ ? wp2 : (s1_ ? wp1 : wp0)), 0, (int)(s2_ ? wb2 : (s1_ ? wb1 : wb0)), 0x00020000); \
        const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc(                           \
            reinterpret_cast<void *>(s2_ ? xp2 : (s1_ ? xp1 : xp0)), 0, (int)(s2_ ? xb2 : (s1_ ? xb1 : xb0)), 0x00020000); \
        const unsigned ldw_ = s2_ ? ldw2 : (s1_ ? ldw1 : ldw0), ldx_ = s2_ ? ldx2 : (s1_ ? ldx1 : ldx0); \
        const unsigned kl_ = s2_ ? kl2 : (s1_ ? kl1 : kl0);                                             \
        const unsigned kb_ = (unsigned)(c_ - base_) * (SK_CH * 4u) + (unsigned)lk * 4u;                 \
        const bool kin_ = kb_ < kl_;   /* klen % 4 == 0: a 16-B piece is all in or all out; c >= ntot lands here too */ \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                              \
            const unsigned off_ = wrow[h_] * ldw_ + kb_;                                                \
            (gw_)[h_] = __builtin_amdgcn_raw_buffer_load_b128(rw_, (kin_ && wok[h_]) ? off_ : OOB, 0, 0); \
        }                                                                                               \
        _Pragma("unroll") for (int h_ = 0; h_ < NX; ++h_) {                                             \
            const unsigned m_ = (unsigned)(lrow + 8 * h_);                                              \
            const unsigned off_ = m_ * ldx_ + kb_;                                                      \
            (gx_)[h_] = __builtin_amdgcn_raw_buffer_load_b128(rx_, (kin_ && (int)m_ < Mrows) ? off_ : OOB, 0, 0); \
        }                                                                                               \
    } while (0)
        // global registers -> the wave's LDS strip -> fragments (LDS instructions of one wave execute in order)
        auto stage = [st, lrow, lk, slot, g](const u32x4 (&gw)[2], const u32x4 (&gx)[NX], f32x4 (&fw)[2], f32x4 (&fx)[MT][2]) __attribute__((always_inline)) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int h = 0; h < 2; ++h)
                *reinterpret_cast<u32x4 *>(st + (lrow + 8 * h) * SLD + lk) = gw[h];
#pragma unroll
            for (int h = 0; h < NX; ++h)
                *reinterpret_cast<u32x4 *>(st + (16 + lrow + 8 * h) * SLD + lk) = gx[h];
            __builtin_amdgcn_wave_barrier();
            fw[0] = *reinterpret_cast<const f32x4 *>(st + slot * SLD + 8 * g);
            fw[1] = *reinterpret_cast<const f32x4 *>(st + slot * SLD + 8 * g + 4);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                fx[mt][0] = *reinterpret_cast<const f32x4 *>(st + (16 + mt * 16 + slot) * SLD + 8 * g);
                fx[mt][1] = *reinterpret_cast<const f32x4 *>(st + (16 + mt * 16 + slot) * SLD + 8 * g + 4);
            }
        };
        auto mfmas = [&acc](const f32x4 (&fw)[2], const f32x4 (&fx)[MT][2]) __attribute__((always_inline)) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt][j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[hh][j], fx[mt][hh][j], acc[mt][j & 1], 0, 0, 0);
        };
        if (mych > 0) {
            u32x4 gw[NB][2], gx[NB][NX];
            f32x4 fw[2][2], fx[2][MT][2];
#pragma unroll
            for (int j = 0; j < NB; ++j) SK_ISSUE(j, gw[j], gx[j]);
            SP_STAMP(1);
            stage(gw[0], gx[0], fw[0], fx[0]);
            SP_STAMP(2);
            SK_ISSUE(NB, gw[0], gx[0]);
            // At the top of iteration i the fragments of chunk i are in registers and chunks i + 1 .. i + NB are in
            // flight / in the ring.  Chunk i + 1 goes through the strip and chunk i + 1 + NB is requested BEFORE the
            // MFMAs of chunk i are issued, so the LDS and memory round trips run under them.
            // (steps written out: a `break` inside an unrolled inner loop keeps the ring indices dynamic and sends the
            // ring to scratch memory)
#define SK_STEP(u)                                                                                      \
    if (i0 + (u) >= mych) break;                                                                        \
    stage(gw[((u) + 1) % NB], gx[((u) + 1) % NB], fw[((u) + 1) & 1], fx[((u) + 1) & 1]);                \
    SK_ISSUE(i0 + (u) + 1 + NB, gw[((u) + 1) % NB], gx[((u) + 1) % NB]);                                \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    mfmas(fw[(u) & 1], fx[(u) & 1]);                                                                    \
    __builtin_amdgcn_sched_barrier(0);
            for (int i0 = 0; i0 < mych; i0 += NB) {
                SK_STEP(0)
                SK_STEP(1)
                if (NB > 2) {
                    SK_STEP(2)
                    SK_STEP(3)
                }
            }
#undef SK_STEP
#undef SK_ISSUE
        }
    } else {
    for (int s = 0; s < p.nseg; ++s) {
        const SkSeg sg = p.seg[s];
        const float *wrow = sg.w + (long)r * sg.ldw;
        const int nch = (sg.klen + SK_CH - 1) / SK_CH;
        // two chunks per trip: all loads of both first, then the MFMAs (two waves share a SIMD, so one
        // wave's loads also overlap the other's matrix work)
        for (int c = wave; c < nch; c += 2 * SK_WAVES) {
            const int k0 = c * SK_CH + 8 * g, k1 = k0 + SK_WAVES * SK_CH;
            float w0[8], w1[8], x0[MT][8], x1[MT][8];
            load8<VEC>(wrow, k0, sg.klen, rvalid, w0);
            load8<VEC>(wrow, k1, sg.klen, rvalid, w1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = mt * 16 + slot;
                load8<VEC>(sg.x + (long)m * sg.ldx, k0, sg.klen, m < p.M, x0[mt]);
                load8<VEC>(sg.x + (long)m * sg.ldx, k1, sg.klen, m < p.M, x1[mt]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[j], x0[mt][j], acc[mt][j & 1], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt][j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[j], x1[mt][j], acc[mt][j & 1], 0, 0, 0);
        }
    }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) red[wave][mt][v][lane] = acc[mt][0][v] + acc[mt][1][v];
    SP_STAMP(3);
    __syncthreads();
    SP_STAMP(4);
    if (tid >= MT * 64) return;
    const int mt = tid >> 6;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < SK_WAVES; ++w) s += red[w][mt][q][lane];
        v[q] = s;
    }
    const int m = mt * 16 + (lane & 15);
    if (m >= p.M) return;
    if (EPI == EPI_STORE || EPI == EPI_TANH_BIAS) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = blockIdx.x * 16 + (lane >> 4) * 4 + q;
            if (rr < p.R) {
                float y = v[q] + e_in[q];
                if (EPI == EPI_TANH_BIAS) y = tanh_fast(y);
                p.out[(long)m * p.ldo + rr] = y;
            }
        }
    } else if (EPI == EPI_LSTM_FWD) {
        const int u = blockIdx.x * 4 + (lane >> 4);
        if (u >= p.H) return;
        const int H = p.H;
        float a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long col = (long)q * H + u;
            (void)col;
            a[q] = v[q] + e_in[q] + e_in[4 + q] + e_in[8 + q];
        }
        float hn;
        if (p.gru) {
            const float gr_ = sigmoid_fast(a[0]), gz = sigmoid_fast(a[1]), gn = tanh_fast(a[2] + gr_ * a[3]);
            hn = (1.f - gz) * gn + gz * e_in[12];
            if (p.gates) {
                float *gr = p.gates + (long)m * 4 * H + u;
                gr[0] = gr_; gr[H] = gz; gr[2 * (long)H] = gn; gr[3 * (long)H] = a[3];
            }
        } else {
            const float gi = sigmoid_fast(a[0]), gf = sigmoid_fast(a[1]), gg = tanh_fast(a[2]),
                        go = sigmoid_fast(a[3]);
            const float cn = gf * e_in[12] + gi * gg;
            hn = go * tanh_fast(cn);
            if (p.gates) {
                float *gr = p.gates + (long)m * 4 * H + u;
                gr[0] = gi; gr[H] = gf; gr[2 * (long)H] = gg; gr[3 * (long)H] = go;
            }
            p.c_new[(long)m * H + u] = cn;
        }
        p.h_new[(long)m * H + u] = hn;
        if (p.h_bm) p.h_bm[(long)m * p.h_bm_ld + u] = hn;
    } else {   // EPI_LSTM_BWD
        const int H = p.H;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = blockIdx.x * 16 + (lane >> 4) * 4 + q;
            if (j >= H) continue;
            float *gr = p.dG + (long)m * 4 * H + j;
            if (p.gru) {
                // h' = (1 - z) n + z h,  n = tanh(n_x + r n_h): saved r, z, n, n_h; bc_prev = h
                const float dh = v[q] + e_in[q * 9 + 0] + e_in[q * 9 + 1] + e_in[q * 9 + 7];
                const float r_ = e_in[q * 9 + 2], z_ = e_in[q * 9 + 3], n_ = e_in[q * 9 + 4], nh = e_in[q * 9 + 5];
                const float dnp = dh * (1.f - z_) * (1.f - n_ * n_);
                gr[0] = dnp * nh * r_ * (1.f - r_);
                gr[H] = dh * (e_in[q * 9 + 8] - n_) * z_ * (1.f - z_);
                gr[2 * (long)H] = dnp;
                gr[3 * (long)H] = dnp * r_;
                p.dc[(long)m * H + j] = dh * z_;
                continue;
            }
            const float dh = v[q] + e_in[q * 9 + 0] + e_in[q * 9 + 1];
            const float gi = e_in[q * 9 + 2], gf = e_in[q * 9 + 3], gg = e_in[q * 9 + 4], go = e_in[q * 9 + 5];
            const float tc = tanh_fast(e_in[q * 9 + 6]);
            const float dct = dh * go * (1.f - tc * tc) + e_in[q * 9 + 7];
            gr[0] = dct * gg * gi * (1.f - gi);
            gr[H] = dct * e_in[q * 9 + 8] * gf * (1.f - gf);
            gr[2 * (long)H] = dct * gi * (1.f - gg * gg);
            gr[3 * (long)H] = dh * tc * go * (1.f - go);
            p.dc[(long)m * H + j] = dct * gf;
        }
    }
    SP_STAMP(9);
}

inline bool al16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <int EPI>
int launch_skinny(const SkArgs &a, hipStream_t s) {
    if (a.M <= 0) return ASRK_OK;
    bool vec = true;
    for (int i = 0; i < a.nseg; ++i)
        vec = vec && al16(a.seg[i].x) && al16(a.seg[i].w) && a.seg[i].ldx % 4 == 0 && a.seg[i].ldw % 4 == 0 &&
              a.seg[i].klen % 4 == 0;
    const int blocks = (EPI == EPI_LSTM_FWD) ? asrk_div_up(a.H, 4) : asrk_div_up(a.R, 16);
    if (blocks <= 0) return ASRK_OK;
    // batch rows beyond 64 run as further passes over the same weights
    for (int m0 = 0; m0 < a.M; m0 += 64) {
        SkArgs p = a;
        p.M = a.M - m0 < 64 ? a.M - m0 : 64;
        for (int i = 0; i < p.nseg; ++i) p.seg[i].x += (long)m0 * p.seg[i].ldx;
        // extents for the buffer loads (bytes behind each base pointer); beyond 2 GiB the scalar-load path runs
        bool small = true;
        const long wrows = (EPI == EPI_LSTM_FWD) ? 4L * p.H : (long)p.R;
        for (int i = 0; i < p.nseg; ++i) {
            const long wb = ((wrows - 1) * p.seg[i].ldw + p.seg[i].klen) * 4L;
            const long xb = ((long)(p.M - 1) * p.seg[i].ldx + p.seg[i].klen) * 4L;
            small = small && wb < (1L << 31) && xb < (1L << 31);
            p.seg[i].wbytes = (unsigned)wb;
            p.seg[i].xbytes = (unsigned)xb;
        }
        for (int i = p.nseg; i < 3; ++i) p.seg[i] = SkSeg{nullptr, nullptr, 0, 0, 0, 0u, 0u};
        if (p.out) p.out += (long)m0 * p.ldo;
        const long H4 = 4L * p.H, H1 = p.H;
        if (p.pre) p.pre += m0 * H4;
        if (p.c_prev) p.c_prev += m0 * H1;
        if (p.c_new) p.c_new += m0 * H1;
        if (p.h_new) p.h_new += m0 * H1;
        if (p.gates) p.gates += m0 * H4;
        if (p.h_bm) p.h_bm += (long)m0 * p.h_bm_ld;
        if (p.add0) p.add0 += (long)m0 * p.ld0;
        if (p.add1) p.add1 += (long)m0 * p.ld1;
        if (p.dG) p.dG += m0 * H4;
        if (p.bc_prev) p.bc_prev += m0 * H1;
        if (p.bc_new) p.bc_new += m0 * H1;
        if (p.dc) p.dc += m0 * H1;
        const int mt = p.M <= 16 ? 1 : (p.M <= 32 ? 2 : 4);
#define SK_LAUNCH_V(MT_, V_)                                                                        \
    do {                                                                                            \
        static AsrkLdsLatch latch_;                                                                 \
        const size_t lds = (size_t)SK_WAVES * (16 + MT_ * 16) * 36 * sizeof(float);                 \
        hipError_t e_ = asrk_max_lds_once(latch_, reinterpret_cast<const void *>(&skinny_kernel<MT_, EPI, V_>), (int)lds); \
        if (e_ != hipSuccess) return (int)e_;                                                       \
        hipLaunchKernelGGL((skinny_kernel<MT_, EPI, V_>), dim3(blocks), dim3(SK_THREADS), lds, s, p); \
    } while (0)
#define SK_LAUNCH(MT_)                                                                              \
    do {                                                                                            \
        if (vec && small) SK_LAUNCH_V(MT_, 2);                                                      \
        else hipLaunchKernelGGL((skinny_kernel<MT_, EPI, 0>), dim3(blocks), dim3(SK_THREADS), 0, s, p); \
    } while (0)
        if (mt == 1) SK_LAUNCH(1);
        else if (mt == 2) SK_LAUNCH(2);
        else SK_LAUNCH(4);
#undef SK_LAUNCH
#undef SK_LAUNCH_V
    }
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

// ------------------------------------------------------------------------------------ F2a: conv + energy
// v where c holds, +0 elsewhere, WITHOUT a select on a loaded value: the compiler sinks a load whose only use is one arm
// of a select into a branch and waits for every outstanding load at the join
__device__ __forceinline__ float maskf(float v, bool c) {
    return __uint_as_float(__float_as_uint(v) & (0u - (unsigned)c));
}
constexpr int RED_RS = 65;   // row pitch of the wave-private transpose tiles ([16][65] floats: conflict-free both ways)

struct AttArgs {
    const float *key, *q, *prev, *Wc, *Wp, *we, *be;
    const int64_t *lens;
    float *conv, *e;
    long prev_ld;
    int Te, A, K, ks, tpb, KP;
    float inv_temp;
    int kvb;   // 1: key / lens have one row per batch entry; 0: one shared utterance (beam search)
    const int *row_mem;   // optional: batch row b attends over memory row row_mem[b] (several utterances' beams)
    unsigned long long *stamps;   // debug timeline slot or nullptr
};

// grid (B, ceil(Te/tpb)), 512 threads
__global__ __launch_bounds__(512) void attend_energy_kernel(AttArgs p) {
    extern __shared__ float sm[];
    const int b = blockIdx.x, t0 = blockIdx.y * p.tpb;
    const int nt = min(p.tpb, p.Te - t0);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int A = p.A, K = p.K, KP = p.KP, KW = 2 * p.ks + 1;
    float *s_prev = sm;                         // [tpb + 2ks]
    float *s_wc = s_prev + p.tpb + 2 * p.ks;    // [K*KW]
    float *s_wp = s_wc + K * KW;                // [A*KP]
    float *s_c = s_wp + A * KP;                 // [tpb*K]
    float *s_q = s_c + p.tpb * K;               // [A]
    float *s_we = s_q + A;                      // [A]
    const int bk = p.row_mem ? p.row_mem[b] : b * p.kvb;
    const int len = min((int)p.lens[bk], p.Te);
    for (int i = tid; i < nt + 2 * p.ks; i += 512) {
        const int t = t0 + i - p.ks;
        s_prev[i] = (t >= 0 && t < p.Te) ? p.prev[(long)b * p.prev_ld + t] : 0.f;
    }
    for (int i = tid; i < K * KW; i += 512) s_wc[i] = p.Wc[i];
    for (int i = tid; i < A * K; i += 512) {
        const int a = i / K, k = i - a * K;
        s_wp[a * KP + k] = p.Wp[i];
    }
    for (int i = tid; i < A; i += 512) {
        s_q[i] = p.q[(long)b * A + i];
        s_we[i] = p.we[i];
    }
    __syncthreads();
    // location features of this block's frames (lanes along time: conflict-free window reads)
    for (int i = tid; i < nt * K; i += 512) {
        const int k = i / nt, tl = i - k * nt;
        const float *wr = s_wc + k * KW, *pr = s_prev + tl;
        float acc = 0.f;
        for (int j = 0; j < KW; ++j) acc += pr[j] * wr[j];
        s_c[tl * K + k] = acc;
        p.conv[((long)b * p.Te + t0 + tl) * K + k] = acc;
    }
    __syncthreads();
    const float be = p.be[0];
    constexpr int FRM = 4, NAM = 5;   // register prefetch covers tpb <= 32 frames, A <= 320
    if (nt <= 8 * FRM && A <= 64 * NAM) {
        // all key values this wave needs are requested up front (independent loads in flight) so the
        // tanh / projection chain of a frame does not wait on one memory round trip per 64 columns
        float kreg[FRM][NAM];
#pragma unroll
        for (int f = 0; f < FRM; ++f) {
            const int tl = wave + 8 * f, t = t0 + tl;
            const bool live = tl < nt && t < len;
            const float *kr = p.key + ((long)bk * p.Te + (live ? t : 0)) * A;
#pragma unroll
            for (int i = 0; i < NAM; ++i) {
                const int a = lane + 64 * i;
                kreg[f][i] = (live && a < A) ? kr[a] : 0.f;
            }
        }
#pragma unroll
        for (int f = 0; f < FRM; ++f) {
            const int tl = wave + 8 * f, t = t0 + tl;
            if (tl >= nt) break;
            if (t >= len) {   // padded frame: never attended (src/module.py:191-193 masked_fill(-inf))
                if (lane == 0) p.e[(long)b * p.Te + t] = -INFINITY;
                continue;
            }
            const float *cr = s_c + tl * K;
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < NAM; ++i) {
                const int a = lane + 64 * i;
                if (a < A) {
                    float u = 0.f;
                    for (int k = 0; k < K; ++k) u += s_wp[a * KP + k] * cr[k];
                    part += s_we[a] * tanh_fast(kreg[f][i] + s_q[a] + tanh_fast(u));
                }
            }
            part = wave_sum(part);
            if (lane == 0) p.e[(long)b * p.Te + t] = (part + be) * p.inv_temp;
        }
        return;
    }
    for (int tl = wave; tl < nt; tl += 8) {
        const int t = t0 + tl;
        if (t >= len) {
            if (lane == 0) p.e[(long)b * p.Te + t] = -INFINITY;
            continue;
        }
        const float *kr = p.key + ((long)bk * p.Te + t) * A;
        const float *cr = s_c + tl * K;
        float part = 0.f;
        for (int a = lane; a < A; a += 64) {
            float u = 0.f;
            for (int k = 0; k < K; ++k) u += s_wp[a * KP + k] * cr[k];
            part += s_we[a] * tanh_fast(kr[a] + s_q[a] + tanh_fast(u));
        }
        part = wave_sum(part);
        if (lane == 0) p.e[(long)b * p.Te + t] = (part + be) * p.inv_temp;
    }
}

// F2a, one memory round trip (round 6).  attend_energy_kernel above is a chain of dependent round trips (staging loops,
// barrier, the location convolution as 160 serial 201-tap sums, barrier, only then the key loads) behind the 5-us
// launch floor: 15.7 us for ~2 us of arithmetic.  Here every global operand is requested before anything waits
// (clamped addresses + masks, cf. energy_bwd_kernel3), and the convolution of a frame is done by the wave that owns
// the frame: lane l multiplies taps l, l + 64, ... of all K filters (filter taps and the frame's window of the
// previous alignment sit in registers, both loaded coalesced straight from global memory), the K partial sums cross
// the lanes through a wave-private LDS tile, and the frame's K location features come back wave-uniform.  One
// workgroup barrier (Wp: global -> registers -> LDS -> the rows each lane multiplies, in registers).
// A <= 64*NA, K <= KM <= 16, 2*ks+1 <= 64*AE_NW, tpb <= 8*AE_FR, A*K <= 512*AE_WPN; otherwise the kernel above runs.
constexpr int AE_FR = 4, AE_NW = 4, AE_WPN = 8;
template <int NA, int KM>
__global__ __launch_bounds__(512) void attend_energy_kernel2(AttArgs p) {
    extern __shared__ float sm[];
    const int b = blockIdx.x, t0 = blockIdx.y * p.tpb;
    const int nt = min(p.tpb, p.Te - t0);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int A = p.A, K = p.K, KP = p.KP, Te = p.Te, ks = p.ks, KW = 2 * ks + 1, AK = A * K;
    float *s_wp = sm;                               // [A*KP]
    float *red_w = s_wp + A * KP + wave * (16 * RED_RS + 16);   // per wave: [16][RED_RS] tile + [16] features
    float *cb_w = red_w + 16 * RED_RS;
    const unsigned kdiv = 0xFFFFFFFFu / (unsigned)K + 1u;
    const int bk = p.row_mem ? p.row_mem[b] : b * p.kvb;
    SP_STAMP(0);

    // ---- one round trip
    const int len_raw = (int)p.lens[bk];
    float kreg[AE_FR][NA], pw[AE_FR][AE_NW];
#pragma unroll
    for (int f = 0; f < AE_FR; ++f) {
        const int t = min(t0 + wave + 8 * f, Te - 1);
        const float *kr = p.key + ((long)bk * Te + t) * A;
#pragma unroll
        for (int j = 0; j < NA; ++j) kreg[f][j] = kr[min(lane + 64 * j, A - 1)];
#pragma unroll
        for (int i = 0; i < AE_NW; ++i) {
            const int tp = t0 + wave + 8 * f - ks + lane + 64 * i;       // frame of tap lane + 64 i
            pw[f][i] = maskf(p.prev[(long)b * p.prev_ld + min(max(tp, 0), Te - 1)],
                             tp >= 0 && tp < Te && lane + 64 * i < KW);
        }
    }
    float wc[KM][AE_NW];
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int i = 0; i < AE_NW; ++i)
            wc[k][i] = maskf(p.Wc[min(k, K - 1) * KW + min(lane + 64 * i, KW - 1)], k < K && lane + 64 * i < KW);
    float wpst[AE_WPN];
#pragma unroll
    for (int r = 0; r < AE_WPN; ++r) wpst[r] = p.Wp[min(tid + 512 * r, AK - 1)];
    float qv[NA], wev[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int a = lane + 64 * j, ac = min(a, A - 1);
        qv[j] = p.q[(long)b * A + ac];
        wev[j] = maskf(p.we[ac], a < A);
    }
    const float be = p.be[0];
    SP_STAMP(1);
    const int len = min(len_raw, Te);
#pragma unroll
    for (int r = 0; r < AE_WPN; ++r) {
        const int i = tid + 512 * r;
        if (i < AK) {
            const int a = (int)__umulhi((unsigned)i, kdiv), k = i - a * K;
            s_wp[a * KP + k] = wpst[r];
        }
    }
    __syncthreads();
    SP_STAMP(2);
    float wp[NA][KM];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int ac = min(lane + 64 * j, A - 1);
#pragma unroll
        for (int k = 0; k < KM; ++k) wp[j][k] = maskf(s_wp[ac * KP + min(k, K - 1)], lane + 64 * j < A && k < K);
    }
    SP_STAMP(3);
#pragma unroll
    for (int f = 0; f < AE_FR; ++f) {
        const int tl = wave + 8 * f, t = t0 + tl;
        if (tl >= nt) break;
        SP_STAMP(4 + f);
        // location features of the frame: c[k] = sum_j prev[t + j - ks] Wc[k, j]
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < AE_NW; ++i) acc += pw[f][i] * wc[k][i];
            red_w[k * RED_RS + lane] = acc;
        }
        __builtin_amdgcn_wave_barrier();
        const int rk = lane & 15, part = lane >> 4;
        float sum = 0.f;
        if (rk < KM) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sum += red_w[rk * RED_RS + part * 16 + i];
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        if (lane < 16) cb_w[lane] = sum;
        if (lane < K) p.conv[((long)b * Te + t) * K + lane] = sum;
        __builtin_amdgcn_wave_barrier();
        if (t >= len) {   // padded frame: never attended (src/module.py:191-193 masked_fill(-inf))
            if (lane == 0) p.e[(long)b * Te + t] = -INFINITY;
            continue;
        }
        float c[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) c[k] = cb_w[k];
        float part_e = 0.f;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            float u = 0.f;
#pragma unroll
            for (int k = 0; k < KM; ++k) u += wp[j][k] * c[k];
            part_e += wev[j] * tanh_fast(kreg[f][j] + qv[j] + tanh_fast(u));
        }
        part_e = wave_sum(part_e);
        if (lane == 0) p.e[(long)b * Te + t] = (part_e + be) * p.inv_temp;
    }
    SP_STAMP(9);
}

// ------------------------------------------------------------------------------------ F2b: softmax + context
struct CtxArgs {
    const float *e, *value;
    float *attn, *ctx;
    long attn_ld, ctx_ld;
    int Te, Dv;
    int kvb;   // as AttArgs::kvb
    const int *row_mem;   // as AttArgs::row_mem
    unsigned long long *stamps;
};

// grid (B, ceil(Dv/256)), 512 threads: wave w takes frames t = w, w+8, ...; lane takes 4 columns
constexpr int CTX_CF = 28;   // frames per wave whose value rows are in flight under the softmax (Te <= 224: all of them)
template <bool VEC>
__global__ __launch_bounds__(512) void softmax_context_kernel(CtxArgs p) {
    extern __shared__ float sm[];   // [Te] weights, then [8][256] partial contexts
    __shared__ float s_red[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Te = p.Te, Dv = p.Dv;
    const int TeP = max((Te + 3) & ~3, 8 * CTX_CF);
    float *s_a = sm, *s_part = sm + TeP;
    const float *er = p.e + (long)b * Te;
    // the value rows of the first 8 * CTX_CF frames are requested BEFORE the softmax (they do not depend on it): the
    // energies' round trip, the three barriers of the softmax and the value stream overlap instead of queueing up
    // (round 5: four dependent batches of eight loads after the softmax, 10.9 us per call)
    SP_STAMP(0);
    const float e0 = er[min(tid, Te - 1)];   // this thread's first (Te <= 512: only) energy, requested ahead of the values
    const int d0 = blockIdx.y * 256 + lane * 4;
    const float *vb = p.value + (long)(p.row_mem ? p.row_mem[b] : b * p.kvb) * Te * Dv + d0;
    f32x4 v0[CTX_CF];
    if (VEC) {
#pragma unroll
        for (int i = 0; i < CTX_CF; ++i)
            v0[i] = *reinterpret_cast<const f32x4 *>(vb - d0 + (long)min(wave + 8 * i, Te - 1) * Dv + min(d0, Dv - 4));
    }
    SP_STAMP(1);
    for (int t = Te + tid; t < TeP; t += 512) s_a[t] = 0.f;
    float mx = tid < Te ? e0 : -INFINITY;
    for (int t = tid + 512; t < Te; t += 512) mx = fmaxf(mx, er[t]);
    mx = wave_max(mx);
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    mx = s_red[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) mx = fmaxf(mx, s_red[w]);
    float sum = 0.f;
    if (tid < Te) {
        const float x = __expf(e0 - mx);      // exp(-inf) = 0 on padded frames
        s_a[tid] = x;
        sum = x;
    }
    for (int t = tid + 512; t < Te; t += 512) {
        const float x = __expf(er[t] - mx);
        s_a[t] = x;
        sum += x;
    }
    sum = wave_sum(sum);
    if (lane == 0) s_red[8 + wave] = sum;
    __syncthreads();
    sum = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) sum += s_red[8 + w];
    const float inv = 1.f / sum;
    for (int t = tid; t < Te; t += 512) {
        const float a = s_a[t] * inv;
        s_a[t] = a;
        if (blockIdx.y == 0) p.attn[(long)b * p.attn_ld + t] = a;
    }
    __syncthreads();
    SP_STAMP(2);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (VEC) {
        if (d0 < Dv) {   // Dv % 4 == 0: the whole float4 is in range
#pragma unroll
            for (int i = 0; i < CTX_CF; ++i) {
                const float a = s_a[wave + 8 * i];           // zero beyond Te
                acc[0] += a * v0[i][0]; acc[1] += a * v0[i][1]; acc[2] += a * v0[i][2]; acc[3] += a * v0[i][3];
            }
            int t = wave + 8 * CTX_CF;
            for (; t + 56 < Te; t += 64) {   // longer memories: 8 independent 16-byte loads in flight per lane
                f32x4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x4 *>(vb + (long)(t + 8 * i) * Dv);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float a = s_a[t + 8 * i];
                    acc[0] += a * v[i][0]; acc[1] += a * v[i][1]; acc[2] += a * v[i][2]; acc[3] += a * v[i][3];
                }
            }
            for (; t < Te; t += 8) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(vb + (long)t * Dv);
                const float a = s_a[t];
                acc[0] += a * v[0]; acc[1] += a * v[1]; acc[2] += a * v[2]; acc[3] += a * v[3];
            }
        }
    } else {
        for (int t = wave; t < Te; t += 8) {
            const float a = s_a[t];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (d0 + j < Dv) acc[j] += a * vb[(long)t * Dv + j];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) s_part[wave * 256 + lane * 4 + j] = acc[j];
    SP_STAMP(3);
    __syncthreads();
    if (tid < 256) {
        const int d = blockIdx.y * 256 + tid;
        if (d < Dv) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += s_part[w * 256 + tid];
            p.ctx[(long)b * p.ctx_ld + d] = s;
        }
    }
    SP_STAMP(9);
}

// F2b for the beams of several utterances (asrk_speller_t::row_group = RG > 0: rows [g*RG, (g+1)*RG) attend over the
// SAME memory row_mem[g*RG]).  softmax_context_kernel reads the whole value memory once per ROW (839 MB of L2 traffic per
// decode position at 32 utterances x 16 hypotheses: 69 us); here a workgroup takes one utterance x 256 columns, each of
// its 8 waves normalises RG/8 rows and then walks the value rows once for them (the waves read the same lines at about
// the same time: L1), so the memory crosses L2 once per utterance.  RG <= 32, Dv % 4 == 0, 16-byte aligned value.
constexpr int CG_MAXR = 4;     // rows per wave
__global__ __launch_bounds__(512) void softmax_context_group_kernel(CtxArgs p, int RG) {
    extern __shared__ float sm[];                  // [RG][TeP] weights
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Te = p.Te, Dv = p.Dv, TeP = (Te + 3) & ~3;
    const int b0 = g * RG;
    const int mem = p.row_mem ? p.row_mem[b0] : b0 * p.kvb;
    const int d0 = blockIdx.y * 256 + lane * 4;
    const float *vb = p.value + (long)mem * Te * Dv + min(d0, Dv - 4);
    int nr = 0;
    int rows[CG_MAXR];
#pragma unroll
    for (int j = 0; j < CG_MAXR; ++j) {
        rows[j] = wave + 8 * j;
        if (rows[j] < RG) nr = j + 1;
    }
    // softmax of this wave's rows (Te <= a few hundred: one pass of loads, everything else in registers / LDS)
#pragma unroll
    for (int j = 0; j < CG_MAXR; ++j) {
        if (j >= nr) break;
        const float *er = p.e + (long)(b0 + rows[j]) * Te;
        float *sa = sm + rows[j] * TeP;
        float mx = -INFINITY;
        for (int t = lane; t < Te; t += 64) mx = fmaxf(mx, er[t]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int t = lane; t < Te; t += 64) {
            const float x = __expf(er[t] - mx);
            sa[t] = x;
            sum += x;
        }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int t = lane; t < Te; t += 64) {
            const float a = sa[t] * inv;
            sa[t] = a;
            if (blockIdx.y == 0) p.attn[(long)(b0 + rows[j]) * p.attn_ld + t] = a;
        }
    }
    __builtin_amdgcn_wave_barrier();               // a wave reads only the rows it wrote
    f32x4 acc[CG_MAXR];
#pragma unroll
    for (int j = 0; j < CG_MAXR; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int t = 0;
    for (; t + 7 < Te; t += 8) {                   // 8 independent 16-byte loads in flight per lane
        f32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x4 *>(vb + (long)(t + i) * Dv);
#pragma unroll
        for (int j = 0; j < CG_MAXR; ++j) {
            if (j >= nr) break;
            const float *sa = sm + rows[j] * TeP + t;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float a = sa[i];
                acc[j][0] += a * v[i][0]; acc[j][1] += a * v[i][1]; acc[j][2] += a * v[i][2]; acc[j][3] += a * v[i][3];
            }
        }
    }
    for (; t < Te; ++t) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(vb + (long)t * Dv);
#pragma unroll
        for (int j = 0; j < CG_MAXR; ++j) {
            if (j >= nr) break;
            const float a = sm[rows[j] * TeP + t];
            acc[j][0] += a * v[0]; acc[j][1] += a * v[1]; acc[j][2] += a * v[2]; acc[j][3] += a * v[3];
        }
    }
    if (d0 < Dv) {
#pragma unroll
        for (int j = 0; j < CG_MAXR; ++j) {
            if (j >= nr) break;
            *reinterpret_cast<f32x4 *>(p.ctx + (long)(b0 + rows[j]) * p.ctx_ld + d0) = acc[j];
        }
    }
}

// ------------------------------------------------------------------------------------ B3: dattn = dctx . value
struct DattnArgs {
    const float *dctx, *value, *extra0, *extra1;
    const int64_t *lens;
    float *dattn;
    long dctx_ld, e0_ld, e1_ld;
    int Te, Dv;
    unsigned long long *stamps;
    int lens_div;   // rows per lens entry - 1 (0 = one row per entry): heads of one utterance share its length
};

// grid (B, ceil(Te/8)), 512 threads: one wave per frame.  Every operand (length, value row, dctx row, the two addends)
// is requested up front - the round-5 form read the length, branched, streamed the row in dependent batches and only
// then fetched the addends: four round trips behind the launch floor for a 16-KiB dot product.
constexpr int DAT_ND = 8;    // 16-byte pieces per lane in flight (Dv <= 2048 in one go)
template <bool VEC>
__global__ __launch_bounds__(512) void dattn_kernel(DattnArgs p) {
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    const int t = blockIdx.y * 8 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (t >= p.Te) return;
    SP_STAMP(0);
    const int len_raw = (int)p.lens[b / (p.lens_div + 1)];
    const float *v = p.value + ((long)b * p.Te + t) * p.Dv;
    const float *g = p.dctx + (long)b * p.dctx_ld;
    float acc = 0.f;
    const float x0 = p.extra0 ? p.extra0[(long)b * p.e0_ld + t] : 0.f;     // wave-uniform pointers: no per-lane branch
    const float x1 = p.extra1 ? p.extra1[(long)b * p.e1_ld + t] : 0.f;
    if (VEC) {
        f32x4 a[DAT_ND], c[DAT_ND];
#pragma unroll
        for (int i = 0; i < DAT_ND; ++i) {
            const int d = min(lane * 4 + 256 * i, p.Dv - 4);
            a[i] = *reinterpret_cast<const f32x4 *>(v + d);
            c[i] = *reinterpret_cast<const f32x4 *>(g + d);
        }
#pragma unroll
        for (int i = 0; i < DAT_ND; ++i)
            acc += maskf(a[i][0] * c[i][0] + a[i][1] * c[i][1] + a[i][2] * c[i][2] + a[i][3] * c[i][3],
                         lane * 4 + 256 * i < p.Dv);
        for (int d = lane * 4 + 256 * DAT_ND; d < p.Dv; d += 256) {
            const f32x4 aa = *reinterpret_cast<const f32x4 *>(v + d);
            const f32x4 cc = *reinterpret_cast<const f32x4 *>(g + d);
            acc += aa[0] * cc[0] + aa[1] * cc[1] + aa[2] * cc[2] + aa[3] * cc[3];
        }
    } else {
        for (int d = lane; d < p.Dv; d += 64) acc += v[d] * g[d];
    }
    SP_STAMP(1);
    acc = wave_sum(acc);
    SP_STAMP(2);
    // beyond the utterance attn is exactly 0: its gradient never matters
    if (lane == 0) p.dattn[(long)b * p.Te + t] = t < min(len_raw, p.Te) ? acc + x0 + x1 : 0.f;
    SP_STAMP(9);
}

// ------------------------------------------------------------------------------------ B4: energy backward
struct EbArgs {
    const float *key, *q, *conv, *Wp, *we, *attn, *dattn;
    const int64_t *lens;
    float *dkey, *dconv, *dq_part, *dwe_part, *dWp_part, *dbe_part;
    long attn_ld;
    int Te, A, K, tpb, KP;
    float inv_temp;
    unsigned long long *stamps;
};

// grid (B, TC), 512 threads.  Phase 1 is elementwise over (frame, a); the small contractions that
// follow (dconv = du Wp, dWp += du^T c, dq = sum_t dz) read du / dz back from LDS, so there are no
// cross-lane reductions and no atomics: every workgroup owns its partial-sum slices across all steps.
__global__ __launch_bounds__(512) void energy_bwd_kernel2(EbArgs p) {
    extern __shared__ float sm[];
    __shared__ float s_red[8];
    const int b = blockIdx.x, chunk = blockIdx.y, TC = gridDim.y, t0 = chunk * p.tpb;
    const int nt = min(p.tpb, p.Te - t0);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int A = p.A, K = p.K, KP = p.KP, Te = p.Te;
    float *s_q = sm;                     // [A]
    float *s_we = s_q + A;               // [A]
    float *s_dwe = s_we + A;             // [A]
    float *s_wp = s_dwe + A;             // [A*KP]
    float *s_c = s_wp + A * KP;          // [tpb*K]
    float *s_de = s_c + p.tpb * K;       // [tpb]
    float *s_du = s_de + p.tpb;          // [tpb*A]
    float *s_dz = s_du + p.tpb * A;      // [tpb*A]
    float *s_ez = s_dz + p.tpb * A;      // [tpb*A] de * z (LDS float atomics for dwe cost 9.5 of 36 us)
    const int len = min((int)p.lens[b], Te);
    for (int i = tid; i < A; i += 512) {
        s_q[i] = p.q[(long)b * A + i];
        s_we[i] = p.we[i];
        s_dwe[i] = 0.f;
    }
    for (int i = tid; i < A * K; i += 512) {
        const int a = i / K, k = i - a * K;
        s_wp[a * KP + k] = p.Wp[i];
    }
    for (int i = tid; i < nt * K; i += 512) s_c[i] = p.conv[((long)b * Te + t0) * K + i];
    // softmax backward needs sum_t attn * dattn over the whole row
    const float *ar = p.attn + (long)b * p.attn_ld, *dr = p.dattn + (long)b * Te;
    float dot = 0.f;
    for (int t = tid; t < len; t += 512) dot += ar[t] * dr[t];
    dot = wave_sum(dot);
    if (lane == 0) s_red[wave] = dot;
    __syncthreads();
    dot = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) dot += s_red[w];
    for (int i = tid; i < nt; i += 512) {
        const int t = t0 + i;
        s_de[i] = (t < len) ? ar[t] * (dr[t] - dot) * p.inv_temp : 0.f;
    }
    __syncthreads();
    const long blk = (long)b * TC + chunk;
    constexpr int WPN = 8;                       // dWp outputs per thread held in registers (A*K <= 4096)
    float old_wp[WPN], old_we = 0.f;
    const bool wp_regs = A * K <= 512 * WPN;
#pragma unroll
    for (int r = 0; r < WPN; ++r) {
        const int i = tid + 512 * r;
        old_wp[r] = (wp_regs && i < A * K) ? p.dWp_part[blk * A * K + i] : 0.f;
    }
    if (tid < A) old_we = p.dwe_part[blk * A + tid];
    // four elements per trip: their key / dkey loads are issued together (a trip per element made the
    // loop one memory round trip per element: 36 us per call at cfg3)
    for (int base = tid; base < nt * A; base += 512 * 4) {
        int tl4[4], a4[4];
        long ki4[4];
        bool ok4[4], live4[4];
        float kv[4], dk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + 512 * u;
            ok4[u] = i < nt * A;
            tl4[u] = ok4[u] ? i / A : 0;
            a4[u] = ok4[u] ? i - tl4[u] * A : 0;
            live4[u] = ok4[u] && (t0 + tl4[u] < len);
            ki4[u] = ((long)b * Te + t0 + tl4[u]) * A + a4[u];
            kv[u] = live4[u] ? p.key[ki4[u]] : 0.f;
            dk[u] = live4[u] ? p.dkey[ki4[u]] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!ok4[u]) continue;
            const int i = base + 512 * u, tl = tl4[u], a = a4[u];
            float dz = 0.f, du = 0.f, ez = 0.f;
            if (live4[u]) {
                const float de = s_de[tl];
                const float *cr = s_c + tl * K;
                float uu = 0.f;
                for (int k = 0; k < K; ++k) uu += s_wp[a * KP + k] * cr[k];
                const float loc = tanh_fast(uu);
                const float z = tanh_fast(kv[u] + s_q[a] + loc);
                dz = de * s_we[a] * (1.f - z * z);
                du = dz * (1.f - loc * loc);
                p.dkey[ki4[u]] = dk[u] + dz;
                ez = de * z;
            }
            s_dz[i] = dz;
            s_du[i] = du;
            s_ez[i] = ez;
        }
    }
    __syncthreads();
    // dconv[t,k] = sum_a du[t,a] Wp[a,k]
    for (int i = tid; i < nt * K; i += 512) {
        const int tl = i / K, k = i - tl * K;
        const float *dur = s_du + tl * A;
        float acc = 0.f;
        for (int a = 0; a < A; ++a) acc += dur[a] * s_wp[a * KP + k];
        p.dconv[((long)b * Te + t0 + tl) * K + k] = acc;
    }
    // dWp[a,k] += sum_t du[t,a] c[t,k]   (this workgroup's slice)
    if (wp_regs) {
#pragma unroll
        for (int r = 0; r < WPN; ++r) {
            const int i = tid + 512 * r;
            if (i < A * K) {
                const int a = i / K, k = i - a * K;
                float acc = 0.f;
                for (int tl = 0; tl < nt; ++tl) acc += s_du[tl * A + a] * s_c[tl * K + k];
                p.dWp_part[blk * A * K + i] = old_wp[r] + acc;
            }
        }
    } else {
        for (int i = tid; i < A * K; i += 512) {
            const int a = i / K, k = i - a * K;
            float acc = 0.f;
            for (int tl = 0; tl < nt; ++tl) acc += s_du[tl * A + a] * s_c[tl * K + k];
            p.dWp_part[blk * A * K + i] += acc;
        }
    }
    for (int a = tid; a < A; a += 512) {
        float acc = 0.f, ew = 0.f;
        for (int tl = 0; tl < nt; ++tl) {
            acc += s_dz[tl * A + a];
            ew += s_ez[tl * A + a];
        }
        p.dq_part[blk * A + a] = acc;
        p.dwe_part[blk * A + a] = (a == tid ? old_we : p.dwe_part[blk * A + a]) + ew;
    }
    if (tid == 0) {
        float acc = 0.f;
        for (int tl = 0; tl < nt; ++tl) acc += s_de[tl];
        p.dbe_part[blk] += acc;
    }
}

// B4, wave-per-frame form (round 6).  Same contract as energy_bwd_kernel2 (same arguments, same partial-sum slices),
// different mapping and ONE memory round trip: a call of kernel2 is 26 us for ~2 us of arithmetic because it is a chain
// of dependent round trips (lens -> row dot product -> staging loops -> barrier -> key / dkey -> ...), each ~1-2 us
// behind a 5-us launch floor.  Here
//   * every global operand of the workgroup is REQUESTED before anything waits (clamped addresses, masks applied to the
//     values afterwards - no load depends on the utterance length or on another load, and no `cond ? *p : 0`, which
//     the compiler turns into a branch with s_waitcnt vmcnt(0) at the join);
//   * wave w owns frames w, w + 8, ... of the chunk and lane l owns attention columns l, l + 64, ...: the rows of Wp a
//     lane needs sit in registers (kernel2 read K of them from LDS per element, behind an integer division per
//     element), the location features of a frame are K wave-uniform LDS reads per frame;
//   * dconv[t,:] = du[t,:] Wp is finished inside the wave (per-lane partial products, one transpose through a
//     wave-private LDS tile, two cross-lane adds) - kernel2 ran nt*K serial 300-term sums on 160 threads;
//   * sum_t dz, sum_t de*z stay in registers across the wave's frames; one cross-wave add at the end.
// A <= 64*NA, K <= KM, tpb <= 8*EB_FR, tpb*KM <= 512, A*K <= 512*EB_WPN, Te <= 512*EB_NTE; otherwise kernel2 runs.
constexpr int EB_FR = 4, EB_WPN = 8, EB_NTE = 2;
template <int NA, int KM>
__global__ __launch_bounds__(512) void energy_bwd_kernel3(EbArgs p) {
    extern __shared__ float sm[];
    __shared__ float s_red8[8], s_be8[8];
    constexpr int AP = 64 * NA;
    const int b = blockIdx.x, chunk = blockIdx.y, TC = gridDim.y, t0 = chunk * p.tpb;
    const int nt = min(p.tpb, p.Te - t0);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int A = p.A, K = p.K, KP = p.KP, Te = p.Te, AK = A * K;
    float *s_wp = sm;                       // [A*KP]
    float *s_c = s_wp + A * KP;             // [tpb][KM]  zero beyond K
    float *s_du = s_c + p.tpb * KM;         // [tpb][AP]
    float *s_red = s_du + p.tpb * AP;       // [8][16][RED_RS]; afterwards [8][2][AP]
    const long blk = (long)b * TC + chunk;
    const unsigned kdiv = 0xFFFFFFFFu / (unsigned)K + 1u;   // i / K == umulhi(i, kdiv) for i < 2^28

    SP_STAMP(0);
    // ---- one round trip: everything this workgroup reads from memory
    const int len_raw = (int)p.lens[b];
    float kv[EB_FR][NA], dk[EB_FR][NA];
#pragma unroll
    for (int f = 0; f < EB_FR; ++f) {
        const int t = min(t0 + wave + 8 * f, Te - 1);
        const long row = ((long)b * Te + t) * A;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const long idx = row + min(lane + 64 * j, A - 1);
            kv[f][j] = p.key[idx];
            dk[f][j] = p.dkey[idx];
        }
    }
    const float *ar = p.attn + (long)b * p.attn_ld, *dr = p.dattn + (long)b * Te;
    float a_e[EB_NTE], d_e[EB_NTE], a_f[EB_FR], d_f[EB_FR];
#pragma unroll
    for (int e = 0; e < EB_NTE; ++e) {
        const int t = min(tid + 512 * e, Te - 1);
        a_e[e] = ar[t];
        d_e[e] = dr[t];
    }
#pragma unroll
    for (int f = 0; f < EB_FR; ++f) {
        const int t = min(t0 + wave + 8 * f, Te - 1);
        a_f[f] = ar[t];
        d_f[f] = dr[t];
    }
    float wpst[EB_WPN], old_wp[EB_WPN];
#pragma unroll
    for (int r = 0; r < EB_WPN; ++r) {
        const int i = min(tid + 512 * r, AK - 1);
        wpst[r] = p.Wp[i];
        old_wp[r] = p.dWp_part[blk * AK + i];
    }
    const float old_we = p.dwe_part[blk * A + min(tid, A - 1)];
    const float old_be = p.dbe_part[blk];
    float cst;
    {
        const int tl = tid / KM, k = tid - tl * KM;
        cst = maskf(p.conv[((long)b * Te + min(t0 + tl, Te - 1)) * K + min(k, K - 1)], tl < nt && k < K);
    }
    float qv[NA], wev[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int a = lane + 64 * j, ac = min(a, A - 1);
        qv[j] = p.q[(long)b * A + ac];
        wev[j] = maskf(p.we[ac], a < A);
    }
    SP_STAMP(1);
    // ---- staging (registers -> LDS) and the row dot product sum_t attn * dattn of the softmax backward
    const int len = min(len_raw, Te);
    float dot = 0.f;
#pragma unroll
    for (int e = 0; e < EB_NTE; ++e) dot += maskf(a_e[e] * d_e[e], tid + 512 * e < len);
    dot = wave_sum(dot);
    if (lane == 0) s_red8[wave] = dot;
#pragma unroll
    for (int r = 0; r < EB_WPN; ++r) {
        const int i = tid + 512 * r;
        if (i < AK) {
            const int a = (int)__umulhi((unsigned)i, kdiv), k = i - a * K;
            s_wp[a * KP + k] = wpst[r];
        }
    }
    if (tid < p.tpb * KM) s_c[tid] = cst;
    __syncthreads();
    SP_STAMP(2);
    dot = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) dot += s_red8[w];
    float de_f[EB_FR], de_sum = 0.f;
#pragma unroll
    for (int f = 0; f < EB_FR; ++f) {
        const int tl = wave + 8 * f;
        de_f[f] = maskf(a_f[f] * (d_f[f] - dot) * p.inv_temp, tl < nt && t0 + tl < len);
        de_sum += de_f[f];
    }
    if (lane == 0) s_be8[wave] = de_sum;
    float wp[NA][KM];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int ac = min(lane + 64 * j, A - 1);
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            const float v = s_wp[ac * KP + min(k, K - 1)];
            wp[j][k] = (lane + 64 * j < A && k < K) ? v : 0.f;
        }
    }

    SP_STAMP(3);
    // ---- phase 1: the wave's frames
    float dq_acc[NA], dwe_acc[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) dq_acc[j] = dwe_acc[j] = 0.f;
    float *red_w = s_red + wave * 16 * RED_RS;
#pragma unroll
    for (int f = 0; f < EB_FR; ++f) {
        const int tl = wave + 8 * f, t = t0 + tl;
        if (tl >= nt) break;
        const bool live = t < len;
        const float de = de_f[f];
        float c[KM], pk[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            c[k] = s_c[tl * KM + k];
            pk[k] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int a = lane + 64 * j;
            float uu = 0.f;
#pragma unroll
            for (int k = 0; k < KM; ++k) uu += wp[j][k] * c[k];
            const float loc = tanh_fast(uu);
            const float z = tanh_fast(kv[f][j] + qv[j] + loc);
            const float dz = de * wev[j] * (1.f - z * z);          // 0 beyond A (wev) and beyond len (de)
            const float du = dz * (1.f - loc * loc);
            if (live && a < A) p.dkey[((long)b * Te + t) * A + a] = dk[f][j] + dz;
            dq_acc[j] += dz;
            dwe_acc[j] += (a < A) ? de * z : 0.f;
            s_du[tl * AP + a] = du;
#pragma unroll
            for (int k = 0; k < KM; ++k) pk[k] += du * wp[j][k];
        }
        // dconv[t,k] = sum over the 64 lanes of pk[k]: transpose through the wave's tile, 16 + 2 adds per lane
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < KM; ++k) red_w[k * RED_RS + lane] = pk[k];
        __builtin_amdgcn_wave_barrier();
        const int rk = lane & 15, part = lane >> 4;
        float sum = 0.f;
        if (rk < KM) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sum += red_w[rk * RED_RS + part * 16 + i];
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        if (lane < K) p.dconv[((long)b * Te + t) * K + lane] = sum;
    }
    SP_STAMP(4);
    __syncthreads();
    SP_STAMP(5);

    // ---- sums over the chunk's frames
    float *s_x = s_red;                      // [8][2][AP]
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        s_x[(wave * 2 + 0) * AP + lane + 64 * j] = dq_acc[j];
        s_x[(wave * 2 + 1) * AP + lane + 64 * j] = dwe_acc[j];
    }
    // dWp[a,k] += sum_t du[t,a] c[t,k]   (this workgroup's slice)
#pragma unroll
    for (int r = 0; r < EB_WPN; ++r) {
        const int i = tid + 512 * r;
        if (i < AK) {
            const int a = (int)__umulhi((unsigned)i, kdiv), k = i - a * K;
            float acc = 0.f;
            for (int tl = 0; tl < nt; ++tl) acc += s_du[tl * AP + a] * s_c[tl * KM + k];
            p.dWp_part[blk * AK + i] = old_wp[r] + acc;
        }
    }
    SP_STAMP(6);
    __syncthreads();
    if (tid < A) {
        float dq = 0.f, dwe = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            dq += s_x[(w * 2 + 0) * AP + tid];
            dwe += s_x[(w * 2 + 1) * AP + tid];
        }
        p.dq_part[blk * A + tid] = dq;
        p.dwe_part[blk * A + tid] = old_we + dwe;
    }
    if (tid == 0) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) acc += s_be8[w];
        p.dbe_part[blk] = old_be + acc;
    }
    SP_STAMP(9);
}

// ------------------------------------------------------------------------------------ dot-product attention (round 6)
// ScaleDotAttention.forward (src/module.py:204-212) inside the one-node loop, for any number of heads: rows r = b * N + n
// of key [B*N,Te,A] / q [B*N,A]; e[r,t] = q[r,:] . key[r,t,:] / temperature, -inf beyond the utterance (lens[r / N]).
struct DotArgs {
    const float *key, *q, *attn, *dattn;
    const int64_t *lens;
    float *e, *dkey, *dq_part;
    long attn_ld;
    int Te, A, lens_div, tpb;
    float inv_temp;
};

// grid (B*N, ceil(Te/8)), 512 threads: one wave per frame
__global__ __launch_bounds__(512) void dot_energy_kernel(DotArgs p) {
    const int r = blockIdx.x, lane = threadIdx.x & 63;
    const int t = blockIdx.y * 8 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (t >= p.Te) return;
    const int len_raw = (int)p.lens[r / p.lens_div];
    const float *k = p.key + ((long)r * p.Te + t) * p.A, *q = p.q + (long)r * p.A;
    float acc = 0.f;
    for (int a = lane; a < p.A; a += 64) acc += k[a] * q[a];
    acc = wave_sum(acc);
    if (lane == 0) p.e[(long)r * p.Te + t] = t < min(len_raw, p.Te) ? acc * p.inv_temp : -INFINITY;
}

// grid (B*N, TC), 512 threads: softmax backward (row dot product recomputed per workgroup) -> de; the wave that owns a
// frame adds de * q to dkey[r,t,:] and keeps de * key[r,t,:] for dq; the eight waves' sums meet in LDS:
// dq_part[(r * TC + chunk), :] (reduced over the chunks by conv_bwd_kernel's dq role)
constexpr int DOT_NA = 8;   // A <= 512
__global__ __launch_bounds__(512) void dot_energy_bwd_kernel(DotArgs p) {
    __shared__ float s_red8[8];
    __shared__ float s_dq[8][64 * DOT_NA];
    const int r = blockIdx.x, chunk = blockIdx.y, TC = gridDim.y, t0 = chunk * p.tpb;
    const int nt = min(p.tpb, p.Te - t0);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int A = p.A, Te = p.Te;
    const int len = min((int)p.lens[r / p.lens_div], Te);
    const float *ar = p.attn + (long)r * p.attn_ld, *dr = p.dattn + (long)r * Te;
    float dot = 0.f;
    for (int t = tid; t < len; t += 512) dot += ar[t] * dr[t];
    dot = wave_sum(dot);
    if (lane == 0) s_red8[wave] = dot;
    __syncthreads();
    dot = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) dot += s_red8[w];
    float qv[DOT_NA], dq[DOT_NA];
#pragma unroll
    for (int j = 0; j < DOT_NA; ++j) {
        const int a = lane + 64 * j;
        qv[j] = a < A ? p.q[(long)r * A + a] : 0.f;
        dq[j] = 0.f;
    }
    for (int tl = wave; tl < nt; tl += 8) {
        const int t = t0 + tl;
        if (t >= len) continue;
        const float de = ar[t] * (dr[t] - dot) * p.inv_temp;
        const long row = ((long)r * Te + t) * A;
#pragma unroll
        for (int j = 0; j < DOT_NA; ++j) {
            const int a = lane + 64 * j;
            if (a < A) {
                dq[j] += de * p.key[row + a];
                p.dkey[row + a] += de * qv[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < DOT_NA; ++j) s_dq[wave][lane + 64 * j] = dq[j];
    __syncthreads();
    for (int a = tid; a < A; a += 512) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) acc += s_dq[w][a];
        p.dq_part[((long)r * TC + chunk) * A + a] = acc;
    }
}

// ------------------------------------------------------------------------------------ B5: conv backward + dq_pre
struct CbArgs {
    const float *dconv, *prev, *Wc, *dq_part, *q;
    float *dprev, *dWc_part, *dq_pre;
    long prev_ld;
    int Te, K, ks, TC, A, nT, want_dprev;
    unsigned long long *stamps;
    int dq_only;   // dot-product attention: grid (rows, 1), only the dq_pre role
};

// grid (B, nT + K + 1), 256 threads.  y < nT: d prev_att for 64 frames (4 waves split the kernels k);
// nT <= y < nT + K: kernel (y - nT)'s taps of this utterance's slice of the filter gradient (one
// workgroup per utterance for all K * (2ks+1) taps was LDS-bandwidth bound: 11 of 25 us on 32 CUs);
// y == nT + K: dq_pre = (sum_chunks dq) (1 - q^2)
__global__ __launch_bounds__(256) void conv_bwd_kernel(CbArgs p) {
    extern __shared__ float sm[];
    const int b = blockIdx.x, y = p.dq_only ? p.nT + p.K : blockIdx.y, tid = threadIdx.x;
    const int Te = p.Te, K = p.K, ks = p.ks, KW = 2 * ks + 1;
    SP_STAMP(0);
    if (y < p.nT) {
        if (!p.want_dprev) return;
        const int s0 = y * 64, span = 64 + 2 * ks;
        float *sd = sm, *part = sm + K * span;   // [K][span] window of dconv, [4][64] partial sums
        float *s_w = part + 256;                 // [K][KW] filter taps (wave-uniform reads)
        for (int i = tid; i < span * K; i += 256) {
            const int o = i / K, k = i - o * K;
            const int t = s0 + o - ks;
            sd[k * span + o] = (t >= 0 && t < Te) ? p.dconv[((long)b * Te + t) * K + k] : 0.f;
        }
        for (int i = tid; i < K * KW; i += 256) s_w[i] = p.Wc[i];
        __syncthreads();
        const int sl = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
        float acc = 0.f;
        for (int k = w; k < K; k += 4) {
            const float *wr = s_w + k * KW;
            const float *col = sd + k * span + sl + 2 * ks;   // t = s - j + ks
            for (int j = 0; j < KW; ++j) acc += col[-j] * wr[j];
        }
        part[w * 64 + sl] = acc;
        __syncthreads();
        if (tid < 64 && s0 + tid < Te)
            p.dprev[(long)b * Te + s0 + tid] = (part[tid] + part[64 + tid]) + (part[128 + tid] + part[192 + tid]);
        SP_STAMP(9);
    } else if (y < p.nT + K) {
        const int k = y - p.nT;
        float *s_dc = sm, *s_pv = sm + Te;   // [Te], [Te + 2ks]
        float old[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = tid + 256 * r;
            old[r] = j < KW ? p.dWc_part[((long)b * K + k) * KW + j] : 0.f;
        }
        for (int t = tid; t < Te; t += 256) s_dc[t] = p.dconv[((long)b * Te + t) * K + k];
        for (int i = tid; i < Te + 2 * ks; i += 256) {
            const int t = i - ks;
            s_pv[i] = (t >= 0 && t < Te) ? p.prev[(long)b * p.prev_ld + t] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = tid + 256 * r;
            if (j < KW) {
                const float *pv = s_pv + j;   // prev[t + j - ks]
                float acc = 0.f;
                for (int t = 0; t < Te; ++t) acc += s_dc[t] * pv[t];
                p.dWc_part[((long)b * K + k) * KW + j] = old[r] + acc;
            }
        }
        for (int j = tid + 1024; j < KW; j += 256) {   // very wide filters
            const float *pv = s_pv + j;
            float acc = 0.f;
            for (int t = 0; t < Te; ++t) acc += s_dc[t] * pv[t];
            p.dWc_part[((long)b * K + k) * KW + j] += acc;
        }
    } else {
        for (int a = tid; a < p.A; a += 256) {
            float acc = 0.f;
            for (int c = 0; c < p.TC; ++c) acc += p.dq_part[((long)b * p.TC + c) * p.A + a];
            const float qv = p.q[(long)b * p.A + a];
            p.dq_pre[(long)b * p.A + a] = acc * (1.f - qv * qv);
        }
        SP_STAMP(9);
    }
}

// dvalue[b,t',d] = sum_l attn[b,l,t'] * dctx[l,b,d]: the encoder-memory gradient of the whole loop in one
// pass (the reference's autograd sums L rank-1 updates of [B,Te,Dv]).  grid (B, ceil(Dv/256)), 256
// threads: a thread owns one feature column, keeps the 64 dctx values of the current step chunk in
// registers and walks the frames; the attention rows sit in LDS (wave-uniform broadcast reads).
constexpr int DV_LC = 64;
__global__ __launch_bounds__(256) void dvalue_kernel(const float *__restrict__ attn, long attn_ld,
                                                     long attn_step, const float *__restrict__ dxh,
                                                     long step_ld, long row_ld, float *__restrict__ dvalue,
                                                     int L, int Te, int Dv) {
    extern __shared__ float s_at[];   // [DV_LC][Te]
    const int b = blockIdx.x, d = blockIdx.y * 256 + threadIdx.x;
    for (int l0 = 0; l0 < L; l0 += DV_LC) {
        const int nl = min(DV_LC, L - l0);
        __syncthreads();
        for (int i = threadIdx.x; i < nl * Te; i += 256) {
            const int l = i / Te, t = i - l * Te;
            s_at[i] = attn[(long)b * attn_ld + (long)(l0 + l) * attn_step + t];
        }
        for (int i = nl * Te + threadIdx.x; i < DV_LC * Te; i += 256) s_at[i] = 0.f;
        __syncthreads();
        if (d < Dv) {
            float g[DV_LC];
#pragma unroll
            for (int l = 0; l < DV_LC; ++l)
                g[l] = l < nl ? dxh[(long)(l0 + l) * step_ld + (long)b * row_ld + d] : 0.f;
            float *out = dvalue + (long)b * Te * Dv + d;
            for (int t = 0; t < Te; ++t) {
                float acc = 0.f;
#pragma unroll
                for (int l = 0; l < DV_LC; ++l) acc += s_at[l * Te + t] * g[l];
                if (l0 == 0) out[(long)t * Dv] = acc;
                else out[(long)t * Dv] += acc;
            }
        }
    }
}

// joint CTC / attention / LM token scores of one beam step (src/decode.py:123-148), one row per live
// hypothesis: out = (1-w) att + w hack, hack = psi - prev_ctc on the CTC candidates and LOG_ZERO
// elsewhere; out[:,0] = LOG_ZERO (<sos>); out += lm_w * lm.  Same operation order (and roundings) as
// the reference's tensor expression.  grid (n), 256 threads.
__global__ __launch_bounds__(256) void joint_score_kernel(const float *__restrict__ att,
                                                          const int64_t *__restrict__ cand,
                                                          const float *__restrict__ psi,
                                                          const float *__restrict__ prev_ctc,
                                                          const float *__restrict__ lm, float *__restrict__ out,
                                                          int V, int C, float wa, float w_ctc, float w_lm,
                                                          float logzero) {
    const int r = blockIdx.x;
    const float *a = att + (long)r * V;
    const float *l = lm ? lm + (long)r * V : nullptr;
    float *o = out + (long)r * V;
    const bool ctc = cand != nullptr;
    const float wz = w_ctc * logzero;
    for (int v = threadIdx.x; v < V; v += 256) {
        float x = a[v];
        if (ctc) x = (v == 0) ? logzero : wa * x + wz;
        if (l) x = x + w_lm * l[v];
        o[v] = x;
    }
    if (!ctc) return;
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const int v = (int)cand[(long)r * C + c];
        if (v <= 0 || v >= V) continue;          // <sos> stays LOG_ZERO
        float x = wa * a[v] + w_ctc * (psi[(long)r * C + c] - prev_ctc[r]);
        if (l) x = x + w_lm * l[v];
        o[v] = x;
    }
}

// out[c][r] = in[r][c]; grid (ceil(cols/32), ceil(rows/32)), 256 threads (32 x 8)
__global__ __launch_bounds__(256) void transpose_ld_kernel(const float *__restrict__ in, long ldi,
                                                           float *__restrict__ out, long ldo, int rows,
                                                           int cols) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = in[(long)(r0 + i) * ldi + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) out[(long)(c0 + i) * ldo + r0 + tx] = tile[tx][i];
}

int pick_tpb(int B, int Te, int A, int K, size_t fixed_floats, size_t per_frame_floats, size_t budget, int wg_target = 512) {
    int tc = asrk_div_up(wg_target, B > 0 ? B : 1);      // ~2 workgroups per CU
    if (tc > asrk_div_up(Te, 8)) tc = asrk_div_up(Te, 8);
    if (tc < 1) tc = 1;
    int tpb = asrk_div_up(Te, tc);
    if (tpb > 8) tpb = (tpb + 7) / 8 * 8;                // 8 waves take one frame each per round
    while (tpb > 1 && (fixed_floats + per_frame_floats * tpb) * sizeof(float) > budget) tpb = (tpb + 1) / 2;
    if ((fixed_floats + per_frame_floats * tpb) * sizeof(float) > budget) return 0;
    return tpb;
}

constexpr size_t LDS_BUDGET = 150 * 1024;

struct Plan {
    int tpb_f, tc_f, tpb_b, tc_b, KP, nT;
    size_t lds_f, lds_b, lds_ctx, lds_cb;
    int eb3_na, eb3_km;   // energy_bwd_kernel3 instantiation (0: shape not covered, kernel2 runs)
    size_t lds_b3;
    int ae2_na, ae2_km;   // attend_energy_kernel2 instantiation (0: shape not covered)
    size_t lds_f2;
};

int make_plan(const asrk_speller_t &d, Plan &pl) {
    if (d.att_mode == 1) {   // dot-product energies: no LDS plan; the backward kernel takes 32 frames per workgroup
        pl = Plan{};
        pl.tpb_f = pl.tpb_b = 32;
        pl.tc_f = pl.tc_b = asrk_div_up(d.Te, 32);
        pl.KP = 1;
        pl.nT = asrk_div_up(d.Te, 64);
        pl.lds_ctx = ((size_t)std::max((d.Te + 3) & ~3, 8 * CTX_CF) + 8 * 256) * sizeof(float);
        pl.lds_cb = 256;
        return pl.lds_ctx > LDS_BUDGET ? ASRK_ESHAPE : ASRK_OK;
    }
    const int KW = 2 * d.ks + 1;
    pl.KP = (d.K % 2 == 0) ? d.K + 1 : d.K;
    // forward: prev window [tpb + 2ks] + Wc + Wp + c chunk + q + we
    const size_t fix_f = (size_t)2 * d.ks + (size_t)d.K * KW + (size_t)d.A * pl.KP + 2 * (size_t)d.A;
    // the one-round-trip kernels (attend_energy_kernel2, energy_bwd_kernel3) hold a lane's rows of Wp in registers
    // (160-190 VGPRs: one workgroup per CU) and pay their prologue per workgroup: ONE round of fat workgroups
    // (<= 256, up to 4 frames per wave) - measured 13.8 / 20.8 us at 224 workgroups against 17.5 / 27.5 us at 416 and
    // 25.6 / 40.6 us at 800 (profiles/r06_speller_grid.log); the staged kernels keep ~2 workgroups per CU (16.6 / 26.6)
    const bool thin = d.A <= 320 && d.K <= 16 && d.A * d.K <= 512 * AE_WPN && 2 * d.ks + 1 <= 64 * AE_NW &&
                      d.Te <= 512 * EB_NTE;
    const int wg_dflt = thin ? 256 : 512;
    pl.tpb_f = pick_tpb(d.B, d.Te, d.A, d.K, fix_f, 1 + (size_t)d.K, LDS_BUDGET, wg_dflt);
    if (pl.tpb_f <= 0) return ASRK_ESHAPE;
    pl.tc_f = asrk_div_up(d.Te, pl.tpb_f);
    pl.lds_f = (fix_f + (size_t)(1 + d.K) * pl.tpb_f) * sizeof(float);
    const size_t fix_b = 3 * (size_t)d.A + (size_t)d.A * pl.KP;
    pl.tpb_b = pick_tpb(d.B, d.Te, d.A, d.K, fix_b, (size_t)d.K + 1 + 3 * (size_t)d.A, LDS_BUDGET, wg_dflt);
    if (pl.tpb_b <= 0) return ASRK_ESHAPE;
    pl.tc_b = asrk_div_up(d.Te, pl.tpb_b);
    pl.lds_b = (fix_b + ((size_t)d.K + 1 + 3 * (size_t)d.A) * pl.tpb_b) * sizeof(float);
    pl.ae2_na = pl.ae2_km = 0;
    pl.lds_f2 = 0;
    if (d.A <= 320 && d.K <= 16 && 2 * d.ks + 1 <= 64 * AE_NW && pl.tpb_f <= 8 * AE_FR && d.A * d.K <= 512 * AE_WPN) {
        pl.ae2_na = d.A <= 128 ? 2 : 5;
        pl.ae2_km = d.K <= 12 ? 12 : 16;
        pl.lds_f2 = ((size_t)d.A * pl.KP + 8 * (16 * RED_RS + 16)) * sizeof(float);
    }
    pl.eb3_na = pl.eb3_km = 0;
    pl.lds_b3 = 0;
    if (d.A <= 320 && d.K <= 16 && pl.tpb_b <= 8 * EB_FR && d.A * d.K <= 512 * EB_WPN && d.Te <= 512 * EB_NTE &&
        pl.tpb_b * (d.K <= 12 ? 12 : 16) <= 512) {
        pl.eb3_na = d.A <= 128 ? 2 : 5;
        pl.eb3_km = d.K <= 12 ? 12 : 16;
        pl.lds_b3 = ((size_t)d.A * pl.KP + (size_t)pl.tpb_b * (pl.eb3_km + 64 * pl.eb3_na) + 8 * 16 * RED_RS) * sizeof(float);
        if (pl.lds_b3 > LDS_BUDGET) pl.eb3_na = 0;
    }
    pl.lds_ctx = ((size_t)std::max((d.Te + 3) & ~3, 8 * CTX_CF) + 8 * 256) * sizeof(float);
    pl.nT = asrk_div_up(d.Te, 64);
    const size_t cb_data = ((size_t)d.K * (64 + 2 * d.ks) + 256 + (size_t)d.K * KW) * sizeof(float);
    const size_t cb_w = ((size_t)2 * d.Te + 2 * d.ks) * sizeof(float);
    pl.lds_cb = cb_data > cb_w ? cb_data : cb_w;
    if (pl.lds_ctx > LDS_BUDGET || pl.lds_cb > LDS_BUDGET) return ASRK_ESHAPE;
    return ASRK_OK;
}

static void launch_attend_energy(const AttArgs &a, const Plan &pl, int B, hipStream_t s) {
    const dim3 grid(B, pl.tc_f);
    if (!pl.ae2_na) hipLaunchKernelGGL(attend_energy_kernel, grid, dim3(512), pl.lds_f, s, a);
    else if (pl.ae2_na == 2 && pl.ae2_km == 12) hipLaunchKernelGGL((attend_energy_kernel2<2, 12>), grid, dim3(512), pl.lds_f2, s, a);
    else if (pl.ae2_na == 2) hipLaunchKernelGGL((attend_energy_kernel2<2, 16>), grid, dim3(512), pl.lds_f2, s, a);
    else if (pl.ae2_km == 12) hipLaunchKernelGGL((attend_energy_kernel2<5, 12>), grid, dim3(512), pl.lds_f2, s, a);
    else hipLaunchKernelGGL((attend_energy_kernel2<5, 16>), grid, dim3(512), pl.lds_f2, s, a);
}


int check_dims(const asrk_speller_t *d) {
    if (!d) return ASRK_EINVAL;
    if (d->B < 0 || d->Te <= 0 || d->A <= 0 || d->Dv <= 0 || (d->K <= 0 && d->att_mode == 0) || d->K < 0 ||
        d->ks < 0 || d->H <= 0 || d->E < 0 || d->L < 0 || d->temperature == 0.f)
        return ASRK_EINVAL;
    if (d->nlayer < 0 || d->nlayer > ASRK_SPELLER_MAX_LAYERS || (d->nlayer > 1 && d->cell != 0)) return ASRK_ESHAPE;
    if (d->att_mode != 0 && d->att_mode != 1) return ASRK_EINVAL;
    // several heads: dot-product attention only (the location-aware form convolves across the heads' alignments)
    if (d->nhead < 0 || (d->nhead > 1 && d->att_mode != 1) || (d->att_mode == 1 && d->A > 64 * DOT_NA)) return ASRK_ESHAPE;
    return ASRK_OK;
}

inline int n_layers(const asrk_speller_t &d) { return d.nlayer > 1 ? d.nlayer : 1; }
inline int n_heads(const asrk_speller_t &d) { return d.nhead > 1 ? d.nhead : 1; }
// layer l's hidden / cell tape ([L+1,B,H]) and gate tape ([L,B,4H]): layer 0 keeps the original fields
inline float *h_of(const asrk_speller_t &d, int l) { return l == 0 ? d.h : d.hu[l - 1]; }
inline float *c_of(const asrk_speller_t &d, int l) { return l == 0 ? d.c : d.cu[l - 1]; }
inline float *g_of(const asrk_speller_t &d, int l) { return l == 0 ? d.gates : d.gu[l - 1]; }
inline bool upper_ok(const asrk_speller_t &d, bool fwd) {
    for (int l = 1; l < n_layers(d); ++l)
        if (!d.Wu_ih[l - 1] || !d.Wu_hh[l - 1] || !d.hu[l - 1] || !d.cu[l - 1] || !d.gu[l - 1] ||
            (fwd && (!d.bu_ih[l - 1] || !d.bu_hh[l - 1])))
            return false;
    return true;
}

template <typename T>
int set_lds(T kernel, size_t bytes) {
    ASRK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return ASRK_OK;
}

}  // namespace

extern "C" void asrk_prof_launches_(int id, int64_t n);

extern "C" int asrk_transpose_ld_f32(const float *in, int64_t ldi, float *out, int64_t ldo, int rows,
                                     int cols, void *stream) {
    if (rows < 0 || cols < 0 || ldi < cols || ldo < rows) return ASRK_EINVAL;
    if (rows == 0 || cols == 0) return ASRK_OK;
    if (!in || !out) return ASRK_EINVAL;
    hipLaunchKernelGGL(transpose_ld_kernel, dim3(asrk_div_up(cols, 32), asrk_div_up(rows, 32)), dim3(256), 0,
                       (hipStream_t)stream, in, (long)ldi, out, (long)ldo, rows, cols);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_speller_dvalue_f32(const float *attn, int64_t attn_ld, int64_t attn_step, const float *dxh,
                                       int64_t step_ld, int64_t row_ld, float *dvalue, int B, int L, int Te,
                                       int Dv, void *stream) {
    if (B < 0 || L <= 0 || Te <= 0 || Dv <= 0) return ASRK_EINVAL;
    if (B == 0) return ASRK_OK;
    if (!attn || !dxh || !dvalue) return ASRK_EINVAL;
    const size_t lds = (size_t)DV_LC * Te * sizeof(float);
    if (lds > LDS_BUDGET) return ASRK_ESHAPE;
    int rc = set_lds(dvalue_kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(dvalue_kernel, dim3(B, asrk_div_up(Dv, 256)), dim3(256), lds, (hipStream_t)stream, attn,
                       (long)attn_ld, (long)attn_step, dxh, (long)step_ld, (long)row_ld, dvalue, L, Te, Dv);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

// debug: device buffer of `slots` x 16 uint64 receiving the kernels' phase stamps (slot = step * 16 + kernel; NULL = off)
extern "C" void asrk_speller_set_debug_(void *buf, int slots) {
    g_sp_dbg = reinterpret_cast<unsigned long long *>(buf);
    g_sp_dbg_slots = slots;
}

extern "C" int asrk_speller_plan(const asrk_speller_t *d, int *tc_fwd, int *tc_bwd) {
    int rc = check_dims(d);
    if (rc) return rc;
    Plan pl;
    rc = make_plan(*d, pl);
    if (rc) return rc;
    if (tc_fwd) *tc_fwd = pl.tc_f;
    if (tc_bwd) *tc_bwd = pl.tc_b;
    return ASRK_OK;
}

// One attention + decoder-cell step (also the body of the training loop).  `step` selects the tape
// slots; prev = previous attention row pointer + row stride.
static int speller_step_fwd(const asrk_speller_t &d, const Plan &pl, int t, const float *prev, long prev_ld,
                            const float *pre, const float *emb, hipStream_t s) {
    const int B = d.B, H = d.H, A = d.A, Te = d.Te, Dv = d.Dv, K = d.K;
    const long In = (long)d.E + Dv;
    const int NH = n_heads(d), BN = B * NH;      // attention rows r = b * NH + n
    float *q_t = d.q + (long)t * BN * A;
    const float *h_t = d.h + (long)t * B * H;
    if (d.Wq) {   // F1 (Wq == NULL, asrk_speller_step_f32 only: q slot t already holds the query)
        SkArgs a{};
        const int NL = n_layers(d);
        a.nseg = NL;                              // the query reads the layer-concatenated state (src/asr.py:207-212)
        for (int l = 0; l < NL; ++l)
            a.seg[l] = SkSeg{h_of(d, l) + (long)t * B * H, d.Wq + (long)l * H, (long)H, (long)NL * H, H};
        a.M = B; a.R = NH * A; a.H = H;
        a.out = q_t; a.ldo = (long)NH * A; a.bias = d.bq;
        a.stamps = sp_slot(t, 0);
        int rc = launch_skinny<EPI_TANH_BIAS>(a, s);
        if (rc) return rc;
    }
    if (d.att_mode == 1) {   // F2a, dot-product form
        DotArgs a{d.key, q_t, nullptr, nullptr, d.lens, d.e_scratch, nullptr, nullptr, 0, Te, A, NH, 0,
                  1.f / d.temperature};
        hipLaunchKernelGGL(dot_energy_kernel, dim3(BN, asrk_div_up(Te, 8)), dim3(512), 0, s, a);
    } else {   // F2a
        AttArgs a{d.key, q_t, prev, d.Wc, d.Wp, d.we, d.be, d.lens, d.conv + (long)t * B * Te * K, d.e_scratch,
                  prev_ld, Te, A, K, d.ks, pl.tpb_f, pl.KP, 1.f / d.temperature, d.shared_kv ? 0 : 1, d.row_mem};
        a.stamps = sp_slot(t, 1);
        launch_attend_energy(a, pl, B, s);
    }
    float *attn_t = d.attn + (long)t * d.attn_step;
    float *ctx_t = d.ctx + (long)t * B * Dv;
    float *ctxh_t = NH > 1 ? d.ctxh + (long)t * BN * Dv : ctx_t;      // per-head contexts (merged below)
    {   // F2b
        CtxArgs a{d.e_scratch, d.value, attn_t, ctxh_t, d.attn_ld, (long)Dv, Te, Dv, d.shared_kv ? 0 : 1, d.row_mem};
        a.stamps = sp_slot(t, 2);
        const bool vec = al16(d.value) && Dv % 4 == 0;
        const dim3 grid(BN, asrk_div_up(Dv, 256));
        const int RG = d.row_group;
        const size_t lds_g = (size_t)RG * ((Te + 3) & ~3) * sizeof(float);
        if (RG > 1 && RG <= 8 * CG_MAXR && NH == 1 && vec && B % RG == 0 && al16(ctxh_t) && Dv % 4 == 0 &&
            lds_g <= 64 * 1024 && !d.shared_kv && (B / RG) * asrk_div_up(Dv, 256) >= 64)   // a few groups: the per-row grid fills the chip better
            hipLaunchKernelGGL(softmax_context_group_kernel, dim3(B / RG, asrk_div_up(Dv, 256)), dim3(512), lds_g, s, a, RG);
        else if (vec) hipLaunchKernelGGL(softmax_context_kernel<true>, grid, dim3(512), pl.lds_ctx, s, a);
        else hipLaunchKernelGGL(softmax_context_kernel<false>, grid, dim3(512), pl.lds_ctx, s, a);
    }
    if (NH > 1) {   // merge_head (src/asr.py:308-311): ctx [B,Dv] = [ctx_head_0 | ... ] Wm^T + bm
        SkArgs a{};
        a.nseg = 1;
        a.seg[0] = SkSeg{ctxh_t, d.Wm, (long)NH * Dv, (long)NH * Dv, NH * Dv};
        a.M = B; a.R = Dv; a.H = H;
        a.out = ctx_t; a.ldo = Dv; a.bias = d.bm;
        int rc = launch_skinny<EPI_STORE>(a, s);
        if (rc) return rc;
    }
    if (!pre && !emb) return ASRK_OK;   // attention only (asrk_speller_step_f32 with emb == NULL): the caller runs the cell
    {   // F3
        SkArgs a{};
        int n = 0;
        if (emb) a.seg[n++] = SkSeg{emb, d.W_ih, (long)d.E, In, d.E};
        a.seg[n++] = SkSeg{ctx_t, d.W_ih + d.E, (long)Dv, In, Dv};
        a.seg[n++] = SkSeg{h_t, d.W_hh, (long)H, (long)H, H};
        a.nseg = n;
        a.M = B; a.R = 4 * H; a.H = H;
        a.pre = pre;
        a.b0 = emb ? d.b_ih : nullptr;
        a.b1 = emb ? d.b_hh : nullptr;
        a.gru = d.cell;
        a.c_prev = d.cell ? h_t : d.c + (long)t * B * H;
        a.c_new = d.cell ? nullptr : d.c + (long)(t + 1) * B * H;
        a.h_new = d.h + (long)(t + 1) * B * H;
        a.gates = d.gates ? d.gates + (long)t * B * 4 * H : nullptr;
        a.h_bm = (d.states && n_layers(d) == 1) ? d.states + (long)t * H : nullptr;
        a.h_bm_ld = (long)d.L * H;
        a.stamps = sp_slot(t, 3);
        int rc = launch_skinny<EPI_LSTM_FWD>(a, s);
        if (rc) return rc;
    }
    // stacked decoder (nn.LSTM(num_layers > 1), src/asr.py:175-176): layer l's step reads the step's new state of the
    // layer below and its own previous state; the top layer's output is the decoder output
    for (int l = 1; l < n_layers(d); ++l) {
        SkArgs a{};
        a.nseg = 2;
        a.seg[0] = SkSeg{h_of(d, l - 1) + (long)(t + 1) * B * H, d.Wu_ih[l - 1], (long)H, (long)H, H};
        a.seg[1] = SkSeg{h_of(d, l) + (long)t * B * H, d.Wu_hh[l - 1], (long)H, (long)H, H};
        a.M = B; a.R = 4 * H; a.H = H;
        a.b0 = d.bu_ih[l - 1];
        a.b1 = d.bu_hh[l - 1];
        a.c_prev = c_of(d, l) + (long)t * B * H;
        a.c_new = c_of(d, l) + (long)(t + 1) * B * H;
        a.h_new = h_of(d, l) + (long)(t + 1) * B * H;
        a.gates = g_of(d, l) + (long)t * B * 4 * H;
        a.h_bm = (d.states && l == n_layers(d) - 1) ? d.states + (long)t * H : nullptr;
        a.h_bm_ld = (long)d.L * H;
        int rc = launch_skinny<EPI_LSTM_FWD>(a, s);
        if (rc) return rc;
    }
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

static int prep_attrs(const Plan &pl) {
    int rc = set_lds(attend_energy_kernel, pl.lds_f);
    if (rc) return rc;
    if (pl.ae2_na) {
        rc = pl.ae2_na == 2 ? (pl.ae2_km == 12 ? set_lds(attend_energy_kernel2<2, 12>, pl.lds_f2) : set_lds(attend_energy_kernel2<2, 16>, pl.lds_f2))
                            : (pl.ae2_km == 12 ? set_lds(attend_energy_kernel2<5, 12>, pl.lds_f2) : set_lds(attend_energy_kernel2<5, 16>, pl.lds_f2));
        if (rc) return rc;
    }
    rc = set_lds(softmax_context_kernel<true>, pl.lds_ctx);
    if (rc) return rc;
    rc = set_lds(softmax_context_kernel<false>, pl.lds_ctx);
    if (rc) return rc;
    rc = set_lds(energy_bwd_kernel2, pl.lds_b);
    if (rc) return rc;
    if (pl.eb3_na) {
        rc = pl.eb3_na == 2 ? (pl.eb3_km == 12 ? set_lds(energy_bwd_kernel3<2, 12>, pl.lds_b3) : set_lds(energy_bwd_kernel3<2, 16>, pl.lds_b3))
                            : (pl.eb3_km == 12 ? set_lds(energy_bwd_kernel3<5, 12>, pl.lds_b3) : set_lds(energy_bwd_kernel3<5, 16>, pl.lds_b3));
        if (rc) return rc;
    }
    return set_lds(conv_bwd_kernel, pl.lds_cb);
}

extern "C" int asrk_speller_fwd_f32(const asrk_speller_t *d, void *stream) {
    int rc = check_dims(d);
    if (rc) return rc;
    if (d->B == 0 || d->L == 0) return ASRK_OK;
    const bool loc = d->att_mode == 0;
    if (!d->key || !d->value || !d->lens || !d->Wq || !d->W_ih ||
        !d->W_hh || !d->eproj || !d->q || !d->attn || !d->ctx || !d->h || (!d->c && !d->cell) ||
        !d->e_scratch || (d->cell != 0 && d->cell != 1) || !upper_ok(*d, true))
        return ASRK_EINVAL;
    if (loc && (!d->Wc || !d->Wp || !d->we || !d->be || !d->conv || !d->prev0)) return ASRK_EINVAL;
    if (n_heads(*d) > 1 && (!d->Wm || !d->ctxh)) return ASRK_EINVAL;
    if (!loc && (d->shared_kv || d->row_mem)) return ASRK_EINVAL;
    Plan pl;
    rc = make_plan(*d, pl);
    if (rc) return rc;
    rc = prep_attrs(pl);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_SPELLER, s);
    for (int t = 0; t < d->L; ++t) {
        const float *prev = t == 0 ? d->prev0 : d->attn + (long)(t - 1) * d->attn_step;
        const long prev_ld = t == 0 ? d->Te : d->attn_ld;
        rc = speller_step_fwd(*d, pl, t, prev, prev_ld, d->eproj + (long)t * d->B * 4 * d->H, nullptr, s);
        if (rc) return rc;
    }
    asrk_prof_end_(PROF_SPELLER, s);
    asrk_prof_launches_(PROF_SPELLER, (3L + n_layers(*d) + (n_heads(*d) > 1)) * d->L - 1);
    return ASRK_OK;
}

extern "C" int asrk_speller_step_f32(const asrk_speller_t *d, int slot, const float *prev_att,
                                     int64_t prev_ld, const float *emb, void *stream) {
    int rc = check_dims(d);
    if (rc) return rc;
    if (d->B == 0) return ASRK_OK;
    if (slot < 0 || slot >= d->L) return ASRK_EINVAL;
    if (d->nlayer > 1 || d->att_mode != 0 || d->nhead > 1) return ASRK_ESHAPE;   // decode paths: one layer, 'loc', one head
    // emb == NULL: the attention half of the step only (query, energies, alignment, context from h slot `slot`); the
    // caller runs the decoder cell itself (many rows: bf16x6 panel GEMMs instead of 64-row weight-streaming tiles)
    if (!d->key || !d->value || !d->lens || !d->Wc || !d->Wp || !d->we || !d->be || !d->q || !d->conv ||
        !d->attn || !d->ctx || !d->h || !d->e_scratch || !prev_att || (d->cell != 0 && d->cell != 1) || d->row_group < 0)
        return ASRK_EINVAL;
    if (!d->Wq && emb) return ASRK_EINVAL;          // a given query comes with the attention-only form
    if (emb && (!d->W_ih || !d->W_hh || !d->b_ih || !d->b_hh || (!d->c && !d->cell))) return ASRK_EINVAL;
    Plan pl;
    rc = make_plan(*d, pl);
    if (rc) return rc;
    rc = prep_attrs(pl);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_SPELLER, s);
    rc = speller_step_fwd(*d, pl, slot, prev_att, (long)prev_ld, nullptr, emb, s);
    asrk_prof_end_(PROF_SPELLER, s);
    asrk_prof_launches_(PROF_SPELLER, 3);
    return rc;
}

extern "C" int asrk_joint_score_f32(const float *att_logp, const int64_t *cand, const float *psi,
                                    const float *prev_ctc, const float *lm_logp, float *out, int n, int V,
                                    int C, double ctc_weight, double lm_weight, double logzero, void *stream) {
    if (n < 0 || V <= 0 || C < 0) return ASRK_EINVAL;
    if (n == 0) return ASRK_OK;
    if (!att_logp || !out || out == att_logp) return ASRK_EINVAL;
    if (cand && (!psi || !prev_ctc)) return ASRK_EINVAL;
    // the reference multiplies f32 tensors by Python (double) scalars: each scalar is rounded to f32 once
    hipLaunchKernelGGL(joint_score_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, att_logp, cand, psi,
                       prev_ctc, lm_logp, out, V, C, (float)(1.0 - ctc_weight), (float)ctc_weight,
                       (float)lm_weight, (float)logzero);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_lstm_cell_fused_f32(const float *x, int64_t ldx, int In, const float *h, const float *c,
                                        const float *W_ih, const float *W_hh, const float *b_ih,
                                        const float *b_hh, float *h_out, float *c_out, int B, int H,
                                        void *stream) {
    if (B < 0 || H <= 0 || In <= 0 || ldx < In) return ASRK_EINVAL;
    if (B == 0) return ASRK_OK;
    if (!x || !h || !c || !W_ih || !W_hh || !h_out || !c_out) return ASRK_EINVAL;
    SkArgs a{};
    a.nseg = 2;
    a.seg[0] = SkSeg{x, W_ih, (long)ldx, (long)In, In};
    a.seg[1] = SkSeg{h, W_hh, (long)H, (long)H, H};
    a.M = B; a.R = 4 * H; a.H = H;
    a.b0 = b_ih; a.b1 = b_hh;
    a.c_prev = c; a.c_new = c_out; a.h_new = h_out;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_CELL, s);
    const int rc = launch_skinny<EPI_LSTM_FWD>(a, s);
    asrk_prof_end_(PROF_CELL, s);
    return rc;
}

extern "C" int asrk_speller_bwd_f32(const asrk_speller_t *d, const asrk_speller_bwd_t *g, void *stream) {
    int rc = check_dims(d);
    if (rc) return rc;
    if (!g) return ASRK_EINVAL;
    if (d->B == 0 || d->L == 0) return ASRK_OK;
    const bool loc = d->att_mode == 0;
    if (!d->key || !d->value || !d->lens || !d->Wq || !d->q ||
        !d->attn || !d->gates || !d->h || (!d->c && !d->cell) || (d->cell != 0 && d->cell != 1) ||
        !g->dstates || !g->WT || !g->WqT ||
        !g->dkey || !g->dxh || !g->dq_pre || !g->dattn || !g->dq_part || !g->dc || !upper_ok(*d, false))
        return ASRK_EINVAL;
    if (loc && (!d->Wc || !d->Wp || !d->we || !d->conv || !d->prev0 || !g->dprev || !g->dconv || !g->dwe_part ||
                !g->dWp_part || !g->dbe_part || !g->dWc_part))
        return ASRK_EINVAL;
    if (n_heads(*d) > 1 && (!g->WmT || !g->dctxh)) return ASRK_EINVAL;
    for (int l = 1; l < n_layers(*d); ++l)
        if (!g->WuT[l - 1] || !g->dxu[l - 1] || !g->dcu[l - 1]) return ASRK_EINVAL;
    if (d->shared_kv || d->row_mem) return ASRK_EINVAL;   // gradients are per batch row
    Plan pl;
    rc = make_plan(*d, pl);
    if (rc) return rc;
    if (g->tc != pl.tc_b) return ASRK_EWORKSPACE;
    rc = prep_attrs(pl);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int B = d->B, H = d->H, A = d->A, Te = d->Te, Dv = d->Dv, K = d->K, L = d->L;
    const long XH = (long)Dv + H;
    asrk_prof_begin_(PROF_SPELLER, s);
    const int NL = n_layers(*d), top = NL - 1, NH = n_heads(*d), BN = B * NH;
    // Cell backward of layer l at decode step `st` (EPI_LSTM_BWD epilogue): dh = [dq_pre_{st+1} Wq_l] + add0 + add1 ->
    // dG^l_st.  add0 = the hidden-to-hidden path from dG^l_{st+1} (none at the last step); add1 = the output gradient:
    // dstates for the top layer, the input path of the layer above (dxu^{l+1}_st) below it.
    auto cell_bwd = [&](int l, int st, const float *dq_pre_next, const float *add0, long ld0, int stamp_t) {
        SkArgs a{};
        a.nseg = dq_pre_next ? 1 : 0;
        if (dq_pre_next) a.seg[0] = SkSeg{dq_pre_next, g->WqT + (long)l * H * NH * A, (long)NH * A, (long)NH * A, NH * A};
        a.M = B; a.R = H; a.H = H;
        a.add0 = add0; a.ld0 = ld0;
        if (l == top) { a.add1 = g->dstates + (long)st * H; a.ld1 = (long)L * H; }
        else { a.add1 = g->dxu[l] + (long)st * B * 2 * H; a.ld1 = 2L * H; }
        a.dG = g_of(*d, l) + (long)st * B * 4 * H;
        a.gru = d->cell;
        a.bc_prev = (d->cell ? d->h : c_of(*d, l)) + (long)st * B * H;
        a.bc_new = d->cell ? a.bc_prev : c_of(*d, l) + (long)(st + 1) * B * H;
        a.dc = l == 0 ? g->dc : g->dcu[l - 1];
        a.dc_valid = dq_pre_next ? 1 : 0;
        if (stamp_t >= 0 && l == 0) a.stamps = sp_slot(stamp_t, 8);
        return launch_skinny<EPI_LSTM_BWD>(a, s);
    };
    // dxu^l_st = dG^l_st [W_ih_l | W_hh_l]: columns [0,H) go to the layer below (same step), [H,2H) to layer l's
    // previous step
    auto upper_dx = [&](int l, int st) {
        SkArgs a{};
        a.nseg = 1;
        a.seg[0] = SkSeg{g_of(*d, l) + (long)st * B * 4 * H, g->WuT[l - 1], 4L * H, 4L * H, 4 * H};
        a.M = B; a.R = 2 * H; a.H = H;
        a.out = g->dxu[l - 1] + (long)st * B * 2 * H; a.ldo = 2L * H;
        return launch_skinny<EPI_STORE>(a, s);
    };
    for (int l = top; l >= 0; --l) {   // the last step: no later step feeds back (no query, no hidden path, no dc yet)
        rc = cell_bwd(l, L - 1, nullptr, nullptr, 0, -1);
        if (rc) return rc;
        if (l >= 1) {
            rc = upper_dx(l, L - 1);
            if (rc) return rc;
        }
    }
    for (int t = L - 1; t >= 0; --t) {
        float *dG_t = d->gates + (long)t * B * 4 * H;
        float *dxh_t = g->dxh + (long)t * B * XH;
        {   // B2
            SkArgs a{};
            a.nseg = 1;
            a.seg[0] = SkSeg{dG_t, g->WT, 4L * H, 4L * H, 4 * H};
            a.M = B; a.R = (int)XH; a.H = H;
            a.out = dxh_t; a.ldo = XH;
            a.stamps = sp_slot(t, 4);
            rc = launch_skinny<EPI_STORE>(a, s);
            if (rc) return rc;
        }
        const float *attn_t = d->attn + (long)t * d->attn_step;
        const float *dctx_rows = dxh_t;          // gradient of the per-row contexts: [BN, Dv] with this row stride
        long dctx_ld = XH;
        if (NH > 1) {   // merge_head backward: d ctx_heads [B, NH*Dv] = dctx_t Wm
            SkArgs a{};
            a.nseg = 1;
            a.seg[0] = SkSeg{dxh_t, g->WmT, XH, (long)Dv, Dv};
            a.M = B; a.R = NH * Dv; a.H = H;
            float *dctxh_t = g->dctxh + (long)t * BN * Dv;
            a.out = dctxh_t; a.ldo = (long)NH * Dv;
            rc = launch_skinny<EPI_STORE>(a, s);
            if (rc) return rc;
            dctx_rows = dctxh_t;
            dctx_ld = Dv;
        }
        {   // B3
            DattnArgs a{dctx_rows, d->value, g->dattn_seq ? g->dattn_seq + (long)t * d->attn_step : nullptr,
                        (loc && t + 1 < L) ? g->dprev : nullptr, d->lens, g->dattn, dctx_ld, d->attn_ld, (long)Te, Te, Dv};
            a.stamps = sp_slot(t, 5);
            a.lens_div = NH - 1;
            const bool vec = al16(d->value) && al16(dctx_rows) && Dv % 4 == 0 && dctx_ld % 4 == 0;
            const dim3 grid(BN, asrk_div_up(Te, 8));
            if (vec) hipLaunchKernelGGL(dattn_kernel<true>, grid, dim3(512), 0, s, a);
            else hipLaunchKernelGGL(dattn_kernel<false>, grid, dim3(512), 0, s, a);
        }
        const float *q_t = d->q + (long)t * BN * A;
        float *dq_pre_t = g->dq_pre + (long)t * BN * A;
        if (!loc) {   // B4 + B5, dot-product form: de -> dkey, dq partials; dq_pre = (sum of the partials) (1 - q^2)
            DotArgs a{d->key, q_t, attn_t, g->dattn, d->lens, nullptr, g->dkey, g->dq_part, d->attn_ld, Te, A, NH,
                      pl.tpb_b, 1.f / d->temperature};
            hipLaunchKernelGGL(dot_energy_bwd_kernel, dim3(BN, pl.tc_b), dim3(512), 0, s, a);
            CbArgs c{nullptr, nullptr, nullptr, g->dq_part, q_t, nullptr, nullptr, dq_pre_t, 0, Te, 0, 0, pl.tc_b, A,
                     0, 0};
            c.dq_only = 1;
            hipLaunchKernelGGL(conv_bwd_kernel, dim3(BN, 1), dim3(256), pl.lds_cb, s, c);
        } else {
        const float *conv_t = d->conv + (long)t * B * Te * K;
        {   // B4
            EbArgs a{d->key, q_t, conv_t, d->Wp, d->we, attn_t, g->dattn, d->lens, g->dkey, g->dconv,
                     g->dq_part, g->dwe_part, g->dWp_part, g->dbe_part, d->attn_ld, Te, A, K, pl.tpb_b,
                     pl.KP, 1.f / d->temperature};
            a.stamps = sp_slot(t, 6);
            const dim3 grid(B, pl.tc_b);
            if (!pl.eb3_na) hipLaunchKernelGGL(energy_bwd_kernel2, grid, dim3(512), pl.lds_b, s, a);
            else if (pl.eb3_na == 2 && pl.eb3_km == 12) hipLaunchKernelGGL((energy_bwd_kernel3<2, 12>), grid, dim3(512), pl.lds_b3, s, a);
            else if (pl.eb3_na == 2) hipLaunchKernelGGL((energy_bwd_kernel3<2, 16>), grid, dim3(512), pl.lds_b3, s, a);
            else if (pl.eb3_km == 12) hipLaunchKernelGGL((energy_bwd_kernel3<5, 12>), grid, dim3(512), pl.lds_b3, s, a);
            else hipLaunchKernelGGL((energy_bwd_kernel3<5, 16>), grid, dim3(512), pl.lds_b3, s, a);
        }
        {   // B5
            const float *prev = t == 0 ? d->prev0 : d->attn + (long)(t - 1) * d->attn_step;
            CbArgs a{g->dconv, prev, d->Wc, g->dq_part, q_t, g->dprev, g->dWc_part, dq_pre_t,
                     t == 0 ? (long)Te : d->attn_ld, Te, K, d->ks, pl.tc_b, A, pl.nT, t > 0 ? 1 : 0};
            a.stamps = sp_slot(t, 7);
            hipLaunchKernelGGL(conv_bwd_kernel, dim3(B, pl.nT + K + 1), dim3(256), pl.lds_cb, s, a);
        }
        }
        if (t > 0) {   // B6: dh_{t-1} and the cell backward of step t-1, top layer first
            for (int l = top; l >= 0; --l) {
                const float *hh = l == 0 ? dxh_t + Dv : g->dxu[l - 1] + (long)t * B * 2 * H + H;
                rc = cell_bwd(l, t - 1, dq_pre_t, hh, l == 0 ? XH : 2L * H, t);
                if (rc) return rc;
                if (l >= 1) {
                    rc = upper_dx(l, t - 1);
                    if (rc) return rc;
                }
            }
        }
        ASRK_LAUNCH_CHECK();
    }
    asrk_prof_end_(PROF_SPELLER, s);
    asrk_prof_launches_(PROF_SPELLER, (5L + 2 * (NL - 1) + (NH > 1)) * L - 1);
    return ASRK_OK;
}
