// f32 GEMM on the bf16 matrix cores by exact operand splitting ("bf16x6"), gfx950.
//
// Why: v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate (157 TF vs 2.5 PF).  An f32 number is
// EXACTLY the sum of three bf16 numbers (24 = 8 + 8 + 8 significand bits, same exponent range):
//     a = a0 + a1 + a2,   a0 = bf16(a), a1 = bf16(a - a0), a2 = a - a0 - a1  (exact; round-to-nearest pieces)
// so  a*b = sum over the nine a_i*b_j.  The three products with i + j >= 3 are below 2^-24 |a b| each -
// the size of ONE f32 rounding - and are dropped; the other six are bf16 MFMAs whose products are exact in
// the f32 accumulator.  The result carries the same kind of error as an f32 FMA chain (a few 2^-24 per
// product, f32 accumulation) at 6/16 of its cost: 417 TF/s-equivalent peak instead of 157.
// tests/test_kernels_gpu.py::test_gemm_split_* check it against float64 beside the exact-f32 kernel.
//
// Serves the same contractions as gemm.hip (the reference's ATen GEMMs: nn.LSTM input projection
// src/module.py:131, heads src/asr.py:96,220, and their autograd GEMMs) when they are big enough to
// amortise the split pass; asrk_gemm_f32 routes here (asrk_gemm_set_split / ASRK_GEMM_SPLIT).
//
// Two steps per operand-pair:
//  1. split_panel_kernel: f32 matrix (either storage order) -> "split panel": for every 64-row block and
//     every group of 8 k's, three 1-KiB pieces (one per bf16 plane), each [64 rows][8 bf16].  A piece is
//     exactly one global_load_lds_dwordx4 wave instruction (64 lanes x 16 B, coalesced, lane-linear in
//     LDS) and exactly the image ds_read_b128 wants: lane l reads row l&31 of a piece, so every 16-lane
//     service group hits 16 distinct 16-B slots (conflict-free without padding or swizzle).  Transposed
//     operands (dW = dY^T X) are handled here, so the GEMM kernel has ONE operand form.
//  2. gemm_bf16x6_kernel: 128x128 tile, 4 waves (2x2), wave tile 64x64 = 2x2 v_mfma_f32_32x32x16_bf16
//     accumulators; per 16-k step a wave reads 6 A + 6 B fragments (3 planes x 2 row tiles each) and
//     issues 24 MFMAs - twice the MFMA work per LDS byte of a plain bf16 GEMM.  Global -> LDS by LDS-DMA
//     (no staging registers), NST-deep ring, one raw s_barrier per k-tile, counted vmcnt.
#include "common.h"
#include "knobs.h"
#include <algorithm>

extern "C" int asrk_cu_count_(void);

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int PIECE = 1024;                 // bytes: 64 rows x 8 16-bit values

struct SplitGemmArgs {
    const unsigned char *Ap, *Bp;           // split panels
    float *C;
    const float *bias, *bias2;
    int M, N, ldc;
    int KC;                                 // 8-k groups per row (K padded to the k-tile)
    size_t rb_stride_a, rb_stride_b;        // bytes between consecutive 64-row blocks of the A / B panel
    int nk;                                 // k-tiles
    int tiles_m, tiles_n;
    float alpha, beta;
};

// ---------------------------------------------------------------------------------- split pass
// dst piece (rb, c, p) at rb*rb_stride + (c*3 + p) * 1024; element (r, e) of it at r*16 + e*2.
// rb_stride = KC*3 KiB + a pad that is odd in units of 256 B: at a given k the row blocks a chip works on
// concurrently then start in different L2 / memory channels instead of all in the same one.
// TRANS = false: src[row*ld + k];  TRANS = true: src[k*ld + row].  Rows >= R and k >= K are zero.
__device__ __forceinline__ void split8(const float (&v)[8], u32x4 (&w)[3]) {
    unsigned h[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        // round-to-nearest pieces (v_cvt_pk_bf16_f32): |a1| <= 2^-9 |a|, |a2| <= 2^-18 |a|, residual signs
        // are unbiased; a - a0 and r1 - a1 are exact in f32 and r2 has <= 6 significant bits -> a2 exact
        const float a = v[e];
        const __bf16 b0 = (__bf16)a;
        const float r1 = a - (float)b0;
        const __bf16 b1 = (__bf16)r1;
        const float r2 = r1 - (float)b1;
        const __bf16 b2 = (__bf16)r2;
        h[0][e] = __builtin_bit_cast(unsigned short, b0);
        h[1][e] = __builtin_bit_cast(unsigned short, b1);
        h[2][e] = __builtin_bit_cast(unsigned short, b2);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) w[p][q] = h[p][2 * q] | (h[p][2 * q + 1] << 16);
}

// Row-major sources ([rows][K]): one wave = one (row block, group of 4 chunk columns) = 64 rows x 32 k, items walked k
// fastest so that the waves of a workgroup and the workgroups running side by side read one contiguous run of every
// source row.  VEC: 16-B loads that touch whole cache lines (4 passes of 16 rows; lane = (row lane >> 2, chunk column
// lane & 3) reads 2 x float4 - the 4 lanes of a row cover one 128-B line); otherwise element loads with bounds checks
// (edges, unaligned views).
template <bool VEC>
__global__ __launch_bounds__(256) void split_panel_kernel(const float *__restrict__ src, int ld, int R, int K,
                                                          unsigned char *__restrict__ dst, int KC, int RB,
                                                          size_t rb_stride) {
    constexpr int CHUNK = 3 * PIECE;             // one 8-k group of one row block: three planes
    const int lane = threadIdx.x & 63;
    const int KG = KC >> 2;                                                   // groups of 4 chunk columns
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= (int64_t)RB * KG) return;
    const int rb = (int)(item / KG), c0 = (int)(item - (int64_t)rb * KG) * 4;
    unsigned char *drb = dst + (size_t)rb * rb_stride;
    const int c = c0 + (lane & 3), k0 = c * 8;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int rl = it * 16 + (lane >> 2), row = rb * 64 + rl;
        float v[8];
        const float *s = src + (size_t)row * ld + k0;
        if (VEC && row < R && k0 + 8 <= K) {
            const f32x4 lo = *reinterpret_cast<const f32x4 *>(s), hi = *reinterpret_cast<const f32x4 *>(s + 4);
            v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
            v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (row < R && k0 + e < K) ? s[e] : 0.f;
        }
        u32x4 w[3];
        split8(v, w);
        unsigned char *d = drb + (size_t)c * CHUNK + rl * 16;
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4 *>(d + p * PIECE) = w[p];
    }
}

// [K][rows] sources WITHOUT a transposition: lane = ROW of a 64-row block, wave = (row block, NCG chunk columns = 8 NCG
// k).  A load instruction reads 64 consecutive floats of ONE source row (256 contiguous bytes = two cache lines; the
// four waves of a workgroup take four neighbouring row blocks, i.e. 1 KiB per k row and workgroup, and the next
// workgroup continues the same k rows), so a lane ends up holding exactly the 8 k's of its row that one 16-byte slot of a
// piece stores: no LDS strip, no cross-lane traffic, every store instruction is a whole 1-KiB piece, and with no LDS and
// ~60 registers the CU holds 8 waves per SIMD of independent 8 NCG-load streams.  (Rounds 3-4 transposed through a
// wave-private LDS strip, 16-byte loads along the rows: the same 5.4-6.6 TB/s in the step, git log.)
template <int NCG>
__global__ __launch_bounds__(256) void split_panel_t_kernel(const float *__restrict__ src, int ld, int R, int K,
                                                            unsigned char *__restrict__ dst, int KC, int RB,
                                                            size_t rb_stride) {
    constexpr int CHUNK = 3 * PIECE;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;
    const int KGn = KC / NCG;
    if (item >= (int64_t)RB * KGn) return;
    const int g = (int)(item / RB), rb = (int)(item - (int64_t)g * RB);     // row block fastest: along the source rows
    const int c0 = g * NCG, k0 = c0 * 8, row = rb * 64 + lane;
    float v[NCG][8];
    const float *s = src + (size_t)k0 * ld + row;
    if (k0 + 8 * NCG <= K && rb * 64 + 64 <= R) {
#pragma unroll
        for (int cc = 0; cc < NCG; ++cc)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[cc][e] = s[(size_t)(cc * 8 + e) * ld];
    } else {
#pragma unroll
        for (int cc = 0; cc < NCG; ++cc)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                v[cc][e] = (row < R && k0 + cc * 8 + e < K) ? s[(size_t)(cc * 8 + e) * ld] : 0.f;
    }
    unsigned char *d = dst + (size_t)rb * rb_stride + (size_t)c0 * CHUNK + lane * 16;
#pragma unroll
    for (int cc = 0; cc < NCG; ++cc) {
        u32x4 w[3];
        split8(v[cc], w);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4 *>(d + (size_t)cc * CHUNK + p * PIECE) = w[p];
    }
}

// ---------------------------------------------------------------------------------- GEMM
// N <= 4 consecutive 1-KiB pieces (contiguous in global memory and in LDS) from a wave-uniform base: one address, one M0
// value, the piece in the instruction's immediate offset (which applies to both addresses; < 4096)
template <int N>
__device__ __forceinline__ void glds16_run(const unsigned char *g, unsigned voff, unsigned char *l) {
    static_assert(N >= 1 && N <= 4, "immediate offset range");
    const auto *gp = (const __attribute__((address_space(1))) unsigned char *)g + voff;
    auto *lp = (__attribute__((address_space(3))) void *)l;
    __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
    if constexpr (N > 1) __builtin_amdgcn_global_load_lds(gp, lp, 16, PIECE, 0);
    if constexpr (N > 2) __builtin_amdgcn_global_load_lds(gp, lp, 16, 2 * PIECE, 0);
    if constexpr (N > 3) __builtin_amdgcn_global_load_lds(gp, lp, 16, 3 * PIECE, 0);
}

// s_waitcnt through the builtin (the compiler's own wait-count tracking sees it; inline asm it would not):
// gfx9 encoding vmcnt = simm16[3:0] + [15:14], expcnt = [6:4], lgkmcnt = [11:8]
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    __builtin_amdgcn_s_waitcnt((N & 15) | 0x70 | 0xF00 | (((N >> 4) & 3) << 14));
}
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }

// NC = 8-k groups per k-tile (BK = 8*NC), NST = LDS ring depth.  Stage = 4 regions (A rows 0..63,
// A rows 64..127, B rows 0..63, B rows 64..127) of NC chunks; wave w fills region w.
// SPEC: 512 threads - waves 0..3 multiply, waves 4..7 only issue the LDS-DMA (an LDS-DMA instruction
// blocks its wave's issue for ~60-100 cycles; 12 per k-tile in the multiplying waves cost a third of the
// MFMA time, in a wave of their own they cost nothing).
// WM = 64-row blocks of A per tile: 2 -> 128x128 tile (4 multiplying + 4 DMA waves), 4 -> 256x128 tile
// (8 multiplying + 3 DMA waves, two regions each; 72 KiB per stage, 2 stages): a quarter fewer L2 -> LDS bytes
// per flop, for launches with enough tiles to fill the chip twice.
template <int NC, int NST, bool SPEC, int WM>
__global__ __launch_bounds__(SPEC ? (WM == 2 ? 512 : 704) : 256) void gemm_bf16x6_kernel(SplitGemmArgs p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    static_assert(WM == 2 || (WM == 4 && SPEC), "256-row tiles only with DMA waves");
    constexpr int NPL = 3, CHUNK = NPL * PIECE;   // three bf16 planes
    constexpr int NCW = 2 * WM;                  // multiplying waves (WM x 2, 64x64 each)
    constexpr int NRG = WM + 2;                  // regions per stage: WM row blocks of A, 2 of B
    constexpr int RPD = WM == 2 ? 1 : 2;         // regions per DMA wave
    constexpr int REGION = NC * CHUNK;
    constexpr int STAGE = NRG * REGION;
    constexpr int LPT = NC * NPL * RPD;          // LDS-DMA instructions per loading wave and k-tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // scalar: LDS-DMA bases stay in SGPRs
    // XCD-aware bijective tile remap (block b runs on XCD b % 8): each XCD walks a contiguous run of tiles,
    // n fastest, so the tiles resident on one XCD share A row blocks and neighbouring B blocks in its L2
    const int ntiles = p.tiles_m * p.tiles_n, bid = blockIdx.x;
    const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = bid & 7, loc = bid >> 3;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    // 8-tile-wide column bands, m fastest inside a band: 32 resident tiles per XCD = 4 x 8 patch
    constexpr int BAND = 8;
    const int band = tile / (BAND * p.tiles_m);
    const int band_w = min(BAND, p.tiles_n - band * BAND);
    const int in_band = tile - band * BAND * p.tiles_m;
    const int tm = in_band / band_w, tn = band * BAND + in_band % band_w;

    const bool loader = !SPEC || wave >= NCW, worker = !SPEC || wave < NCW;
    const int wr = worker ? wave >> 1 : 0, wc = wave & 1;
    // loader role: DMA wave d fills regions d*RPD .. d*RPD + RPD - 1 (without DMA waves: wave w fills region w)
    const int ltm = tm, ltn = tn;
    const int r0 = (SPEC ? wave - NCW : wave) * RPD;
    const unsigned char *gsrc[RPD];
#pragma unroll
    for (int q = 0; q < RPD; ++q) {
        const int r = r0 + q;
        gsrc[q] = (r < WM ? p.Ap + (size_t)(ltm * WM + r) * p.rb_stride_a
                          : p.Bp + (size_t)(ltn * 2 + r - WM) * p.rb_stride_b);          // wave-uniform
    }
    const unsigned voff = lane * 16;
    unsigned char *ldst = lds + r0 * REGION;
    static_assert((NC * NPL) % 4 == 0, "a region is issued as runs of four pieces");
    auto issue = [&](int kt, int stage) {
        unsigned char *l = ldst + stage * STAGE;
#pragma unroll
        for (int q = 0; q < RPD; ++q) {
            const unsigned char *g = gsrc[q] + (size_t)kt * REGION;
#pragma unroll
            for (int j = 0; j < NC * NPL; j += 4) glds16_run<4>(g + j * PIECE, voff, l + q * REGION + j * PIECE);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.nk;
    // fragment addresses: row tile i, plane pl, k-step ks: piece ((ks*2 + h)*3 + pl), row i*32 + (lane&31)
    const int frag_off = ((lane >> 5) * NPL) * PIECE + (lane & 31) * 16;
    const unsigned char *abase = lds + wr * REGION + frag_off;
    const unsigned char *bbase = lds + (WM + wc) * REGION + frag_off;
    constexpr int S = NC / 2;                    // 16-k steps per k-tile

    bf16x8 fa[2][2][NPL], fb[2][2][NPL];         // [buffer][row tile][plane]
    auto load_frags = [&](int buf, int stage, int ks) {      // prologue only
        const unsigned char *a_st = abase + stage * STAGE + ks * 2 * NPL * PIECE;
        const unsigned char *b_st = bbase + stage * STAGE + ks * 2 * NPL * PIECE;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                fa[buf][i][pl] = *reinterpret_cast<const bf16x8 *>(a_st + pl * PIECE + i * 512);
                fb[buf][i][pl] = *reinterpret_cast<const bf16x8 *>(b_st + pl * PIECE + i * 512);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    // One 16-k step: the products with i + j <= 2 (small terms first, four independent accumulators interleaved) on
    // fragment set `buf`; when `load`, the 4 * NPL fragment reads of step (lstage, lks) into the OTHER set are issued
    // one after each of the first 4 * NPL MFMAs, in the order the next step's products need them.  A clump of reads
    // ahead of the MFMAs idles the matrix pipe while it issues (a ds_read_b128 takes ~16 cycles of its wave's issue,
    // an MFMA 32: one read fits in an MFMA's shadow); the MFMAs after the last read cover its latency.
    auto step = [&](int buf, bool load, int lstage, int lks) {
        constexpr int NT = 6;
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
        constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
        // read order: the planes of the first product first.  r -> (operand, plane, row tile)
        constexpr int RPL[6] = {2, 0, 1, 1, 0, 2};
        const unsigned char *a_st = abase + lstage * STAGE + lks * 2 * NPL * PIECE;
        const unsigned char *b_st = bbase + lstage * STAGE + lks * 2 * NPL * PIECE;
        const int nb = buf ^ 1;
#pragma unroll
        for (int m = 0; m < 4 * NT; ++m) {
            const int t = m >> 2, i = (m >> 1) & 1, j = m & 1;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][i][PA[t]], fb[buf][j][PB[t]], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (load && m < 4 * NPL) {
                const int g = m >> 1, h = m & 1, pl = RPL[g];          // g even: A fragment, g odd: B fragment
                if ((g & 1) == 0) fa[nb][h][pl] = *reinterpret_cast<const bf16x8 *>(a_st + pl * PIECE + h * 512);
                else fb[nb][h][pl] = *reinterpret_cast<const bf16x8 *>(b_st + pl * PIECE + h * 512);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // prologue: fill the ring, wait for tile 0, first fragments
    if (loader) {
#pragma unroll
        for (int s = 0; s < NST; ++s)
            if (s < nk) issue(s, s);
        const int later = min(NST - 1, nk - 1);
        if (NST > 3 && later >= 3) wait_vm<(NST > 3 ? 3 * LPT : 0)>();
        else if (NST > 2 && later == 2) wait_vm<(NST > 2 ? 2 * LPT : 0)>();
        else if (later == 1) wait_vm<LPT>();
        else wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (SPEC && !worker) {
        // DMA wave: per k-tile  [tile kt+1 landed] -> barrier -> refill the stage the workers just left
        int stage = 0;
        for (int kt = 0; kt + 1 < nk; ++kt) {
            const int later = min(NST - 2, nk - 2 - kt);
            if (NST > 3 && later >= 2) wait_vm<(NST > 3 ? 2 * LPT : 0)>();
            else if (NST > 2 && later == 1) wait_vm<(NST > 2 ? LPT : 0)>();
            else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            if (kt + NST < nk) issue(kt + NST, stage);
            if (++stage == NST) stage = 0;
        }
        return;
    }
    load_frags(0, 0, 0);

    // One barrier per k-tile, placed before the tile's LAST 16-k step: by then every wave holds that step's
    // fragments in registers, so the stage can be refilled (tile kt + NST) and the next tile's first
    // fragments are fetched under the last step's MFMAs - no LDS latency is exposed at tile boundaries.
    int stage = 0;
    static_assert(S >= 2 && S % 2 == 0, "fragment double buffer assumes an even number of steps per tile");
    for (int kt = 0; kt + 1 < nk; ++kt) {
#pragma unroll
        for (int ks = 0; ks < S - 1; ++ks) {
            wait_lgkm0();                        // the fragments fetched under the previous step's MFMAs
            step(ks & 1, true, stage, ks + 1);
        }
        int nstage = stage + 1;
        if (nstage == NST) nstage = 0;
        // tile kt+1 has landed once at most NST-2 younger tiles are outstanding
        const int later = min(NST - 2, nk - 2 - kt);
        wait_lgkm0();                            // my reads of this stage are done
        if (!SPEC) {
            if (NST > 3 && later >= 2) wait_vm<(NST > 3 ? 2 * LPT : 0)>();
            else if (NST > 2 && later == 1) wait_vm<(NST > 2 ? LPT : 0)>();
            else wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();
        if (!SPEC && kt + NST < nk) issue(kt + NST, stage);
        step((S - 1) & 1, true, nstage, 0);
        stage = nstage;
    }
    {   // last tile: nothing left to fetch after its last step's fragments
#pragma unroll
        for (int ks = 0; ks < S - 1; ++ks) {
            wait_lgkm0();
            step(ks & 1, true, stage, ks + 1);
        }
        wait_lgkm0();
        step((S - 1) & 1, false, 0, 0);
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int row0 = tm * 64 * WM + wr * 64 + 4 * (lane >> 5), col0 = tn * 128 + wc * 64 + (lane & 31);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = col0 + j * 32;
        if (col >= p.N) continue;
        float bsum = 0.f;
        if (p.bias) bsum += p.bias[col];
        if (p.bias2) bsum += p.bias2[col];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                if (row >= p.M) continue;
                float *c = p.C + (size_t)row * p.ldc + col;
                float v = p.alpha * acc[i][j][r] + bsum;
                if (p.beta != 0.f) v += p.beta * *c;
                *c = v;
            }
    }
}

// ---------------------------------------------------------------------------------- 128 x 256 tile variant
// Same panels, same six products, a wider block tile: 128 (M) x 256 (N), four multiplying waves of 64 x 128 (2 x 4
// accumulators of v_mfma_f32_32x32x16_bf16 = 128 accumulator registers) + three LDS-DMA waves, BK = 16 (one 16-k step
// per k-tile), four 36-KiB stages.  Per MFMA this moves a quarter fewer bytes global -> LDS (DMA) and LDS -> registers
// (18 fragment reads per 48 MFMAs instead of 12 per 24): the LDS port, which the 128 x 128 kernel keeps ~75 % busy
// (reads + DMA writes) beside a power-bound matrix pipe, drops to ~56 %.
// 128 accumulator registers leave no room for double-buffered fragments (2 x 72); four of the six planes are
// single-buffered and only A0 / B0 - used by the last product of a step and the first of the next - have two sets (96 registers in all).  What
// hides the LDS latency is the order of the six products and refilling every fragment right after its last use (see
// ASRK_STEP below): every fragment of step k + 1 is requested during step k, >= 512 cycles before its first use.
template <int NST, int NDW>
__global__ __launch_bounds__((4 + NDW) * 64) void gemm_bf16x6_w256_kernel(SplitGemmArgs p, int rb_b, int BAND) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    constexpr int NPL = 3, NC = 2, CHUNK = NPL * PIECE;
    constexpr int REGION = NC * CHUNK;           // 6 KiB: 64 rows x 16 k x 3 planes
    constexpr int NRG = 6;                       // A rows 0..63, 64..127, B rows 0..63, ..., 192..255
    constexpr int STAGE = NRG * REGION;          // 36 KiB
    constexpr int LPT = NRG * NC * NPL / NDW;    // LDS-DMA instructions (1-KiB pieces) per DMA wave and k-tile: 12 or 9
    static_assert(LPT * NDW == NRG * NC * NPL && (NDW == 3 || NDW == 4), "36 pieces per stage over 3 or 4 DMA waves");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = p.tiles_m * p.tiles_n, bid = blockIdx.x;
    const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = bid & 7, loc = bid >> 3;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    // BAND tiles of 256 columns per band (default 2: measured 1 % ahead of 4 = the 1024-column bands of the 128 x 128 kernel)
    const int band = tile / (BAND * p.tiles_m);
    const int band_w = min(BAND, p.tiles_n - band * BAND);
    const int in_band = tile - band * BAND * p.tiles_m;
    const int tm = in_band / band_w, tn = band * BAND + in_band % band_w;
    const int nk = p.nk;                         // 16-k tiles

    if (wave >= 4) {
        // ---- DMA wave d = wave - 4 fills pieces [d * LPT, (d + 1) * LPT) of every stage (a stage = 6 regions of 6
        // pieces, region r at r * REGION): they lie in two consecutive regions, ra and ra + 1.  An LDS-DMA instruction
        // costs the SIMD it is issued on ~35 cycles of MFMA issue; four DMA waves put 9 on every SIMD instead of 12 on three
        // Issued as runs of three pieces that are contiguous in global memory AND in LDS (never crossing a region): one
        // wave-uniform base + one M0 value per run, the instruction's immediate offset (applied to both addresses)
        // selects the piece - no per-piece address arithmetic in the DMA waves
        const int d = wave - 4, first = d * LPT;
        constexpr int NRUN = LPT / 3;
        static_assert(NRUN * 3 == LPT, "runs of three pieces");
        const unsigned char *run_g[NRUN];
#pragma unroll
        for (int k = 0; k < NRUN; ++k) {
            const int idx = first + 3 * k, r = idx / (NC * NPL), j = idx - r * (NC * NPL);      // j is 0 or 3
            run_g[k] = (r < 2 ? p.Ap + (size_t)(tm * 2 + r) * p.rb_stride_a
                              : p.Bp + (size_t)min(tn * 4 + r - 2, rb_b - 1) * p.rb_stride_b) + j * PIECE;
        }
        const unsigned voff = lane * 16;
        unsigned char *ldst = lds + first * PIECE;
        auto issue = [&](int kt, int stage) {
            unsigned char *l = ldst + stage * STAGE;
            const size_t gk = (size_t)kt * REGION;
#pragma unroll
            for (int k = 0; k < NRUN; ++k) glds16_run<3>(run_g[k] + gk, voff, l + 3 * k * PIECE);
        };
#pragma unroll
        for (int s = 0; s < NST; ++s)
            if (s < nk) issue(s, s);
        // tiles 0 and 1 must have landed before the first barrier (the workers read A2 / B0 of tile 1 after it)
        {
            const int later = min(NST - 2, nk - 2);
            if (later >= 2) wait_vm<2 * LPT>();
            else if (later == 1) wait_vm<LPT>();
            else wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();            // prologue barrier
        int stage = 0;
        for (int kt = 0; kt < nk; ++kt) {
            // barrier kt: every worker has read stage `stage` (tile kt) completely, and tile kt + 1 has landed;
            // the workers go on to read A2 / B0 of tile kt + 1 right after it -> tile kt + 1 must be there NOW,
            // and tile kt + 2 by the next barrier
            const int later = min(NST - 2, nk - 2 - kt);     // tiles younger than kt + 1 still in flight are fine
            if (later >= 2) wait_vm<2 * LPT>();
            else if (later == 1) wait_vm<LPT>();
            else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            if (kt + NST < nk) issue(kt + NST, stage);
            if (++stage == NST) stage = 0;
        }
        return;
    }

    // ---- multiplying wave: rows wr*64.., columns wc*128..
    const int wr = wave >> 1, wc = wave & 1;
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int frag_off = ((lane >> 5) * NPL) * PIECE + (lane & 31) * 16;
    const unsigned char *abase = lds + wr * REGION + frag_off;
    const unsigned char *bbase = lds + (2 + wc * 2) * REGION + frag_off;   // two consecutive B regions
    // fragment registers: planes A1, A2, B1, B2 single-buffered; A0 and B0 - with the product order below the two planes
    // that are used by the LAST product of a step and by the first two of the next - ping-pong between two sets (96
    // registers in all)
    bf16x8 fa1[2], fa2[2], fb1[4], fb2[4], fa0[2][2], fb0[2][4];
    auto rd_a1 = [&](bf16x8 &dst, int stage, int pl, int i) {
        dst = *reinterpret_cast<const bf16x8 *>(abase + stage * STAGE + pl * PIECE + i * 512);
    };
    auto rd_b1 = [&](bf16x8 &dst, int stage, int pl, int j) {
        dst = *reinterpret_cast<const bf16x8 *>(bbase + stage * STAGE + (j >> 1) * REGION + pl * PIECE + (j & 1) * 512);
    };
    auto rd_a = [&](bf16x8 (&dst)[2], int stage, int pl) {
#pragma unroll
        for (int i = 0; i < 2; ++i) rd_a1(dst[i], stage, pl, i);
    };
    auto rd_b = [&](bf16x8 (&dst)[4], int stage, int pl) {
#pragma unroll
        for (int j = 0; j < 4; ++j) rd_b1(dst[j], stage, pl, j);
    };
#define ASRK_TERM8(FA, FB)                                                                                 \
    do {                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j)         \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[i], FB[j], acc[i][j], 0, 0, 0);          \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    } while (0)
    // a product whose eight MFMAs are interleaved with LDS reads: RD(m) runs right after MFMA m, so that the reads
    // issue in the MFMAs' shadow (a clump of reads between two products idles the matrix pipe while they issue)
#define ASRK_TERM8_RD(FA, FB, RD)                                                                          \
    do {                                                                                                   \
        _Pragma("unroll") for (int m = 0; m < 8; ++m) {                                                    \
            acc[m >> 2][m & 3] =                                                                           \
                __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[m >> 2], FB[m & 3], acc[m >> 2][m & 3], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            RD(m);                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                             \
        }                                                                                                  \
    } while (0)
    // One 16-k step on buffer set P (A0 / B0 of this step) - the next step's live in set Q = 1 - P.  Products and the reads
    // (all of the NEXT step's fragments, from the next stage) issued inside them:
    //   T1 (A2,B0)  -                  | lgkmcnt(0), k-tile barrier: every read of this step's stage is complete, the DMA
    //   T2 (A0,B2)  A2' B0'[0,1]       |   waves refill it under T2..T6; the next tile has landed
    //   T3 (A0,B1)  B2'                  (B2 died with T2)
    //   T4 (A1,B1)  B0'[2,3] A0'         (set Q)
    //   T5 (A1,B0)  B1'                  (B1 died with T4)
    //   T6 (A0,B0)  A1'                  (A1 died with T5)
    // At most one read per two MFMAs, and every fragment is requested at least two products (512 cycles) before its
    // first use.  The last step reads a stale stage into registers nobody uses (no branch around the reads).
#define ASRK_STEP(P)                                                                                       \
    do {                                                                                                   \
        int nstage = stage + 1;                                                                            \
        if (nstage == NST) nstage = 0;                                                                     \
        ASRK_TERM8(fa2, fb0[P]);                                                                           \
        wait_lgkm0();                                                                                      \
        __builtin_amdgcn_s_barrier();                                           \
        auto rd2 = [&](int m) {                                                                            \
            if (m == 0) rd_a1(fa2[0], nstage, 2, 0);                                                       \
            else if (m == 2) rd_a1(fa2[1], nstage, 2, 1);                                                  \
            else if (m == 4) rd_b1(fb0[1 - (P)][0], nstage, 0, 0);                                         \
            else if (m == 6) rd_b1(fb0[1 - (P)][1], nstage, 0, 1);                                         \
        };                                                                                                 \
        ASRK_TERM8_RD(fa0[P], fb2, rd2);                                                                   \
        auto rd3 = [&](int m) {                                                                            \
            if ((m & 1) == 0) rd_b1(fb2[m >> 1], nstage, 2, m >> 1);                                       \
        };                                                                                                 \
        ASRK_TERM8_RD(fa0[P], fb1, rd3);                                                                   \
        auto rd4 = [&](int m) {                                                                            \
            if (m == 0) rd_b1(fb0[1 - (P)][2], nstage, 0, 2);                                              \
            else if (m == 2) rd_b1(fb0[1 - (P)][3], nstage, 0, 3);                                         \
            else if (m == 4) rd_a1(fa0[1 - (P)][0], nstage, 0, 0);                                         \
            else if (m == 6) rd_a1(fa0[1 - (P)][1], nstage, 0, 1);                                         \
        };                                                                                                 \
        ASRK_TERM8_RD(fa1, fb1, rd4);                                                                      \
        auto rd5 = [&](int m) {                                                                            \
            if ((m & 1) == 0) rd_b1(fb1[m >> 1], nstage, 1, m >> 1);                                       \
        };                                                                                                 \
        ASRK_TERM8_RD(fa1, fb0[P], rd5);                                                                   \
        auto rd6 = [&](int m) {                                                                            \
            if (m == 1) rd_a1(fa1[0], nstage, 1, 0);                                                       \
            else if (m == 4) rd_a1(fa1[1], nstage, 1, 1);                                                  \
        };                                                                                                 \
        ASRK_TERM8_RD(fa0[P], fb0[P], rd6);                                                                \
        stage = nstage;                                                                                    \
    } while (0)

    __builtin_amdgcn_s_barrier();                // prologue: tiles 0 and 1 are in LDS
    rd_a(fa2, 0, 2); rd_b(fb0[0], 0, 0); rd_a(fa0[0], 0, 0); rd_b(fb2, 0, 2); rd_b(fb1, 0, 1); rd_a(fa1, 0, 1);
    int stage = 0;
    for (int kt = 0; kt < nk; kt += 2) {         // nk is even (two 16-k tiles per 32-k panel tile)
        ASRK_STEP(0);
        ASRK_STEP(1);
    }
#undef ASRK_STEP
#undef ASRK_TERM8_RD
#undef ASRK_TERM8

    const int row0 = tm * 128 + wr * 64 + 4 * (lane >> 5), col0 = tn * 256 + wc * 128 + (lane & 31);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = col0 + j * 32;
        if (col >= p.N) continue;
        float bsum = 0.f;
        if (p.bias) bsum += p.bias[col];
        if (p.bias2) bsum += p.bias2[col];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                if (row >= p.M) continue;
                float *c = p.C + (size_t)row * p.ldc + col;
                float v = p.alpha * acc[i][j][r] + bsum;
                if (p.beta != 0.f) v += p.beta * *c;
                *c = v;
            }
    }
}

template <int NST, int NDW>
int launch_split_gemm_w256(const SplitGemmArgs &a, int rb_b, hipStream_t s) {
    constexpr int lds = NST * 6 * 2 * 3 * PIECE;
    auto kern = gemm_bf16x6_w256_kernel<NST, NDW>;
    static AsrkLdsLatch latch;
    ASRK_HIP(asrk_max_lds_once(latch, reinterpret_cast<const void *>(kern), lds));
    const int band = 2;                          // tile-order band width in 256-column tiles
    hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3((4 + NDW) * 64), lds, s, a, rb_b, band);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

// ---------------------------------------------------------------------------------- host side
// No state here: the panel workspace is the caller's (asrk_gemm_ws_bytes), the split mode is a call flag.
template <int NC, int NST, bool SPEC, int WM>
int launch_split_gemm(const SplitGemmArgs &a, hipStream_t s) {
    constexpr int lds = NST * (WM + 2) * NC * 3 * PIECE;
    auto kern = gemm_bf16x6_kernel<NC, NST, SPEC, WM>;
    static AsrkLdsLatch latch;
    ASRK_HIP(asrk_max_lds_once(latch, reinterpret_cast<const void *>(kern), lds));
    hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(SPEC ? (WM == 2 ? 512 : 704) : 256), lds, s, a);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

}  // namespace

namespace {
constexpr int SPLIT_NC = 4;                      // k-tile = 32
constexpr int NPL = 3;                           // bf16 planes per operand

struct PanelGeom {
    int KC, rb;                                  // 8-k groups per row (padded to the k-tile), 64-row blocks
    size_t rb_stride, bytes;
};
// rows are padded to whole 128-row tiles, K to whole k-tiles (the split pass writes zeros there)
// slack: one more (zero) k-tile, so that a k range starting at any multiple of 8 can run its last k-tile past K.
// rb_stride carries 4352 bytes of padding (odd in units of 256 B): at a given k the row blocks a chip works on
// concurrently start in different L2 / memory channels
PanelGeom panel_geom(int rows, int K, bool slack = false) {
    PanelGeom g;
    g.KC = (asrk_div_up(K, 8 * SPLIT_NC) + (slack ? 1 : 0)) * SPLIT_NC;
    g.rb = asrk_div_up(rows, 128) * 2;
    g.rb_stride = (size_t)g.KC * NPL * PIECE + 4352;
    g.bytes = (size_t)g.rb * g.rb_stride;
    return g;
}

// panel rows = `rows` of the logical [rows][K] operand; trans: src is stored [K][rows]
int run_split(const float *src, int ld, int rows, int K, bool trans, unsigned char *dstp, const PanelGeom &g,
              hipStream_t s) {
    // one wave per (row block, 4 chunk columns) either way
    const dim3 grid((unsigned)asrk_div_up64((int64_t)g.rb * (g.KC / 4), 4));
    // HBM-bound streaming pass: 4 B read per source element, 6 B written per (padded) panel element
    asrk_prof_work_(PROF_SPLIT, 4.0 * (double)rows * (double)K + 6.0 * (double)g.rb * 64.0 * (double)g.KC * 8.0);
    asrk_prof_begin_(PROF_SPLIT, s);
    if (trans) {
        hipLaunchKernelGGL((split_panel_t_kernel<4>), grid, dim3(256), 0, s, src, ld, rows, K, dstp, g.KC, g.rb, g.rb_stride);
    } else {
        const bool vec = (reinterpret_cast<uintptr_t>(src) & 15) == 0 && ld % 4 == 0;
        if (vec) hipLaunchKernelGGL((split_panel_kernel<true>), grid, dim3(256), 0, s, src, ld, rows, K, dstp, g.KC, g.rb, g.rb_stride);
        else hipLaunchKernelGGL((split_panel_kernel<false>), grid, dim3(256), 0, s, src, ld, rows, K, dstp, g.KC, g.rb, g.rb_stride);
    }
    asrk_prof_end_(PROF_SPLIT, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

// C[M,N] = alpha * A_rows * B_rows^T + ...: Ap / Bp point at the first row block and chunk column to use
int run_panel_gemm(int M, int N, int nk, float alpha, const unsigned char *Ap, size_t stride_a,
                   const unsigned char *Bp, size_t stride_b, float beta, float *C, int ldc, const float *bias,
                   const float *bias2, hipStream_t s) {
    const AsrkKnobs &kn = asrk_knobs_();
    SplitGemmArgs a;
    a.Ap = Ap; a.Bp = Bp; a.C = C; a.bias = bias; a.bias2 = bias2;
    a.M = M; a.N = N; a.ldc = ldc; a.KC = 0; a.nk = nk; a.rb_stride_a = stride_a; a.rb_stride_b = stride_b;
    a.tiles_m = asrk_div_up(M, 128); a.tiles_n = asrk_div_up(N, 128);
    a.alpha = alpha; a.beta = beta;
    // 128 x 256 tiles (gemm_bf16x6_w256_kernel): a quarter fewer LDS bytes per MFMA; needs enough tiles to fill the
    // chip (>= 2 per CU) and at least two 64-row blocks of B per tile row to make the wider tile worth it
    const int w256 = kn.get(kn.split_w256, 1);
    if (w256 && N >= 512) {
        const int tm = asrk_div_up(M, 128), tn = asrk_div_up(N, 256);
        const int ncu = asrk_cu_count_() > 0 ? asrk_cu_count_() : 256;
        if (w256 == 2 || (long)tm * tn >= 2L * ncu) {
            SplitGemmArgs b = a;
            b.tiles_m = tm; b.tiles_n = tn; b.nk = 2 * nk;
            // Wave quantisation: one workgroup per CU, so tm * tn tiles take ceil(tiles / CUs) rounds.  When the last
            // round would be less than ~60 % full (25600 x 4096: 3200 tiles = 12.5 rounds), the wide kernel takes the
            // row tiles that fill whole rounds and the remaining rows run as 128 x 128 tiles (half the tile time, twice
            // the tiles: a full half-length round instead of a half-empty full-length one).
            const long tiles = (long)tm * tn, rem = tiles % ncu;
            const int tm1 = (int)((tiles - rem) / tn);
            if (w256 == 1 && rem > 0 && rem * 10 < 6L * ncu && tm1 > 0 && tm1 < tm && kn.get(kn.split_tail, 1)) {
                b.tiles_m = tm1; b.M = tm1 * 128;
                int rc = launch_split_gemm_w256<4, 4>(b, 2 * asrk_div_up(N, 128), s);
                if (rc != ASRK_OK) return rc;
                SplitGemmArgs c = a;
                c.Ap = a.Ap + (size_t)tm1 * 2 * a.rb_stride_a;
                c.C = a.C + (size_t)tm1 * 128 * ldc;
                c.M = M - tm1 * 128;
                c.tiles_m = asrk_div_up(c.M, 128);
                return launch_split_gemm<4, 3, true, 2>(c, s);
            }
            return launch_split_gemm_w256<4, 4>(b, 2 * asrk_div_up(N, 128), s);
        }
    }
    return launch_split_gemm<4, 3, true, 2>(a, s);    // BK 32, 3 stages (144 KiB), DMA waves
}
}  // namespace

// bytes between consecutive 64-row blocks of a first-class panel (asrk_split_panel_bytes geometry): for the
// recurrence kernels that write panels themselves (lstm_rec.hip)
extern "C" size_t asrk_split_panel_stride_(int rows, int K) { return panel_geom(rows, K, true).rb_stride; }

// Does asrk_gemm_f32 run this contraction on the split path under `flags` (ASRK_GEMM_SPLIT_*)?
extern "C" int asrk_gemm_takes_split(int M, int N, int K, int flags) {
    const int mode = flags & 3;
    if (mode == ASRK_GEMM_SPLIT_OFF || mode == 3 || K < 8 || M < 1 || N < 1) return 0;
    if (mode == ASRK_GEMM_SPLIT_AUTO) {
        // worth it when the MFMA time saved beats the split pass: both output extents large, K deep
        const double hm = 2.0 * (double)M * (double)N / ((double)M + (double)N);
        // shallow K only when the output is huge (layer-0 input projection, K = 80: 1.05 -> 0.91 ms)
        const bool shallow_ok = K >= 64 && (double)M * (double)N >= 134217728.0;
        if (hm < 1500.0 || (K < 256 && !shallow_ok) || M < 256 || N < 256) return 0;
    }
    return 1;
}

// Bytes of caller-owned, 16-byte aligned device scratch asrk_gemm_f32 needs for this call (the split
// panels of both operands); 0 when the call does not take the split path.
extern "C" size_t asrk_gemm_ws_bytes(int M, int N, int K, int flags) {
    if (!asrk_gemm_takes_split(M, N, K, flags)) return 0;
    return panel_geom(M, K).bytes + panel_geom(N, K).bytes;
}

// Same argument meaning as asrk_gemm_f32 (no split-K).
extern "C" int asrk_gemm_split_run_(int transA, int transB, int M, int N, int K, float alpha, const float *A,
                                    int lda, const float *B, int ldb, float beta, float *C, int ldc,
                                    const float *bias, const float *bias2, void *wsp, int flags, hipStream_t s) {
    const PanelGeom ga = panel_geom(M, K), gb = panel_geom(N, K);
    unsigned char *ws = reinterpret_cast<unsigned char *>(wsp);
    unsigned char *Ap = ws, *Bp = ws + ga.bytes;
    int rc = run_split(A, lda, M, K, transA != 0, Ap, ga, s);
    if (rc != ASRK_OK) return rc;
    // B as stored: transB ? [N][K] : [K][N]; the panel wants rows = n
    rc = run_split(B, ldb, N, K, transB == 0, Bp, gb, s);
    if (rc != ASRK_OK) return rc;
    return run_panel_gemm(M, N, ga.KC / SPLIT_NC, alpha, Ap, ga.rb_stride, Bp, gb.rb_stride, beta, C, ldc, bias,
                          bias2, s);
}

// ---- split panels as first-class operands: split once, multiply several times (dW_ih and dW_hh share dG^T)
// flags: reserved (0).
extern "C" size_t asrk_split_panel_bytes(int rows, int K, int flags) {
    if (rows <= 0 || K <= 0 || flags != 0) return 0;
    return panel_geom(rows, K, true).bytes;
}

extern "C" int asrk_split_panel_f32(const float *src, int ld, int rows, int K, int trans, void *panel, int flags,
                                    void *stream) {
    if (!src || !panel || rows <= 0 || K <= 0 || ld < (trans ? rows : K) || flags != 0) return ASRK_EINVAL;
    if ((reinterpret_cast<uintptr_t>(panel) & 15) != 0) return ASRK_EINVAL;
    // counted in the GEMM family of the optional profiling hooks: the split pass is part of the GEMM's cost
    asrk_prof_begin_(PROF_GEMM, (hipStream_t)stream);
    const int rc = run_split(src, ld, rows, K, trans != 0, reinterpret_cast<unsigned char *>(panel),
                             panel_geom(rows, K, true), (hipStream_t)stream);
    asrk_prof_end_(PROF_GEMM, (hipStream_t)stream);
    return rc;
}

extern "C" int asrk_gemm_panels_f32(int M, int N, int K, float alpha, const void *A_panel, int a_rows, int a_K,
                                    int a_row0, int a_k0, const void *B_panel, int b_rows, int b_K, int b_row0,
                                    int b_k0, float beta, float *C, int ldc, const float *bias,
                                    const float *bias2, int flags, void *stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !A_panel || !B_panel || !C || ldc < N || flags != 0) return ASRK_EINVAL;
    if (a_row0 < 0 || b_row0 < 0 || a_k0 < 0 || b_k0 < 0 || a_row0 % 128 || b_row0 % 128 || a_k0 % 8 || b_k0 % 8)
        return ASRK_EINVAL;
    if (a_row0 + M > a_rows || b_row0 + N > b_rows || a_k0 + K > a_K || b_k0 + K > b_K) return ASRK_EINVAL;
    // rows a_row0 + M .. of the A panel that fall into the last tile are computed but never stored (row < M
    // mask); a ragged last k-tile must run into the zero padding of at least one panel
    if (K % 32 != 0 && a_k0 + K != a_K && b_k0 + K != b_K) return ASRK_ESHAPE;
    const PanelGeom ga = panel_geom(a_rows, a_K, true), gb = panel_geom(b_rows, b_K, true);
    // the k-tiles read must stay inside both panels' padded K
    const int nk = asrk_div_up(K, 32);
    if (a_k0 / 8 + nk * SPLIT_NC > ga.KC || b_k0 / 8 + nk * SPLIT_NC > gb.KC) return ASRK_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_work_(PROF_GEMM, 2.0 * (double)M * (double)N * (double)K);
    asrk_prof_begin_(PROF_GEMM, s);
    const unsigned char *A0 = reinterpret_cast<const unsigned char *>(A_panel);
    const unsigned char *B0 = reinterpret_cast<const unsigned char *>(B_panel);
    const unsigned char *Ap = A0 + (size_t)(a_row0 / 64) * ga.rb_stride + (size_t)(a_k0 / 8) * NPL * PIECE;
    const unsigned char *Bp = B0 + (size_t)(b_row0 / 64) * gb.rb_stride + (size_t)(b_k0 / 8) * NPL * PIECE;
    const int rc = run_panel_gemm(M, N, nk, alpha, Ap, ga.rb_stride, Bp, gb.rb_stride, beta, C, ldc, bias, bias2, s);
    asrk_prof_end_(PROF_GEMM, s);
    return rc;
}
