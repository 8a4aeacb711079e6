// Exact-f32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces the dense contractions the reference dispatches through ATen: the LSTM
// input-to-hidden projection inside nn.LSTM (src/module.py:131), the CTC / character heads
// (src/asr.py:96,220), the attention key/query projections (src/asr.py:280,290) and every
// autograd GEMM of those (dX = dY*W, dW = dY^T*X).
//
// Design (gfx950): 128x128 block tile, BK = 32, 4 waves (2x2), each wave owns a 64x64 sub-tile
// as 2x2 MFMA 32x32 accumulators (64 acc VGPRs).  Global -> register -> LDS staging with a
// double-buffered LDS ring (one barrier per K tile, next tile's global loads issued before the
// MFMA block so HBM latency hides under 64 x 64-cycle MFMAs).  Two LDS operand images:
//   K-contiguous  [row][36]  : operand stored with K fastest in memory; fragment = one
//                              ds_read_b128 (4 consecutive k) per 4 MFMAs, conflict-free
//                              (row stride 144 B -> 16 distinct 16-B slots per lane group);
//   M/N-contiguous [k][136]  : operand stored with M (or N) fastest; fragment = ds_read_b32.
// The MFMA K index is a free permutation as long as A and B agree: inside each group of 8 k's
// MFMA j (0..3) consumes k = 8*quad + 4*(lane>>5) + j from BOTH operands.
// Tile ids are remapped so each XCD (private L2) walks a contiguous run of tiles.
#include "common.h"
#include "knobs.h"
#include <cstdlib>
#include <algorithm>
#include <type_traits>

extern "C" int asrk_cu_count_(void);
extern "C" void asrk_prof_launches_(int id, int64_t n);
extern "C" int asrk_gemm_split_run_(int transA, int transB, int M, int N, int K, float alpha, const float *A,
                                    int lda, const float *B, int ldb, float beta, float *C, int ldc,
                                    const float *bias, const float *bias2, void *ws, int flags, hipStream_t s);

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int KC_LD = 36;          // floats per row, K-contiguous image
constexpr int MC_LD = 136;         // floats per k-row, M-contiguous image (4*LD % 64 == 32:
                                   // the two half-waves (k, k+4) hit disjoint banks)
constexpr int TILE_FLOATS = 4608;  // max(128*36, 32*136)
constexpr int GEMM_LDS_BYTES = 4 * TILE_FLOATS * 4;

struct GemmArgs {
    const float *A, *B;
    float *C;
    const float *bias, *bias2;
    int M, N, K, lda, ldb, ldc;
    float alpha, beta;
    int splitk, k_per_split, tiles_m, tiles_n;
    int dbg;  // ASRK_GEMM_DBG experiments: bit0 skip in-loop global loads, bit1 skip in-loop LDS stores
};

// Load this thread's share of one operand tile (4 x float4) into registers.
// KC: operand stored [R, K] (row = m or n), tile = 128 rows x 32 k.
// !KC: operand stored [K, R], tile = 32 k x 128 rows.
template <bool KC, bool VEC>
__device__ __forceinline__ void load_tile(const float *__restrict__ P, int ld, int R, int r0,
                                          int kbase, int kend, int tid, f32x4 (&reg)[4]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (KC) {
            const int row = r0 + (tid >> 3) + 32 * p;
            const int k = kbase + (tid & 7) * 4;
            if (row < R) {
                const float *g = P + (size_t)row * ld + k;
                if (VEC) {
                    if (k < kend) v = *reinterpret_cast<const f32x4 *>(g);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (k + j < kend) v[j] = g[j];
                }
            }
        } else {
            const int k = kbase + (tid >> 5) + 8 * p;
            const int row = r0 + (tid & 31) * 4;
            if (k < kend) {
                const float *g = P + (size_t)k * ld + row;
                if (VEC) {
                    if (row < R) v = *reinterpret_cast<const f32x4 *>(g);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (row + j < R) v[j] = g[j];
                }
            }
        }
        reg[p] = v;
    }
}

template <bool KC>
__device__ __forceinline__ void store_tile(float *S, int tid, const f32x4 (&reg)[4]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (KC) {
            const int row = (tid >> 3) + 32 * p;
            *reinterpret_cast<f32x4 *>(S + row * KC_LD + (tid & 7) * 4) = reg[p];
        } else {
            const int k = (tid >> 5) + 8 * p;
            *reinterpret_cast<f32x4 *>(S + k * MC_LD + (tid & 31) * 4) = reg[p];
        }
    }
}

// fragment for one 32-row sub tile, quad qd (8 k's): returns 4 floats = the operand for MFMA 0..3
template <bool KC>
__device__ __forceinline__ f32x4 read_frag(const float *S, int row, int qd, int kk) {
    if (KC) {
        return *reinterpret_cast<const f32x4 *>(S + row * KC_LD + qd * 8 + 4 * kk);
    } else {
        f32x4 v;
        const float *b = S + (qd * 8 + 4 * kk) * MC_LD + row;
        v[0] = b[0];
        v[1] = b[MC_LD];
        v[2] = b[2 * MC_LD];
        v[3] = b[3 * MC_LD];
        return v;
    }
}

template <bool A_KC, bool B_KC, bool VEC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // XCD-aware bijective tile remap: block b runs on XCD b%8; give each XCD a contiguous run.
    const int ntiles = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = bid & 7, loc = bid >> 3;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int kbeg = blockIdx.y * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nk = (kend - kbeg + BK - 1) / BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[4], rb[4];
    const int l31 = lane & 31, kk = lane >> 5;
    const int arow = wr * 64 + l31, brow = wc * 64 + l31;

    // Software pipeline (one K tile = 4 quads of 8 k):
    //   global(t+2) --regs--> LDS[(t+1)&1] mid tile t+1 ... fragments of quad q+1 are read from LDS
    //   while the 16 MFMAs of quad q execute (two fragment register sets), across tile boundaries too.
    //   Two barriers per tile, both in the middle of MFMA work: A (everyone is done reading the buffer
    //   about to be overwritten) and B (the new tile is visible); neither drains the MFMA pipe.
#define ASRK_FRAGS(As_, Bs_, qd_, fa_, fb_)                                        \
    do {                                                                          \
        fa_[0] = read_frag<A_KC>(As_, arow, qd_, kk);                             \
        fa_[1] = read_frag<A_KC>(As_, arow + 32, qd_, kk);                        \
        fb_[0] = read_frag<B_KC>(Bs_, brow, qd_, kk);                             \
        fb_[1] = read_frag<B_KC>(Bs_, brow + 32, qd_, kk);                        \
    } while (0)
#define ASRK_MFMA16(fa_, fb_)                                                      \
    do {                                                                          \
        _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                          \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                      \
                _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                  \
                    acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x2f32(           \
                        fa_[i_][s_], fb_[j_][s_], acc[i_][j_], 0, 0, 0);          \
    } while (0)

    f32x4 fa0[2], fb0[2], fa1[2], fb1[2];
    if (nk > 0) {
        load_tile<A_KC, VEC>(p.A, p.lda, p.M, m0, kbeg, kend, tid, ra);
        load_tile<B_KC, VEC>(p.B, p.ldb, p.N, n0, kbeg, kend, tid, rb);
        store_tile<A_KC>(smem, tid, ra);
        store_tile<B_KC>(smem + TILE_FLOATS, tid, rb);
        __syncthreads();
        if (nk > 1) {
            load_tile<A_KC, VEC>(p.A, p.lda, p.M, m0, kbeg + BK, kend, tid, ra);
            load_tile<B_KC, VEC>(p.B, p.ldb, p.N, n0, kbeg + BK, kend, tid, rb);
        }
        ASRK_FRAGS(smem, (smem + TILE_FLOATS), 0, fa0, fb0);
    }

    for (int t = 0; t < nk; ++t) {
        const float *As = smem + (t & 1) * 2 * TILE_FLOATS;
        const float *Bs = As + TILE_FLOATS;
        float *An = smem + ((t + 1) & 1) * 2 * TILE_FLOATS;
        const bool more = (t + 1 < nk);

        ASRK_FRAGS(As, Bs, 1, fa1, fb1);
        ASRK_MFMA16(fa0, fb0);

        ASRK_FRAGS(As, Bs, 2, fa0, fb0);
        ASRK_MFMA16(fa1, fb1);

        if (more) {
            __syncthreads();                     // A: nobody still reads An (tile t-1's image)
            if (!(p.dbg & 2)) {
                store_tile<A_KC>(An, tid, ra);   // tile t+1, loaded 3/4 of a tile ago
                store_tile<B_KC>(An + TILE_FLOATS, tid, rb);
            }
        }

        ASRK_FRAGS(As, Bs, 3, fa1, fb1);
        ASRK_MFMA16(fa0, fb0);

        if (more) {
            __syncthreads();                     // B: tile t+1 is visible
            if (t + 2 < nk && !(p.dbg & 1)) {
                load_tile<A_KC, VEC>(p.A, p.lda, p.M, m0, kbeg + (t + 2) * BK, kend, tid, ra);
                load_tile<B_KC, VEC>(p.B, p.ldb, p.N, n0, kbeg + (t + 2) * BK, kend, tid, rb);
            }
            ASRK_FRAGS(An, (An + TILE_FLOATS), 0, fa0, fb0);
        }
        ASRK_MFMA16(fa1, fb1);
    }
#undef ASRK_FRAGS
#undef ASRK_MFMA16

    // epilogue: C/D map of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool atomic = p.splitk > 1;
    const bool add_bias = (blockIdx.y == 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wc * 64 + j * 32 + l31;
        if (col >= p.N) continue;
        float bv = 0.f;
        if (add_bias) {
            if (p.bias) bv += p.bias[col];
            if (p.bias2) bv += p.bias2[col];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (row >= p.M) continue;
                float *c = p.C + (size_t)row * p.ldc + col;
                const float v = p.alpha * acc[i][j][r] + bv;
                if (atomic) {
                    unsafeAtomicAdd(c, v);
                } else if (p.beta != 0.f) {
                    *c = v + p.beta * (*c);
                } else {
                    *c = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fast path (16-B aligned operands, ld % 4 == 0, contiguous extents % 4 == 0): the same tiling
// and LDS images as gemm_f32_kernel, but
//   * global loads are branch-free: out-of-range rows are CLAMPED to a valid row (their products
//     land in C rows/columns that are never stored), only the one partial K tile pays a select;
//   * per-thread source pointers advance by a constant per K tile (no index arithmetic in the loop);
//   * every quad (16 MFMAs) carries its share of the tile's memory instructions and
//     sched_group_barrier pins the interleave (1 LDS/VMEM instruction per 2-4 MFMAs), so the matrix
//     pipe keeps issuing while fragments, staging stores and global loads are in flight.
// Launch hint (ASRK_GEMM_LDS_HINT(kib) in asrk_gemm_f32's `flags`): minimum dynamic-LDS request in KiB for the tiled kernels.
// The host layer sets 96 around the weight-gradient GEMMs it launches on its side stream: they share
// the chip with latency-critical kernels of the main stream (the persistent BPTT, the dX GEMM), and
// at one workgroup per CU instead of two they give CUs back sooner when those are launched — measured
// 27.5 -> 25.9 ms/step at cfg2.  0 = no hint (two workgroups per CU).

template <bool KC>
struct TileSrc {
    const float *ptr[4];  // this thread's 4 x 16-B pieces of the current K tile
    int kpos[4];          // k index of piece p relative to the tile start
};

template <bool KC>
__device__ __forceinline__ void src_init(TileSrc<KC> &s, const float *P, int ld, int R, int K, int r0,
                                         int kbeg, int tid) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (KC) {
            const int row = min(r0 + (tid >> 3) + 32 * p, R - 1);
            s.kpos[p] = (tid & 7) * 4;
            s.ptr[p] = P + (size_t)row * ld + kbeg + s.kpos[p];
        } else {
            const int row = min(r0 + (tid & 31) * 4, R - 4);
            s.kpos[p] = (tid >> 5) + 8 * p;
            s.ptr[p] = P + (size_t)(kbeg + s.kpos[p]) * ld + row;
        }
    }
}

// CHK: the tile may run past kend (or past K): clamp the address, zero what is out of range
template <bool KC, bool CHK>
__device__ __forceinline__ void src_load(const TileSrc<KC> &s, int ld, int K, int ktile, int kend,
                                         f32x4 (&reg)[4]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (!CHK) {
            reg[p] = *reinterpret_cast<const f32x4 *>(s.ptr[p]);
        } else {
            const int k = ktile + s.kpos[p];
            if (KC) {
                const int over = max(0, k - (K - 4));      // keep the 16-B read inside the row
                f32x4 v = *reinterpret_cast<const f32x4 *>(s.ptr[p] - over);
                if (k >= kend) v = f32x4{0.f, 0.f, 0.f, 0.f};
                reg[p] = v;
            } else {
                const int over = max(0, k - (K - 1));
                f32x4 v = *reinterpret_cast<const f32x4 *>(s.ptr[p] - (size_t)over * ld);
                if (k >= kend) v = f32x4{0.f, 0.f, 0.f, 0.f};
                reg[p] = v;
            }
        }
    }
}

template <bool KC>
__device__ __forceinline__ void src_advance(TileSrc<KC> &s, int ld) {
#pragma unroll
    for (int p = 0; p < 4; ++p) s.ptr[p] += KC ? BK : (size_t)BK * ld;
}

#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
constexpr int SG_MFMA = 0x008, SG_VMEM_RD = 0x020, SG_DS_RD = 0x100, SG_DS_WR = 0x200;

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_fast_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int ntiles = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = bid & 7, loc = bid >> 3;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int kbeg = blockIdx.y * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int nk = (kend - kbeg + BK - 1) / BK;
    const bool tail = ((kend - kbeg) % BK) != 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int l31 = lane & 31, kk = lane >> 5;
    const int arow = wr * 64 + l31, brow = wc * 64 + l31;
    TileSrc<A_KC> sa;
    TileSrc<B_KC> sb;
    src_init<A_KC>(sa, p.A, p.lda, p.M, p.K, m0, kbeg, tid);
    src_init<B_KC>(sb, p.B, p.ldb, p.N, p.K, n0, kbeg, tid);
    f32x4 ra[4], rb[4];
    f32x4 fa0[2], fb0[2], fa1[2], fb1[2];

#define FRAGS(As_, Bs_, qd_, fa_, fb_)                                            \
    do {                                                                          \
        fa_[0] = read_frag<A_KC>(As_, arow, qd_, kk);                             \
        fa_[1] = read_frag<A_KC>(As_, arow + 32, qd_, kk);                        \
        fb_[0] = read_frag<B_KC>(Bs_, brow, qd_, kk);                             \
        fb_[1] = read_frag<B_KC>(Bs_, brow + 32, qd_, kk);                        \
    } while (0)
#define MFMA16(fa_, fb_)                                                          \
    do {                                                                          \
        _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_)                          \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                      \
                _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                  \
                    acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x2f32(           \
                        fa_[i_][s_], fb_[j_][s_], acc[i_][j_], 0, 0, 0);          \
    } while (0)
    // number of LDS read instructions one FRAGS issues (b128 per K-contiguous fragment, 4 x b32 else)
    constexpr int NRD = (A_KC ? 2 : 8) + (B_KC ? 2 : 8);
    // interleave NRD LDS reads with 16 MFMAs
#define PIN_READS()                                                               \
    do {                                                                          \
        if (NRD == 4) {                                                           \
            _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) { SGB(SG_DS_RD, 1); SGB(SG_MFMA, 4); } \
        } else if (NRD == 10) {                                                   \
            _Pragma("unroll") for (int g_ = 0; g_ < 5; ++g_) { SGB(SG_DS_RD, 2); SGB(SG_MFMA, 3); } \
            SGB(SG_MFMA, 1);                                                      \
        } else {                                                                  \
            _Pragma("unroll") for (int g_ = 0; g_ < 16; ++g_) { SGB(SG_DS_RD, 1); SGB(SG_MFMA, 1); } \
        }                                                                         \
    } while (0)

    if (nk > 0) {
        if (nk == 1 && tail) {
            src_load<A_KC, true>(sa, p.lda, p.K, kbeg, kend, ra);
            src_load<B_KC, true>(sb, p.ldb, p.K, kbeg, kend, rb);
        } else {
            src_load<A_KC, false>(sa, p.lda, p.K, kbeg, kend, ra);
            src_load<B_KC, false>(sb, p.ldb, p.K, kbeg, kend, rb);
        }
        src_advance<A_KC>(sa, p.lda);
        src_advance<B_KC>(sb, p.ldb);
        store_tile<A_KC>(smem, tid, ra);
        store_tile<B_KC>(smem + TILE_FLOATS, tid, rb);
        __syncthreads();
        if (nk > 1) {
            if (nk == 2 && tail) {
                src_load<A_KC, true>(sa, p.lda, p.K, kbeg + BK, kend, ra);
                src_load<B_KC, true>(sb, p.ldb, p.K, kbeg + BK, kend, rb);
            } else {
                src_load<A_KC, false>(sa, p.lda, p.K, kbeg + BK, kend, ra);
                src_load<B_KC, false>(sb, p.ldb, p.K, kbeg + BK, kend, rb);
            }
            src_advance<A_KC>(sa, p.lda);
            src_advance<B_KC>(sb, p.ldb);
        }
        FRAGS(smem, (smem + TILE_FLOATS), 0, fa0, fb0);
    }

    // one K tile: STORE = a next tile exists (stage it), LOAD = a tile after that exists (fetch it),
    // CHK = that tile is the partial one
    auto body = [&](int t, auto store_c, auto load_c, auto chk_c) {
        constexpr bool STORE = decltype(store_c)::value, LOAD = decltype(load_c)::value,
                       CHK = decltype(chk_c)::value;
        const float *As = smem + (t & 1) * 2 * TILE_FLOATS;
        const float *Bs = As + TILE_FLOATS;
        float *An = smem + ((t + 1) & 1) * 2 * TILE_FLOATS;

        FRAGS(As, Bs, 1, fa1, fb1);
        MFMA16(fa0, fb0);
        PIN_READS();

        FRAGS(As, Bs, 2, fa0, fb0);
        MFMA16(fa1, fb1);
        PIN_READS();

        if (STORE) {
            __syncthreads();                     // A: nobody still reads An (tile t-1's image)
            store_tile<A_KC>(An, tid, ra);       // tile t+1, loaded 3/4 of a tile ago
            store_tile<B_KC>(An + TILE_FLOATS, tid, rb);
        }
        FRAGS(As, Bs, 3, fa1, fb1);
        MFMA16(fa0, fb0);
        if (STORE) {
            // 8 staging stores + NRD fragment reads spread over the 16 MFMAs
            _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) { SGB(SG_DS_WR, 1); SGB(SG_MFMA, 1); }
            if (NRD == 4) {
                _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) { SGB(SG_DS_RD, 1); SGB(SG_MFMA, 2); }
            } else {
                _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) { SGB(SG_DS_RD, NRD / 8 + 1); SGB(SG_MFMA, 1); }
            }
        } else {
            PIN_READS();
        }

        if (STORE) {
            __syncthreads();                     // B: tile t+1 is visible
            if (LOAD) {
                src_load<A_KC, CHK>(sa, p.lda, p.K, kbeg + (t + 2) * BK, kend, ra);
                src_load<B_KC, CHK>(sb, p.ldb, p.K, kbeg + (t + 2) * BK, kend, rb);
                src_advance<A_KC>(sa, p.lda);
                src_advance<B_KC>(sb, p.ldb);
            }
            FRAGS(An, (An + TILE_FLOATS), 0, fa0, fb0);
        }
        MFMA16(fa1, fb1);
        if (STORE) {
            if (LOAD) {
                _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) { SGB(SG_VMEM_RD, 1); SGB(SG_MFMA, 1); }
                if (NRD == 4) {
                    _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) { SGB(SG_DS_RD, 1); SGB(SG_MFMA, 2); }
                } else {
                    _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) { SGB(SG_DS_RD, NRD / 8 + 1); SGB(SG_MFMA, 1); }
                }
            } else {
                PIN_READS();
            }
        }
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    int t = 0;
    const int n_plain = nk - 2 - (tail ? 1 : 0);     // tiles whose t+2 successor is a full tile
    for (; t < n_plain; ++t) body(t, T_{}, T_{}, F_{});
    for (; t < nk; ++t) {
        if (t + 2 < nk) body(t, T_{}, T_{}, T_{});
        else if (t + 1 < nk) body(t, T_{}, F_{}, F_{});
        else body(t, F_{}, F_{}, F_{});
    }
#undef FRAGS
#undef MFMA16
#undef PIN_READS

    // epilogue: C/D map of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool atomic = p.splitk > 1;
    const bool add_bias = (blockIdx.y == 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wc * 64 + j * 32 + l31;
        if (col >= p.N) continue;
        float bv = 0.f;
        if (add_bias) {
            if (p.bias) bv += p.bias[col];
            if (p.bias2) bv += p.bias2[col];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (row >= p.M) continue;
                float *c = p.C + (size_t)row * p.ldc + col;
                const float v = p.alpha * acc[i][j][r] + bv;
                if (atomic) {
                    unsafeAtomicAdd(c, v);
                } else if (p.beta != 0.f) {
                    *c = v + p.beta * (*c);
                } else {
                    *c = v;
                }
            }
        }
    }
}
#undef SGB

template <bool A_KC, bool B_KC>
int launch_gemm_fast(const GemmArgs &a, int lds_hint_kib, hipStream_t s) {
    static AsrkLdsLatch latch;
    auto kern = gemm_f32_fast_kernel<A_KC, B_KC>;
    const int BG_LDS = std::min(158 * 1024, std::max(GEMM_LDS_BYTES, lds_hint_kib * 1024));
    ASRK_HIP(asrk_max_lds_once(latch, reinterpret_cast<const void *>(kern), 158 * 1024));
    dim3 grid(a.tiles_m * a.tiles_n, a.splitk, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), BG_LDS, s, a);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

// ---------------------------------------------------------------------------------------------
// Skinny-M path (M <= 32: decoder steps, beam-search batches, RNN-LM steps).  These GEMMs stream
// the weight matrix once and are HBM-bound, so the kernel is built around the stream, not the
// tile: no LDS, no barriers, every wave owns an output slab and a K range (split-K fills the
// chip), loads its operands straight into MFMA-fragment registers as 16-B pieces (>= 128 B
// contiguous per row per wave-load) one chunk ahead, and the free K / column permutations of the
// MFMA make the register layout match what coalesced loads deliver.
struct SkinnyArgs {
    const float *A, *B;
    float *C;
    const float *bias, *bias2;
    int M, N, K, lda, ldb, ldc;
    float alpha;
    int splitk, k_per_split, slabs;
    int overwrite;   // single K range and beta == 0: store instead of accumulate (C not pre-zeroed)
};

__device__ __forceinline__ f32x4 ld4_or_zero(const float *p, bool ok) {
    f32x4 v = *reinterpret_cast<const f32x4 *>(p);
    if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
    return v;
}

// NT: C[32-block, 32-slab] += A[M,K] . B[N,K]^T ; lane = (row|col = lane%32, k-half = lane/32);
// per 32-k chunk each lane holds 16 consecutive k of its A row and its B row.
__global__ __launch_bounds__(256) void gemm_skinny_nt_kernel(SkinnyArgs p) {
    __shared__ float red[3][16][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slab = blockIdx.x;
    const int l31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.z * 32, n0 = slab * 32;
    // the block's K range is cut 4 ways over its waves; their partial tiles meet in LDS, so only one
    // wave per block touches C (4x fewer atomics than giving every wave its own split)
    const int bk0 = blockIdx.y * p.k_per_split, bk1 = min(p.K, bk0 + p.k_per_split);
    const int per_wave = ((bk1 - bk0 + 127) / 128) * 32;
    const int kbeg = min(bk1, bk0 + wave * per_wave), kend = min(bk1, kbeg + per_wave);
    const float *ap = p.A + (size_t)min(m0 + l31, p.M - 1) * p.lda + 16 * h;
    const float *bp = p.B + (size_t)min(n0 + l31, p.N - 1) * p.ldb + 16 * h;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 a[2][4], b[2][4];
    auto load = [&](int kb, f32x4 (&ar)[4], f32x4 (&br)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = kb + 16 * h + 4 * i;
            const bool ok = k < kend;
            const int kc = ok ? kb + 4 * i : 0;          // clamped offset stays inside the row
            ar[i] = ld4_or_zero(ap + (ok ? kc : -16 * h), ok);
            br[i] = ld4_or_zero(bp + (ok ? kc : -16 * h), ok);
        }
    };
    auto fma16 = [&](const f32x4 (&ar)[4], const f32x4 (&br)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[i][j], br[i][j], acc, 0, 0, 0);
    };
    // two chunks per trip so both register buffers are addressed statically; loads past kend
    // return zeros from a clamped (valid) address
    load(kbeg, a[0], b[0]);
    for (int kb = kbeg; kb < kend; kb += 64) {
        load(kb + 32, a[1], b[1]);
        fma16(a[0], b[0]);
        load(kb + 64, a[0], b[0]);
        fma16(a[1], b[1]);
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += red[0][r][lane] + red[1][r][lane] + red[2][r][lane];
    const int col = n0 + l31;
    if (col >= p.N) return;
    float bv = 0.f;
    if (blockIdx.y == 0) {
        if (p.bias) bv += p.bias[col];
        if (p.bias2) bv += p.bias2[col];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= p.M) continue;
        float *c = p.C + (size_t)row * p.ldc + col;
        const float v = p.alpha * acc[r] + bv;
        if (p.splitk > 1) unsafeAtomicAdd(c, v);
        else if (p.overwrite) *c = v;
        else *c += v;                       // C holds beta*C on entry
    }
}

// NN: C[32-block, 128-slab] += A[M,K] . B[K,N] ; lane c = lane%32 owns columns n0+4c..4c+3 (one 16-B
// piece per B row: a wave-load is 2 rows x 512 contiguous bytes); MFMA q computes columns {4c+q}.
__global__ __launch_bounds__(256) void gemm_skinny_nn_kernel(SkinnyArgs p) {
    __shared__ float red[3][64][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slab = blockIdx.x;
    const int l31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.z * 32, n0 = slab * 128;
    const int bk0 = blockIdx.y * p.k_per_split, bk1 = min(p.K, bk0 + p.k_per_split);
    const int per_wave = ((bk1 - bk0 + 63) / 64) * 16;
    const int kbeg = min(bk1, bk0 + wave * per_wave), kend = min(bk1, kbeg + per_wave);
    const int ncol = min(n0 + 4 * l31, p.N - 4);
    const float *ap = p.A + (size_t)min(m0 + l31, p.M - 1) * p.lda;
    const float *bp = p.B + ncol;
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    // one chunk = 16 k: 2 blocks of 8 (half h takes k = 8*blk + 4h + j)
    f32x4 a[2][2], b[2][2][4];
    auto load = [&](int kb, f32x4 (&ar)[2], f32x4 (&br)[2][4]) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int k0 = kb + 8 * blk + 4 * h;
            const bool ok = k0 < kend;                    // kend % 4 == 0: all four k's or none
            ar[blk] = ld4_or_zero(ap + (ok ? k0 : 0), ok);
#pragma unroll
            for (int j = 0; j < 4; ++j) br[blk][j] = ld4_or_zero(bp + (size_t)(ok ? k0 + j : 0) * p.ldb, ok);
        }
    };
    auto fma32 = [&](const f32x4 (&ar)[2], const f32x4 (&br)[2][4]) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[blk][j], br[blk][j][q], acc[q], 0, 0, 0);
    };
    load(kbeg, a[0], b[0]);
    for (int kb = kbeg; kb < kend; kb += 32) {
        load(kb + 16, a[1], b[1]);
        fma32(a[0], b[0]);
        load(kb + 32, a[0], b[0]);
        fma32(a[1], b[1]);
    }
    if (wave > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave - 1][q * 16 + r][lane] = acc[q][r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[q][r] += red[0][q * 16 + r][lane] + red[1][q * 16 + r][lane] + red[2][q * 16 + r][lane];
    if (n0 + 4 * l31 >= p.N) return;       // clamped lanes duplicate valid columns: do not store
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int col = ncol + q;
        float bv = 0.f;
        if (blockIdx.y == 0) {
            if (p.bias) bv += p.bias[col];
            if (p.bias2) bv += p.bias2[col];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row >= p.M) continue;
            float *c = p.C + (size_t)row * p.ldc + col;
            const float v = p.alpha * acc[q][r] + bv;
            if (p.splitk > 1) unsafeAtomicAdd(c, v);
            else if (p.overwrite) *c = v;
            else *c += v;
        }
    }
}

template <bool A_KC, bool B_KC, bool VEC>
int launch_gemm(const GemmArgs &a, hipStream_t s) {
    static AsrkLdsLatch latch;
    auto kern = gemm_f32_kernel<A_KC, B_KC, VEC>;
    ASRK_HIP(asrk_max_lds_once(latch, reinterpret_cast<const void *>(kern), GEMM_LDS_BYTES));
    dim3 grid(a.tiles_m * a.tiles_n, a.splitk, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), GEMM_LDS_BYTES, s, a);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

__global__ void scale_rows_kernel(float *C, int M, int N, int ldc, float beta) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * N) return;
    const int r = (int)(i / N), c = (int)(i - (int64_t)r * N);
    C[(size_t)r * ldc + c] *= beta;
}

}  // namespace

extern "C" int asrk_gemm_f32(int transA, int transB, int M, int N, int K, float alpha,
                             const float *A, int lda, const float *B, int ldb, float beta,
                             float *C, int ldc, const float *bias, const float *bias2,
                             int splitk, int flags, void *ws, size_t ws_bytes, void *stream) {
    if (M < 0 || N < 0 || K < 0 || flags < 0) return ASRK_EINVAL;
    const AsrkKnobs &kn = asrk_knobs_();
    const int lds_hint = (flags >> 8) & 0xff;
    if (M == 0 || N == 0) return ASRK_OK;
    if (!A || !B || !C) return ASRK_EINVAL;
    if (transA && transB) return ASRK_EINVAL;  // TT never occurs on this path
    hipStream_t s = (hipStream_t)stream;
    const int prof_id = lds_hint > 80 ? PROF_GEMM_BG : PROF_GEMM;
    asrk_prof_work_(prof_id, 2.0 * (double)M * (double)N * (double)K);
    const bool a_kc = !transA, b_kc = transB != 0;
    if (lda < (a_kc ? K : M) || ldb < (b_kc ? K : N) || ldc < N) return ASRK_EINVAL;

    {
        // skinny-M path: weight-streaming kernels (see gemm_skinny_*): NT / NN, 16-B accessible operands
        auto al16s = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
        const bool no_skinny = kn.is_set(kn.gemm_noskinny);
        const bool nt = a_kc && b_kc, nn = a_kc && !b_kc;
        if (!no_skinny && M <= 32 && (nt || nn) && K >= 32 && K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
            al16s(A) && al16s(B) && (nt || (N % 4 == 0 && N >= 4)) && (splitk <= 0 || splitk == 1)) {
            SkinnyArgs k;
            k.A = A; k.B = B; k.C = C; k.bias = bias; k.bias2 = bias2;
            k.M = M; k.N = N; k.K = K; k.lda = lda; k.ldb = ldb; k.ldc = ldc; k.alpha = alpha;
            k.slabs = asrk_div_up(N, nt ? 32 : 128);
            const int mblocks = asrk_div_up(M, 32);
            const int chunk = nt ? 32 : 16;
            // one block = one slab x one K range, cut 4 ways over its waves.  Enough blocks to keep
            // every CU streaming (~2 blocks = 8 waves per CU), at least 2 chunks per wave, and at
            // most 8 K ranges per output element (atomic traffic; measured optimum 4-8)
            int want = asrk_div_up(2 * (asrk_cu_count_() > 0 ? asrk_cu_count_() : 256), k.slabs * mblocks);
            int sk = std::max(1, std::min(std::min(want, 8), K / (8 * chunk)));
            if (splitk == 1 || kn.get(kn.deterministic, 0)) sk = 1;   // deterministic: one K range per output element
            else if (kn.is_set(kn.skinny_sk)) sk = std::max(1, kn.skinny_sk);
            k.k_per_split = asrk_div_up(asrk_div_up(K, sk), 4 * chunk) * 4 * chunk;
            k.splitk = asrk_div_up(K, k.k_per_split);
            // the kernels accumulate into C: establish beta*C first (unless one K range overwrites it)
            k.overwrite = (k.splitk == 1 && beta == 0.f) ? 1 : 0;
            if (k.overwrite) {
            } else if (beta == 0.f) {
                ASRK_HIP(hipMemset2DAsync(C, (size_t)ldc * 4, 0, (size_t)N * 4, M, s));
            } else if (beta != 1.f) {
                const int64_t n = (int64_t)M * N;
                hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)asrk_div_up64(n, 256)), dim3(256), 0, s, C, M,
                                   N, ldc, beta);
                ASRK_LAUNCH_CHECK();
            }
            asrk_prof_begin_(prof_id, s);
            const dim3 grid(k.slabs, k.splitk, mblocks);
            if (nt) hipLaunchKernelGGL(gemm_skinny_nt_kernel, grid, dim3(256), 0, s, k);
            else hipLaunchKernelGGL(gemm_skinny_nn_kernel, grid, dim3(256), 0, s, k);
            asrk_prof_end_(prof_id, s);
            ASRK_LAUNCH_CHECK();
            return ASRK_OK;
        }
    }

    if (splitk <= 0) {
        // big contractions: bf16x6 operand splitting on the bf16 matrix cores (gemm_split.hip)
        if (asrk_gemm_takes_split(M, N, K, flags)) {
            // the split panels live in the CALLER's workspace (asrk_gemm_ws_bytes): nothing is allocated here
            if (!ws || ws_bytes < asrk_gemm_ws_bytes(M, N, K, flags)) return ASRK_EWORKSPACE;
            if ((reinterpret_cast<uintptr_t>(ws) & 15) != 0) return ASRK_EINVAL;
            asrk_prof_begin_(prof_id, s);
            const int rc = asrk_gemm_split_run_(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias,
                                                bias2, ws, flags, s);
            asrk_prof_end_(prof_id, s);
            asrk_prof_launches_(prof_id, 2);     // the two split passes
            return rc;
        }
    }

    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.bias2 = bias2;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.alpha = alpha; g.beta = beta;
    g.dbg = kn.get(kn.gemm_dbg, 0);
    g.tiles_m = asrk_div_up(M, BM);
    g.tiles_n = asrk_div_up(N, BN);
    const int tiles = g.tiles_m * g.tiles_n;
    const int kiters = asrk_div_up(K, BK);
    if (splitk <= 0) {
        // Wave-quantisation model: the chip runs `slots` = 2 workgroups per CU at a time, so a
        // launch of W workgroups takes ceil(W/slots) rounds; splitting K by s makes each round
        // 1/s as long but adds atomic read-modify-write traffic on C.  Pick the s that minimises
        //   rounds(tiles*s)/s * (1 + 3% per extra split),   keeping >= 8 K tiles per split.
        splitk = 1;
        const int slots = 2 * (asrk_cu_count_() > 0 ? asrk_cu_count_() : 256);
        const int max_split = kn.get(kn.deterministic, 0) ? 0 : kiters / 8;   // deterministic: no atomic split-K
        if (max_split >= 2 && tiles < 4 * slots) {
            double best = (double)asrk_div_up(tiles, slots);
            for (int sk = 2; sk <= 64 && sk <= max_split; ++sk) {
                const double cost = (double)asrk_div_up(tiles * sk, slots) / sk * (1.0 + 0.03 * (sk - 1));
                if (cost < best * 0.97) {
                    best = cost;
                    splitk = sk;
                }
                if (tiles * sk >= 8 * slots) break;
            }
        }
    }
    if (kiters == 0) {
        splitk = 1;
        g.k_per_split = BK;
    } else {
        if (splitk > kiters) splitk = kiters;
        g.k_per_split = asrk_div_up(kiters, splitk) * BK;
        splitk = asrk_div_up(K, g.k_per_split);
    }
    g.splitk = splitk;

    if (splitk > 1) {
        // partials are accumulated with atomics on top of beta*C
        if (beta == 0.f) {
            ASRK_HIP(hipMemset2DAsync(C, (size_t)ldc * 4, 0, (size_t)N * 4, M, s));
        } else if (beta != 1.f) {
            const int64_t n = (int64_t)M * N;
            hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)asrk_div_up64(n, 256)),
                               dim3(256), 0, s, C, M, N, ldc, beta);
            ASRK_LAUNCH_CHECK();
        }
        g.beta = 1.f;
    }

    // vector (16-B) global loads need aligned bases, ld % 4 == 0 and the contiguous extent % 4 == 0
    auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec = al16(A) && al16(B) && (lda % 4 == 0) && (ldb % 4 == 0) &&
                     ((a_kc ? K : M) % 4 == 0) && ((b_kc ? K : N) % 4 == 0);

    asrk_prof_begin_(prof_id, s);
    int rc;
    // fast path needs 16-B vector access everywhere plus K >= 4 and (for M/N-contiguous operands)
    // at least 4 rows to clamp into
    const bool no_fast = kn.is_set(kn.gemm_nofast);
    const bool fast = vec && !no_fast && K >= 4 && (a_kc || M >= 4) && (b_kc || N >= 4);
    if (fast) {
        if (a_kc && b_kc) rc = launch_gemm_fast<true, true>(g, lds_hint, s);
        else if (a_kc && !b_kc) rc = launch_gemm_fast<true, false>(g, lds_hint, s);
        else rc = launch_gemm_fast<false, false>(g, lds_hint, s);
        asrk_prof_end_(prof_id, s);
        return rc;
    }
    if (a_kc && b_kc)
        rc = vec ? launch_gemm<true, true, true>(g, s) : launch_gemm<true, true, false>(g, s);
    else if (a_kc && !b_kc)
        rc = vec ? launch_gemm<true, false, true>(g, s) : launch_gemm<true, false, false>(g, s);
    else
        rc = vec ? launch_gemm<false, false, true>(g, s) : launch_gemm<false, false, false>(g, s);
    asrk_prof_end_(prof_id, s);
    return rc;
}

