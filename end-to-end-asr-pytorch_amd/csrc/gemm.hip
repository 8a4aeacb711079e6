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
//   M/N-contiguous [k][132]  : operand stored with M (or N) fastest; fragment = ds_read_b32.
// The MFMA K index is a free permutation as long as A and B agree: inside each group of 8 k's
// MFMA j (0..3) consumes k = 8*quad + 4*(lane>>5) + j from BOTH operands.
// Tile ids are remapped so each XCD (private L2) walks a contiguous run of tiles.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int KC_LD = 36;          // floats per row, K-contiguous image
constexpr int MC_LD = 132;         // floats per k-row, M-contiguous image
constexpr int TILE_FLOATS = 4608;  // max(128*36, 32*132)
constexpr int GEMM_LDS_BYTES = 4 * TILE_FLOATS * 4;

struct GemmArgs {
    const float *A, *B;
    float *C;
    const float *bias, *bias2;
    int M, N, K, lda, ldb, ldc;
    float alpha, beta;
    int splitk, k_per_split, tiles_m, tiles_n;
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
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    if (nk > 0) {
        load_tile<A_KC, VEC>(p.A, p.lda, p.M, m0, kbeg, kend, tid, ra);
        load_tile<B_KC, VEC>(p.B, p.ldb, p.N, n0, kbeg, kend, tid, rb);
        store_tile<A_KC>(smem, tid, ra);
        store_tile<B_KC>(smem + TILE_FLOATS, tid, rb);
    }
    __syncthreads();

    const int l31 = lane & 31, kk = lane >> 5;
    for (int t = 0; t < nk; ++t) {
        const float *As = smem + (t & 1) * 2 * TILE_FLOATS;
        const float *Bs = As + TILE_FLOATS;
        const bool more = (t + 1 < nk);
        if (more) {
            load_tile<A_KC, VEC>(p.A, p.lda, p.M, m0, kbeg + (t + 1) * BK, kend, tid, ra);
            load_tile<B_KC, VEC>(p.B, p.ldb, p.N, n0, kbeg + (t + 1) * BK, kend, tid, rb);
        }
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = read_frag<A_KC>(As, wr * 64 + i * 32 + l31, qd, kk);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = read_frag<B_KC>(Bs, wc * 64 + j * 32 + l31, qd, kk);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s],
                                                                         acc[i][j], 0, 0, 0);
        }
        if (more) {
            float *An = smem + ((t + 1) & 1) * 2 * TILE_FLOATS;
            store_tile<A_KC>(An, tid, ra);
            store_tile<B_KC>(An + TILE_FLOATS, tid, rb);
        }
        __syncthreads();
    }

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

template <bool A_KC, bool B_KC, bool VEC>
int launch_gemm(const GemmArgs &a, hipStream_t s) {
    static bool attr_set = false;
    auto kern = gemm_f32_kernel<A_KC, B_KC, VEC>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           GEMM_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
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
                             int splitk, void *stream) {
    if (M < 0 || N < 0 || K < 0) return ASRK_EINVAL;
    if (M == 0 || N == 0) return ASRK_OK;
    if (!A || !B || !C) return ASRK_EINVAL;
    if (transA && transB) return ASRK_EINVAL;  // TT never occurs on this path
    hipStream_t s = (hipStream_t)stream;
    const bool a_kc = !transA, b_kc = transB != 0;
    if (lda < (a_kc ? K : M) || ldb < (b_kc ? K : N) || ldc < N) return ASRK_EINVAL;

    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.bias2 = bias2;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.alpha = alpha; g.beta = beta;
    g.tiles_m = asrk_div_up(M, BM);
    g.tiles_n = asrk_div_up(N, BN);
    const int tiles = g.tiles_m * g.tiles_n;
    const int kiters = asrk_div_up(K, BK);
    if (splitk <= 0) {
        splitk = 1;
        if (tiles < 192 && kiters >= 16) {
            splitk = asrk_div_up(512, tiles);
            const int max_split = kiters / 8;  // keep >= 8 K tiles (256 k) per split
            if (splitk > max_split) splitk = max_split;
            if (splitk > 64) splitk = 64;
            if (splitk < 1) splitk = 1;
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

    asrk_prof_begin_(PROF_GEMM, s);
    int rc;
    if (a_kc && b_kc)
        rc = vec ? launch_gemm<true, true, true>(g, s) : launch_gemm<true, true, false>(g, s);
    else if (a_kc && !b_kc)
        rc = vec ? launch_gemm<true, false, true>(g, s) : launch_gemm<true, false, false>(g, s);
    else
        rc = vec ? launch_gemm<false, false, true>(g, s) : launch_gemm<false, false, false>(g, s);
    asrk_prof_end_(PROF_GEMM, s);
    return rc;
}
