// 3x3 / stride 1 / pad 1 convolutions of the VGG prenet (src/module.py:21-33: Conv2d(64,64), Conv2d(64,128),
// Conv2d(128,128), kernel 3, padding 1) WITHOUT a patch matrix: implicit GEMMs on the f32-input matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulation - the arithmetic of the im2col + asrk_gemm_f32 path
// they replace, which wrote and re-read M x 9C floats per convolution: 1.2 GB at the shipped conv2).
//
// Activations are contiguous channels-last [B, H(time), W(freq), C], C and Cout multiples of 64, W <= 128.
//
//   forward / data gradient (conv3x3_kernel):  y[pos, n] = bias[n] + sum_{tap, c} x[pos + tap - (1,1), c] * w[tap][c][n]
//     One workgroup = TH full image rows (TH * W <= 128 output positions = the 128 rows of its MFMA tile) x all Cout.
//     It stages the (TH+2) x (W+2) halo of 64 input channels in LDS once (row stride 68 floats: 16 consecutive positions
//     cover the 64 banks exactly once under ds_read_b128) and all nine taps read it at constant offsets; input channels
//     beyond 64 are further passes over a re-staged halo.  The weights come in FRAGMENT order
//     (asrk_conv3x3_weight_f32: [tap][c/8][n/32][lane][4] - one wave-wide 1-KB load per 8 channels x 32 outputs) straight
//     from L2 into registers through a ring of 8 chunks, 4 in flight.  The K index is permuted inside each 8-channel
//     chunk (lane half kk holds channels 4kk..4kk+3 of the chunk, MFMA s contracts channels {s, 4+s}) so that both
//     operands are 16-byte reads.  The data gradient is the same kernel on dy with the weight transposed and flipped;
//     `xmask` (the layer's own ReLU output) zeroes dy where the activation was clamped - no separate relu_bwd pass.
//
//   weight gradient (conv3x3_wgrad_kernel):  dW[co][tap][ci] = sum_pos dy[pos, co] * x[pos + tap - (1,1), ci]
//     One workgroup owns 64 output x 64 input channels x all nine taps (9 x 64 x 64 accumulators = 144 registers per
//     lane) and walks a strided set of position tiles: dy tile and x halo in LDS, positions are the contraction index
//     (two per MFMA).  Partial sums go to the caller's workspace, one slab per workgroup; a second small kernel adds the
//     slabs in a fixed order (deterministic - no atomics) straight into the parameter's [Cout][Cin][3][3] layout, and
//     the bias gradient (column sums of the masked dy tile) rides along.
#include "common.h"
#include <algorithm>

namespace {

constexpr int CH = 64;        // input channels per staged halo
constexpr int CS = CH + 4;    // LDS floats per halo position

__device__ __forceinline__ f32x4 mask4(f32x4 v, bool keep) {
    // bit mask instead of a select: the load feeding a select arm must not be sunk into a branch
    const unsigned m = 0u - (unsigned)keep;
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = __uint_as_float(__float_as_uint(v[e]) & m);
    return r;
}

__device__ __forceinline__ f32x4 relu_gate4(f32x4 v, f32x4 gate) {
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = gate[e] > 0.f ? v[e] : 0.f;
    return r;
}

// halo of rows h0-1 .. h0+TH, columns -1 .. W of channels [c0, c0+64) -> xs[(r*(W+2) + col)*CS + c]
__device__ __forceinline__ void stage_halo(float *xs, const float *__restrict__ x, const float *__restrict__ gate,
                                           int b, int h0, int TH, int H, int W, int C, int c0, int tid) {
    const int W2 = W + 2;
    const int NP = (TH + 2) * W2 * 16;
    constexpr int U = 4;
    for (int p0 = tid; p0 < NP; p0 += 256 * U) {
        f32x4 v[U], g[U];
        int dst[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = min(p0 + u * 256, NP - 1);
            const int c4 = p & 15, pc = p >> 4;
            const int r = pc / W2, col = pc - r * W2;
            const int h = h0 - 1 + r, w = col - 1;
            const bool ok = h >= 0 && h < H && w >= 0 && w < W;
            const int hc = min(max(h, 0), H - 1), wc = min(max(w, 0), W - 1);
            const size_t off = (((size_t)b * H + hc) * W + wc) * C + c0 + c4 * 4;
            v[u] = mask4(*reinterpret_cast<const f32x4 *>(x + off), ok);
            if (gate) g[u] = *reinterpret_cast<const f32x4 *>(gate + off);
            dst[u] = pc * CS + c4 * 4;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (gate) v[u] = relu_gate4(v[u], g[u]);
            if (p0 + u * 256 < NP) *reinterpret_cast<f32x4 *>(xs + dst[u]) = v[u];
        }
    }
}

struct C3Args {
    const float *x;       // [B,H,W,C]
    const float *xmask;   // nullable, shape of x: x counts where xmask > 0
    const float *wf;      // fragment-ordered weight [9][C/8][COUT/32][64][4]
    const float *bias;    // nullable [COUT]
    float *y;             // [B,H,W,COUT]
    int B, H, W, C, tiles_img;    // tiles_img = ceil(H*W / 128)
};

template <int COUT, bool RELU>
__global__ __launch_bounds__(256) void conv3x3_kernel(C3Args p) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    constexpr int NB = COUT / 64;               // 32-column blocks per wave: wave tile 64 rows x COUT/2 columns
    constexpr int NBT = COUT / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, kk = lane >> 5;
    // tile = 128 consecutive positions of one image (NOT whole rows: 16 x 400 x 20 positions are 1008 such tiles - two
    // full rounds of the chip at two workgroups per CU - but 1072 tiles of six rows, i.e. a third round for 5 % of the work)
    const int W = p.W, W2 = W + 2;
    const int b = blockIdx.x / p.tiles_img, p0 = (blockIdx.x - b * p.tiles_img) * 128;
    const int nvalid = min(128, p.H * W - p0);
    const int h0 = p0 / W, w0 = p0 - h0 * W;
    const int TH = (p0 + nvalid - 1) / W - h0 + 1;            // image rows the tile touches

    int aoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = w0 + min(wm * 64 + i * 32 + l31, nvalid - 1);
        const int hl = m / W, w = m - hl * W;
        aoff[i] = (hl * W2 + w) * CS + 4 * kk;
    }
    f32x16 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;

    const int nhalf = p.C / CH;
    const size_t jstride = (size_t)NBT * 64;                    // f32x4 units between channel chunks
    const size_t tstride = (size_t)(p.C / 8) * jstride;         // ... between taps
    for (int half = 0; half < nhalf; ++half) {
        if (half) __syncthreads();
        stage_halo(xs, p.x, p.xmask, b, h0, TH, p.H, W, p.C, half * CH, tid);
        const f32x4 *bp = reinterpret_cast<const f32x4 *>(p.wf) + (size_t)(half * 8) * jstride + (wn * NB) * 64 + lane;
        f32x4 br[8][NB];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < NB; ++n) br[j][n] = bp[j * jstride + n * 64];
        __syncthreads();
        // software pipeline, pinned with scheduling barriers (left alone, the scheduler sinks each weight load to just
        // before its first use and waits for it with vmcnt(0)): chunk q's MFMAs run while chunk q+4's weights and chunk
        // q+1's activations are in flight
        f32x4 ac0 = *reinterpret_cast<const f32x4 *>(xs + aoff[0]), ac1 = *reinterpret_cast<const f32x4 *>(xs + aoff[1]);
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap - kh * 3;
            const int tn = min(tap + 1, 8), khn = tn / 3, kwn = tn - khn * 3;
            const int tapoff = (kh * W2 + kw) * CS, tapoff_n = (khn * W2 + kwn) * CS;
            const f32x4 *bt = bp + tap * tstride;
            const f32x4 *btn = bp + tn * tstride;
            const float *a0p = xs + aoff[0] + tapoff, *a1p = xs + aoff[1] + tapoff;
            const float *a0n = xs + aoff[0] + tapoff_n, *a1n = xs + aoff[1] + tapoff_n;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    br[(j + 4) & 7][n] = (j < 4) ? bt[(j + 4) * jstride + n * 64] : btn[(j - 4) * jstride + n * 64];
                const f32x4 an0 = *reinterpret_cast<const f32x4 *>(j < 7 ? a0p + (j + 1) * 8 : a0n);
                const f32x4 an1 = *reinterpret_cast<const f32x4 *>(j < 7 ? a1p + (j + 1) * 8 : a1n);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int n = 0; n < NB; ++n) {
                        acc[0][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac0[s], br[j][n][s], acc[0][n], 0, 0, 0);
                        acc[1][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac1[s], br[j][n][s], acc[1][n], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
                ac0 = an0;
                ac1 = an1;
            }
        }
    }

    // C/D map of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float *yb = p.y + ((size_t)b * p.H * W + p0) * COUT;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int col = wn * (COUT / 2) + n * 32 + l31;
        const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (m < nvalid) {
                    float v = acc[i][n][r] + bv;
                    if (RELU) v = fmaxf(v, 0.f);
                    yb[(size_t)m * COUT + col] = v;
                }
            }
    }
}

// parameter w[Cout][Cin][3][3] -> fragment order.  transpose == 0: outputs n = cout, contraction c = cin, same taps
// (forward); transpose != 0: n = cin, c = cout, taps flipped (the data gradient's weight).
__global__ __launch_bounds__(256) void conv3x3_weight_kernel(const float *__restrict__ w, float *__restrict__ wf, int Cout,
                                                             int Cin, int transpose) {
    const int N = transpose ? Cin : Cout, Kc = transpose ? Cout : Cin;
    const int total = 9 * N * Kc;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int e = i & 3, lane = (i >> 2) & 63;
        int t = i >> 8;
        const int nb = t % (N / 32);
        t /= (N / 32);
        const int jc = t % (Kc / 8), tap = t / (Kc / 8);
        const int n = nb * 32 + (lane & 31), c = jc * 8 + 4 * (lane >> 5) + e;
        wf[i] = transpose ? w[((size_t)c * Cin + n) * 9 + (8 - tap)] : w[((size_t)n * Cin + c) * 9 + tap];
    }
}

struct W3Args {
    const float *x;       // [B,H,W,C]
    const float *dy;      // [B,H,W,COUT]
    const float *ymask;   // nullable, shape of dy: dy counts where ymask > 0
    float *part;          // [splits][9][64][G][64]
    float *bpart;         // [G][COUT]
    int B, H, W, C, COUT, TH, tiles_h, ntiles, npos2;
};

__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_kernel(W3Args p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cob = wave >> 1, cib = wave & 1, l31 = lane & 31, kk = lane >> 5;
    const int W = p.W, W2 = W + 2, TH = p.TH;
    const int ncih = p.C / CH;
    const int cih = blockIdx.y % ncih, coh = blockIdx.y / ncih;
    float *xs = smem;
    float *dys = xs + (TH + 2) * W2 * CS;
    int *postab = reinterpret_cast<int *>(dys + p.npos2 * CS);
    for (int m = tid; m < p.npos2; m += 256) {
        const int mm = min(m, TH * W - 1), hl = mm / W;
        postab[m] = (hl * W2 + (mm - hl * W)) * CS;
    }
    int tapoff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tapoff[t] = ((t / 3) * W2 + (t % 3)) * CS + cib * 32 + l31;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;

    const int NPD = p.npos2 * 16;
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int b = tile / p.tiles_h, h0 = (tile - b * p.tiles_h) * TH;
        const int nvalid = min(TH, p.H - h0) * W;
        __syncthreads();                         // the previous tile's reads are done (and postab is written)
        stage_halo(xs, p.x, nullptr, b, h0, TH, p.H, W, p.C, cih * CH, tid);
        const size_t pos0 = ((size_t)b * p.H + h0) * W;
        for (int q0 = tid; q0 < NPD; q0 += 1024) {
            f32x4 v[4], g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = min(q0 + u * 256, NPD - 1);
                const int c4 = q & 15, m = q >> 4;
                const size_t off = (pos0 + min(m, nvalid - 1)) * p.COUT + coh * 64 + c4 * 4;
                v[u] = mask4(*reinterpret_cast<const f32x4 *>(p.dy + off), m < nvalid);
                if (p.ymask) g[u] = *reinterpret_cast<const f32x4 *>(p.ymask + off);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = q0 + u * 256;
                if (p.ymask) v[u] = relu_gate4(v[u], g[u]);
                if (q < NPD) *reinterpret_cast<f32x4 *>(dys + (q >> 4) * CS + (q & 15) * 4) = v[u];
            }
        }
        __syncthreads();
        if (cih == 0 && tid < 64) {
            float s0 = 0.f, s1 = 0.f;
            for (int m = 0; m < p.npos2; m += 2) {
                s0 += dys[m * CS + tid];
                s1 += dys[(m + 1) * CS + tid];
            }
            bsum += s0 + s1;
        }
        // positions two at a time; step ks+2's operands (and step ks+4's halo offset) are read while step ks multiplies
        const float *ap = dys + kk * CS + cob * 32 + l31;
        const int last = p.npos2 - 2;
        float a_c = ap[0], b_c[9];
        {
            const float *bq = xs + postab[kk];
#pragma unroll
            for (int t = 0; t < 9; ++t) b_c[t] = bq[tapoff[t]];
        }
        int po_n = postab[min(2, last) + kk];
        for (int ks = 0; ks < p.npos2; ks += 2) {
            const int kn = min(ks + 2, last);
            const float a_n = ap[kn * CS];
            const float *bq = xs + po_n;
            float b_n[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) b_n[t] = bq[tapoff[t]];
            po_n = postab[min(ks + 4, last) + kk];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c, b_c[t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a_c = a_n;
#pragma unroll
            for (int t = 0; t < 9; ++t) b_c[t] = b_n[t];
        }
    }

    // part[split][tap][co][g][ci]: the G x 64 floats that add up to one row of dW[., tap, .] are contiguous, so the
    // reduction streams (slabs [g][split][tap][co][ci] made it read 256-byte pieces 147-590 KB apart: 0.6 TB/s)
    const size_t G = gridDim.x;
    float *slab = p.part + (size_t)blockIdx.y * 9 * 64 * G * 64 + (size_t)blockIdx.x * 64;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cob * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            slab[((size_t)t * 64 + co) * G * 64 + cib * 32 + l31] = acc[t][r];
        }
    if (cih == 0 && tid < 64) p.bpart[(size_t)blockIdx.x * p.COUT + coh * 64 + tid] = bsum;
}

// sum over g of rows[row][g][NL] (NL = 64 or 32 lanes wide), 256 threads = NL lanes x 256/NL interleaved g groups with 16
// loads in flight each (one thread per output walked G = 512 slabs in 32 dependent rounds: 124 us for 75 MB); the groups'
// sums meet in LDS and are added in a fixed order
template <int NL>
__device__ __forceinline__ float slab_sum(const float *__restrict__ row, int G, float *red) {
    constexpr int NG = 256 / NL;
    const int tid = threadIdx.x, l = tid % NL, gl = tid / NL;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int g = gl;
    for (; g + 15 * NG < G; g += 16 * NG) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = row[(size_t)(g + u * NG) * NL + l];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u & 3] += v[u];
    }
    for (; g < G; g += NG) acc[0] += row[(size_t)g * NL + l];
    red[tid] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    float s = 0.f;
    if (tid < NL) {
#pragma unroll
        for (int q = 0; q < NG; ++q) s += red[q * NL + tid];
    }
    return s;
}

// sum over g = gl, gl+4, ... of bpart[g][co], 8 loads in flight (one dependent load per step made this the longest
// block of the reduction: 256 steps at G = 1024)
__device__ __forceinline__ float bias_partial(const float *__restrict__ bpart, int G, int Cout, int co, int gl) {
    float a0 = 0.f, a1 = 0.f;
    int g = gl;
    for (; g + 28 < G; g += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = bpart[(size_t)(g + 4 * u) * Cout + co];
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            a0 += v[u];
            a1 += v[u + 1];
        }
    }
    for (; g < G; g += 4) a0 += bpart[(size_t)g * Cout + co];
    return a0 + a1;
}

// dw[co][ci][tap] = sum_g part[split(co/64, ci/64)][tap][co%64][g][ci%64];  db[co] = sum_g bpart[g][co]
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_kernel(const float *__restrict__ part,
                                                                   const float *__restrict__ bpart, float *__restrict__ dw,
                                                                   float *__restrict__ db, int G, int Cout, int Cin) {
    __shared__ float red[256];
    const int ncih = Cin / 64, nrows = ncih * (Cout / 64) * 9 * 64;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < nrows) {
        if (!dw) return;
        const int row = blockIdx.x, col = row & 63, t = (row >> 6) % 9, split = row / (9 * 64);
        const float s = slab_sum<64>(part + (size_t)row * G * 64, G, red);
        const int co = (split / ncih) * 64 + col, ci = (split % ncih) * 64 + tid;
        if (tid < 64) dw[((size_t)co * Cin + ci) * 9 + t] = s;
    } else if (db) {
        // bpart[g][Cout]: 64 channels per block, 4 interleaved g groups
        const int co = (blockIdx.x - nrows) * 64 + (tid & 63), gl = tid >> 6;
        red[tid] = bias_partial(bpart, G, Cout, co, gl);
        __syncthreads();
        if (tid < 64) db[co] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
    }
}

// ---- the FIRST layer (Conv2d(in_channel, 64, 3, padding=1), in_channel = 1..3 delta-feature planes, src/module.py:21 and
// view_input, 44-57): K = 9 * Cin <= 27, so the whole contraction is one 32-deep MFMA tile; the input is the [B,T,C*F]
// feature tensor read in place through element strides.  Both kernels are HBM-bound on the 64-channel activation
// (write y forward; read dy and the ReLU output backward), which the im2col path crossed five more times
// (patches written and read, relu passes, a column-sum pass, a 128 x 128-tile GEMM for a 64 x 27 result).
struct F1Args {
    const float *x;       // addressed x[b*sb + h*sh + w*sw + c*sc]
    const float *w;       // parameter layout [Cout][Cin*9]
    const float *bias;    // nullable
    float *y;             // [B,H,W,Cout]
    const float *dy;      // wgrad: [B,H,W,Cout]
    const float *ymask;   // wgrad: nullable
    float *part;          // wgrad: [Cout/64][64 co][G][32 k]
    float *bpart;         // wgrad: [G][Cout]
    int B, H, W, C, Cout, TH, tiles_h, ntiles;
    int64_t sb, sh, sw, sc;
};

// xs[c][(TH+2)][(W+2)]: the halo of every input plane
__device__ __forceinline__ void stage_planes(float *xs, const F1Args &p, int b, int h0, int tid) {
    const int W2 = p.W + 2, R = p.TH + 2;
    const int NP = p.C * R * W2;
    for (int q = tid; q < NP; q += 256) {
        const int c = q / (R * W2), rem = q - c * (R * W2);
        const int r = rem / W2, col = rem - r * W2;
        const int h = h0 - 1 + r, w = col - 1;
        const bool ok = h >= 0 && h < p.H && w >= 0 && w < p.W;
        const int hc = min(max(h, 0), p.H - 1), wc = min(max(w, 0), p.W - 1);
        const float v = p.x[b * p.sb + hc * p.sh + wc * p.sw + c * p.sc];
        xs[q] = __uint_as_float(__float_as_uint(v) & (0u - (unsigned)ok));
    }
}

template <bool RELU>
__global__ __launch_bounds__(256) void conv3x3_first_kernel(F1Args p) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, kk = lane >> 5;
    const int W = p.W, W2 = W + 2, TH = p.TH, K = 9 * p.C;
    const int b = blockIdx.x / p.tiles_h, h0 = (blockIdx.x - b * p.tiles_h) * TH;
    const int nvalid = min(TH, p.H - h0) * W;
    const int col = blockIdx.y * 64 + wn * 32 + l31;

    // B fragments: w[col][k], k = 2s + kk (zero beyond K); the matching LDS offsets of A's k
    float bw[16];
    int koff[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int k = 2 * s + kk, kc = min(k, K - 1);
        const float v = p.w[(size_t)col * K + kc];
        bw[s] = k < K ? v : 0.f;
        const int c = kc / 9, t = kc - c * 9;
        koff[s] = (c * (TH + 2) + t / 3) * W2 + (t % 3);
    }
    stage_planes(xs, p, b, h0, tid);
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const int m = min(wm * 64 + i * 32 + l31, nvalid - 1);
        const int hl = m / W;
        const float *ap = xs + hl * W2 + (m - hl * W);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float a = (2 * s + kk < K) ? ap[koff[s]] : 0.f;
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw[s], acc[i], 0, 0, 0);
        }
    }
    float *yb = p.y + ((size_t)b * p.H + h0) * W * p.Cout;
    const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            if (m < nvalid) {
                float v = acc[i][r] + bv;
                if (RELU) v = fmaxf(v, 0.f);
                yb[(size_t)m * p.Cout + col] = v;
            }
        }
}

// dW[co][k] = sum_pos dy[pos][co] * patch[pos][k]: rows = 64 output channels (two MFMA blocks), columns = k (one block),
// positions are the contraction; wave w contracts positions 32w .. 32w+31 of every 128-position tile
__global__ __launch_bounds__(256) void conv3x3_first_wgrad_kernel(F1Args p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kk = lane >> 5;
    const int W = p.W, W2 = W + 2, TH = p.TH, K = 9 * p.C;
    const int coh = blockIdx.y;
    float *dys = smem;                                        // [128][CS]
    float *xs = dys + 128 * CS;                               // [C][TH+2][W2]
    int *postab = reinterpret_cast<int *>(xs + p.C * (TH + 2) * W2);
    for (int m = tid; m < 128; m += 256) {
        const int mm = min(m, TH * W - 1), hl = mm / W;
        postab[m] = hl * W2 + (mm - hl * W);
    }
    const int kc = min(l31, K - 1), kci = kc / 9, kt = kc - kci * 9;
    const int koff = (kci * (TH + 2) + kt / 3) * W2 + (kt % 3);
    const bool kok = l31 < K;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float bsum = 0.f;

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int b = tile / p.tiles_h, h0 = (tile - b * p.tiles_h) * TH;
        const int nvalid = min(TH, p.H - h0) * W;
        __syncthreads();
        stage_planes(xs, p, b, h0, tid);
        const size_t pos0 = ((size_t)b * p.H + h0) * W;
#pragma unroll
        for (int half = 0; half < 2; ++half) {                // 128 positions x 16 pieces = 2048 = 2 x 4 x 256
            f32x4 v[4], g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = tid + (half * 4 + u) * 256;
                const int c4 = q & 15, m = q >> 4;
                const size_t off = (pos0 + min(m, nvalid - 1)) * p.Cout + coh * 64 + c4 * 4;
                v[u] = mask4(*reinterpret_cast<const f32x4 *>(p.dy + off), m < nvalid);
                if (p.ymask) g[u] = *reinterpret_cast<const f32x4 *>(p.ymask + off);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = tid + (half * 4 + u) * 256;
                if (p.ymask) v[u] = relu_gate4(v[u], g[u]);
                *reinterpret_cast<f32x4 *>(dys + (q >> 4) * CS + (q & 15) * 4) = v[u];
            }
        }
        __syncthreads();
        if (tid < 64) {
            float s0 = 0.f, s1 = 0.f;
            for (int m = 0; m < 128; m += 2) {
                s0 += dys[m * CS + tid];
                s1 += dys[(m + 1) * CS + tid];
            }
            bsum += s0 + s1;
        }
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const int pos = wave * 32 + 2 * s + kk;
            const float bv = xs[postab[pos] + koff];
            const float bq = kok ? bv : 0.f;
            const float *ap = dys + pos * CS + l31;
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[0], bq, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[32], bq, acc[1], 0, 0, 0);
        }
    }
    // the four waves' partial sums -> one slab (fixed order)
    __syncthreads();
    float *red = smem;                                        // [4][64][32] over the dy tile
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            red[(wave * 64 + co) * 32 + l31] = acc[i][r];
        }
    __syncthreads();
    const size_t G = gridDim.x;
    for (int e = tid; e < 64 * 32; e += 256) {
        const float v = (red[e] + red[2048 + e]) + (red[4096 + e] + red[6144 + e]);
        p.part[(((size_t)coh * 64 + (e >> 5)) * G + blockIdx.x) * 32 + (e & 31)] = v;
    }
    if (tid < 64) p.bpart[(size_t)blockIdx.x * p.Cout + coh * 64 + tid] = bsum;
}

// dw[co][k] = sum_g part[co][g][k];  db[co] = sum_g bpart[g][co]
__global__ __launch_bounds__(256) void conv3x3_first_reduce_kernel(const float *__restrict__ part,
                                                                   const float *__restrict__ bpart, float *__restrict__ dw,
                                                                   float *__restrict__ db, int G, int Cout, int K) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < Cout) {
        if (!dw) return;
        const int co = blockIdx.x;
        const float s = slab_sum<32>(part + (size_t)co * G * 32, G, red);
        if (tid < K) dw[(size_t)co * K + tid] = s;
    } else if (db) {
        const int co = (blockIdx.x - Cout) * 64 + (tid & 63), gl = tid >> 6;
        red[tid] = bias_partial(bpart, G, Cout, co, gl);
        __syncthreads();
        if (tid < 64) db[co] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
    }
}

struct Plan3 {
    int TH, tiles_h, lds;
};

inline bool dims_ok(int B, int H, int W, int C, int Cout) {
    return B >= 0 && H > 0 && W > 0 && W <= 128 && C > 0 && C % 64 == 0 && Cout > 0 && Cout % 64 == 0 &&
           (int64_t)B * H * W * std::max(C, Cout) < ((int64_t)1 << 40) && (int64_t)B * H * W < (1 << 30);
}

inline Plan3 fwd_plan(int H, int W) {
    Plan3 q;
    int gcd = 128, r = W;                         // tiles start at multiples of 128: first columns are multiples of gcd(128, W)
    while (r) {
        const int t = gcd % r;
        gcd = r;
        r = t;
    }
    q.TH = (W - gcd + 127) / W + 1;               // most image rows one tile can touch
    q.tiles_h = asrk_div_up(H * W, 128);          // tiles per image
    q.lds = (std::min(q.TH, H) + 2) * (W + 2) * CS * 4;
    return q;
}

inline int wgrad_lds(int TH, int W) {
    const int npos2 = (TH * W + 1) & ~1;
    return ((TH + 2) * (W + 2) + npos2) * CS * 4 + npos2 * 4;
}

inline int wgrad_groups(int ntiles, int splits) { return std::max(1, std::min(ntiles, 512 / splits)); }

// rows per tile: the largest that keeps two workgroups per CU is not always the fastest - a workgroup walks
// ceil(tiles / groups) tiles, so 1072 tiles over 256 workgroups cost five tiles' time for 4.2 tiles' work.  Pick the TH
// that minimises (tiles per workgroup) x (cost of a tile: its MFMAs plus the staging of its halo).
inline Plan3 wgrad_plan(int B, int H, int W, int splits) {
    int thmax = 1;
    while (thmax < H && wgrad_lds(thmax + 1, W) <= 80 * 1024) ++thmax;
    Plan3 best{thmax, asrk_div_up(H, thmax), wgrad_lds(thmax, W)};
    double best_cost = 1e300;
    for (int th = thmax; th >= std::max(1, thmax / 2); --th) {
        const int tiles_h = asrk_div_up(H, th), ntiles = std::max(1, B) * tiles_h;
        const int G = wgrad_groups(ntiles, splits);
        const double cost = (double)asrk_div_up(ntiles, G) * ((double)th * W + 0.3 * (th + 2) * (W + 2) + 12.0);
        if (cost < best_cost) {
            best_cost = cost;
            best = Plan3{th, tiles_h, wgrad_lds(th, W)};
        }
    }
    return best;
}

inline bool al16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

template <int COUT, bool RELU>
int launch_conv(const C3Args &a, int lds, hipStream_t s) {
    static AsrkLdsLatch latch;
    auto kern = conv3x3_kernel<COUT, RELU>;
    ASRK_HIP(asrk_max_lds_once(latch, reinterpret_cast<const void *>(kern), 158 * 1024));
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * a.tiles_img)), dim3(256), lds, s, a);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

}  // namespace

extern "C" int asrk_conv3x3_supported(int H, int W, int C, int Cout) {
    if (!dims_ok(1, H, W, C, Cout) || (Cout != 64 && Cout != 128) || (C != 64 && C != 128)) return 0;
    return fwd_plan(H, W).lds <= 158 * 1024 && wgrad_plan(1, H, W, 1).lds <= 158 * 1024;
}

extern "C" int asrk_conv3x3_weight_f32(const float *w, float *wf, int Cout, int Cin, int transpose, void *stream) {
    if (Cout <= 0 || Cin <= 0 || Cout % 64 || Cin % 64 || !w || !wf || w == wf) return ASRK_EINVAL;
    if ((int64_t)Cout * Cin * 9 >= (int64_t)1 << 31) return ASRK_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_CONV, s);
    hipLaunchKernelGGL(conv3x3_weight_kernel, dim3((unsigned)asrk_div_up(9 * Cout * Cin, 256)), dim3(256), 0, s, w, wf, Cout,
                       Cin, transpose);
    asrk_prof_end_(PROF_CONV, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_conv3x3_f32(const float *x, const float *xmask, const float *wf, const float *bias, float *y, int B,
                                int H, int W, int C, int Cout, int relu, void *stream) {
    if (B < 0 || H <= 0 || W <= 0 || C <= 0 || Cout <= 0) return ASRK_EINVAL;
    if (B == 0) return ASRK_OK;
    if (!x || !wf || !y) return ASRK_EINVAL;
    if (!dims_ok(B, H, W, C, Cout)) return ASRK_ESHAPE;
    if (!asrk_conv3x3_supported(H, W, C, Cout) || !al16(x) || !al16(wf) || (xmask && !al16(xmask))) return ASRK_ESHAPE;
    const Plan3 q = fwd_plan(H, W);
    C3Args a{x, xmask, wf, bias, y, B, H, W, C, q.tiles_h};
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_work_(PROF_CONV_MFMA, 2.0 * (double)B * H * W * 9.0 * C * Cout);
    asrk_prof_begin_(PROF_CONV_MFMA, s);
    int rc;
    if (Cout == 64) rc = relu ? launch_conv<64, true>(a, q.lds, s) : launch_conv<64, false>(a, q.lds, s);
    else rc = relu ? launch_conv<128, true>(a, q.lds, s) : launch_conv<128, false>(a, q.lds, s);
    asrk_prof_end_(PROF_CONV_MFMA, s);
    return rc;
}

extern "C" size_t asrk_conv3x3_wgrad_ws_bytes(int B, int H, int W, int C, int Cout) {
    if (!dims_ok(B, H, W, C, Cout) || B == 0) return 0;
    const int splits = (C / 64) * (Cout / 64);
    const Plan3 q = wgrad_plan(B, H, W, splits);
    const int G = wgrad_groups(B * q.tiles_h, splits);
    return ((size_t)G * splits * 9 * 4096 + (size_t)G * Cout) * sizeof(float);
}

extern "C" int asrk_conv3x3_wgrad_f32(const float *x, const float *dy, const float *ymask, float *dw, float *db, int B,
                                      int H, int W, int C, int Cout, void *ws, size_t ws_bytes, void *stream) {
    if (B < 0 || H <= 0 || W <= 0 || C <= 0 || Cout <= 0) return ASRK_EINVAL;
    if (!dw && !db) return ASRK_OK;
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) {
        if (dw) ASRK_HIP(hipMemsetAsync(dw, 0, (size_t)Cout * C * 9 * 4, s));
        if (db) ASRK_HIP(hipMemsetAsync(db, 0, (size_t)Cout * 4, s));
        return ASRK_OK;
    }
    if (!x || !dy) return ASRK_EINVAL;
    if (!dims_ok(B, H, W, C, Cout)) return ASRK_ESHAPE;
    if (!asrk_conv3x3_supported(H, W, C, Cout) || !al16(x) || !al16(dy) || (ymask && !al16(ymask))) return ASRK_ESHAPE;
    if (!ws || ws_bytes < asrk_conv3x3_wgrad_ws_bytes(B, H, W, C, Cout)) return ASRK_EWORKSPACE;
    if (!al16(ws)) return ASRK_EINVAL;
    const int splits = (C / 64) * (Cout / 64);
    const Plan3 q = wgrad_plan(B, H, W, splits);
    const int ntiles = B * q.tiles_h;
    const int G = wgrad_groups(ntiles, splits);
    float *part = reinterpret_cast<float *>(ws);
    float *bpart = part + (size_t)G * splits * 9 * 4096;
    W3Args a{x, dy, ymask, part, bpart, B, H, W, C, Cout, q.TH, q.tiles_h, ntiles, (q.TH * W + 1) & ~1};
    static AsrkLdsLatch latch;
    ASRK_HIP(asrk_max_lds_once(latch, reinterpret_cast<const void *>(conv3x3_wgrad_kernel), 158 * 1024));
    asrk_prof_work_(PROF_CONV_MFMA, 2.0 * (double)B * H * W * 9.0 * C * Cout);
    asrk_prof_begin_(PROF_CONV_MFMA, s);
    hipLaunchKernelGGL(conv3x3_wgrad_kernel, dim3((unsigned)G, (unsigned)splits), dim3(256), q.lds, s, a);
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3((unsigned)(splits * 9 * 64 + Cout / 64)), dim3(256), 0, s, part,
                       bpart, dw, db, G, Cout, C);
    asrk_prof_end_(PROF_CONV_MFMA, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

// ---- first layer: Cin <= 3 planes addressed through element strides, Cout a multiple of 64
namespace {
inline bool first_ok(int B, int H, int W, int C, int Cout) {
    return B >= 0 && H > 0 && W > 0 && W <= 128 && C >= 1 && C <= 3 && Cout > 0 && Cout % 64 == 0 && Cout <= 65535 * 64 &&
           (int64_t)B * H < (1 << 30);
}
inline Plan3 first_plan(int H, int W, int C, bool wgrad) {
    Plan3 q;
    q.TH = std::max(1, std::min(H, 128 / W));
    q.tiles_h = asrk_div_up(H, q.TH);
    q.lds = C * (q.TH + 2) * (W + 2) * 4 + (wgrad ? 128 * CS * 4 + 128 * 4 : 0);
    return q;
}
}  // namespace

extern "C" int asrk_conv3x3_first_supported(int H, int W, int C, int Cout) { return first_ok(1, H, W, C, Cout) ? 1 : 0; }

extern "C" int asrk_conv3x3_first_f32(const float *x, const float *w, const float *bias, float *y, int B, int H, int W,
                                      int C, int Cout, int64_t sb, int64_t sh, int64_t sw, int64_t sc, int relu,
                                      void *stream) {
    if (B < 0 || H <= 0 || W <= 0 || C <= 0 || Cout <= 0) return ASRK_EINVAL;
    if (B == 0) return ASRK_OK;
    if (!x || !w || !y) return ASRK_EINVAL;
    if (!first_ok(B, H, W, C, Cout)) return ASRK_ESHAPE;
    const Plan3 q = first_plan(H, W, C, false);
    F1Args a{x, w, bias, y, nullptr, nullptr, nullptr, nullptr, B, H, W, C, Cout, q.TH, q.tiles_h, B * q.tiles_h, sb, sh, sw, sc};
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_CONV, s);
    const dim3 grid((unsigned)(B * q.tiles_h), (unsigned)(Cout / 64));
    if (relu) hipLaunchKernelGGL(conv3x3_first_kernel<true>, grid, dim3(256), q.lds, s, a);
    else hipLaunchKernelGGL(conv3x3_first_kernel<false>, grid, dim3(256), q.lds, s, a);
    asrk_prof_end_(PROF_CONV, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" size_t asrk_conv3x3_first_wgrad_ws_bytes(int B, int H, int W, int C, int Cout) {
    if (!first_ok(B, H, W, C, Cout) || B == 0) return 0;
    const Plan3 q = first_plan(H, W, C, true);
    const int G = std::max(1, std::min(B * q.tiles_h, 1024));
    return ((size_t)Cout * G * 32 + (size_t)G * Cout) * sizeof(float);
}

extern "C" int asrk_conv3x3_first_wgrad_f32(const float *x, const float *dy, const float *ymask, float *dw, float *db, int B,
                                            int H, int W, int C, int Cout, int64_t sb, int64_t sh, int64_t sw, int64_t sc,
                                            void *ws, size_t ws_bytes, void *stream) {
    if (B < 0 || H <= 0 || W <= 0 || C <= 0 || Cout <= 0) return ASRK_EINVAL;
    if (!dw && !db) return ASRK_OK;
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) {
        if (dw) ASRK_HIP(hipMemsetAsync(dw, 0, (size_t)Cout * C * 9 * 4, s));
        if (db) ASRK_HIP(hipMemsetAsync(db, 0, (size_t)Cout * 4, s));
        return ASRK_OK;
    }
    if (!x || !dy) return ASRK_EINVAL;
    if (!first_ok(B, H, W, C, Cout) || !al16(dy) || (ymask && !al16(ymask))) return ASRK_ESHAPE;
    if (!ws || ws_bytes < asrk_conv3x3_first_wgrad_ws_bytes(B, H, W, C, Cout)) return ASRK_EWORKSPACE;
    if (!al16(ws)) return ASRK_EINVAL;
    const Plan3 q = first_plan(H, W, C, true);
    const int ntiles = B * q.tiles_h;
    const int G = std::max(1, std::min(ntiles, 1024));
    float *part = reinterpret_cast<float *>(ws);
    float *bpart = part + (size_t)Cout * G * 32;
    F1Args a{x, nullptr, nullptr, nullptr, dy, ymask, part, bpart, B, H, W, C, Cout, q.TH, q.tiles_h, ntiles, sb, sh, sw, sc};
    asrk_prof_begin_(PROF_CONV, s);
    hipLaunchKernelGGL(conv3x3_first_wgrad_kernel, dim3((unsigned)G, (unsigned)(Cout / 64)), dim3(256), q.lds, s, a);
    hipLaunchKernelGGL(conv3x3_first_reduce_kernel, dim3((unsigned)(Cout + Cout / 64)), dim3(256), 0, s, part, bpart, dw, db,
                       G, Cout, 9 * C);
    asrk_prof_end_(PROF_CONV, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}
