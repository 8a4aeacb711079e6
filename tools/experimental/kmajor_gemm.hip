// EXPERIMENT (not part of libasrk, not on any product path): the split GEMM with K-MAJOR operands.
//   C[M][N] = sum_k A[k][m] * B[k][n]      A stored [Kd][M], B stored [Kd][N] (f32, row-major)
// i.e. the weight-gradient contraction dW = dG^T X read straight from the ROW-major split panels of dG and X
// (pieces [64 k-rows][8 columns] bf16, three planes) instead of from transposed panels that cost a second split
// pass.  A k-tile is 32 k-rows; per operand the LDS image is eight [32 k][16 m] blocks per
// plane (32 B per k-row: two 8-column pieces side by side), filled by LDS-DMA with lane -> (k-row lane >> 1, piece
// lane & 1), read with ds_read_b64_tr_b16 (tools/tr_probe.hip: lane i of a 16-lane group gets column i of the 4 x 16
// tile its group addresses): two reads give the 8 consecutive k of the 32x32x16 MFMA fragment.  128 B of padding per
// 16-column group makes a 32-lane pass hit 64 distinct banks.
// -DVARIANT=1: the two pieces of a block are loaded by the two HALVES of the wave (lanes 0-31: piece 0, lanes 32-63:
// piece 1 with its rows stored at slot row ^ 4) instead of alternating lanes, so that every quad of lanes reads 64
// contiguous bytes; LDS block = two [32 k][8 m] sub-blocks, the row swizzle keeps a 16-lane group on 32 distinct banks.
// build: hipcc --offload-arch=gfx950 -O3 [-DVARIANT=1] tools/experimental/kmajor_gemm.hip -o tools/experimental/kmajor_gemm
// run:   tools/experimental/kmajor_gemm            (numerics on a small shape, then timing on 8192 x 4096 x 25600)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#ifndef VARIANT
#define VARIANT 0
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int PIECE = 1024, NPL = 3, NST = 3;
constexpr int GROUP = NPL * 1024 + 128;          // one 16-column group: 3 plane blocks + bank-rotating pad
constexpr int REGION = 8 * GROUP;                // one operand's k-tile: 128 columns
constexpr int STAGE = 2 * REGION;                // A + B
constexpr int LDS_BYTES = NST * STAGE;           // 153 600 B

struct Args {
    const unsigned char *Ap, *Bp;                // row-major split panels: rows = k, "K" = columns (m / n)
    size_t rbs_a, rbs_b;                         // row-block strides
    float *C;
    int M, N, ldc, nk, tiles_m, tiles_n;
};

__device__ __forceinline__ void glds16(const unsigned char *g, unsigned char *l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    __builtin_amdgcn_s_waitcnt((N & 15) | 0x70 | 0xF00 | (((N >> 4) & 3) << 14));
}
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(0xC07F); }
__device__ __forceinline__ s16x4 tr_read(const unsigned char *p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p);
}

// 512 threads: waves 0..3 multiply (2 x 2 of 64 x 64), waves 4..7 issue the LDS-DMA (12 blocks each per k-tile)
__global__ __launch_bounds__(512) void gemm_km_kernel(Args p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = p.tiles_m * p.tiles_n, bid = blockIdx.x;
    const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = bid & 7, loc = bid >> 3;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    constexpr int BAND = 8;
    const int band = tile / (BAND * p.tiles_m);
    const int band_w = min(BAND, p.tiles_n - band * BAND);
    const int in_band = tile - band * BAND * p.tiles_m;
    const int tm = in_band / band_w, tn = band * BAND + in_band % band_w;
    const int nk = p.nk;
    const bool worker = wave < 4;

    if (!worker) {
        // DMA wave d: d < 2 -> A groups 4d .. 4d+3, else B groups 4(d-2) .. ; lane -> (k-row lane >> 1, piece lane & 1)
        const int d = wave - 4;
        const bool isA = d < 2;
        const int j0 = (d & 1) * 4;
        const unsigned char *panel = isA ? p.Ap : p.Bp;
        const size_t rbs = isA ? p.rbs_a : p.rbs_b;
        const int col0 = (isA ? tm : tn) * 128;                       // first column of the tile
#if VARIANT
        const int pc = lane >> 5, prow = (lane & 31) ^ (pc << 2);
        const unsigned char *gbase = panel + (size_t)((col0 >> 3) + pc) * NPL * PIECE + prow * 16;
#else
        const unsigned char *gbase = panel + (size_t)((col0 >> 3) + (lane & 1)) * NPL * PIECE + (lane >> 1) * 16;
#endif
        unsigned char *lbase = lds + (isA ? 0 : REGION) + j0 * GROUP;
        auto issue = [&](int kt, int stage) {
            const unsigned char *g = gbase + (size_t)(kt >> 1) * rbs + (kt & 1) * 512;
            unsigned char *l = lbase + stage * STAGE;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    glds16(g + (size_t)((j0 + j) * 2 * NPL + pl) * PIECE, l + j * GROUP + pl * PIECE);
        };
        constexpr int LPT = 12;
#pragma unroll
        for (int s = 0; s < NST; ++s)
            if (s < nk) issue(s, s);
        const int later0 = min(NST - 1, nk - 1);
        if (later0 == 2) wait_vm<2 * LPT>();
        else if (later0 == 1) wait_vm<LPT>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        int stage = 0;
        for (int kt = 0; kt + 1 < nk; ++kt) {
            const int later = min(NST - 2, nk - 2 - kt);
            if (later == 1) wait_vm<LPT>();
            else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            if (kt + NST < nk) issue(kt + NST, stage);
            if (++stage == NST) stage = 0;
        }
        return;
    }

    const int wr = wave >> 1, wc = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment address of row tile i, plane pl, 16-k step ks, read q (k + 4q):
    //   group (w*4 + 2i + g16), k-row 16 ks + 8 h + 4 q + (s >> 2), quad s & 3
    const int g16 = (lane >> 4) & 1, h = lane >> 5, s = lane & 15;
#if VARIANT
    const int sub = (s >> 1) & 1;                 // quad s & 3 -> sub-block (quad >> 1), 8-byte half (quad & 1)
    const int frag = g16 * GROUP + sub * 512 + (8 * h + (s >> 2)) * 16 + (s & 1) * 8;
    const int off0 = sub ? 64 : 0, off1 = sub ? 0 : 64;          // row ^ 4 for the second piece
    constexpr int KSTEP = 16 * 16;
#else
    const int frag = g16 * GROUP + (8 * h + (s >> 2)) * 32 + (s & 3) * 8;
    const int off0 = 0, off1 = 128;
    constexpr int KSTEP = 16 * 32;
#endif
    const unsigned char *abase = lds + wr * 4 * GROUP + frag;
    const unsigned char *bbase = lds + REGION + wc * 4 * GROUP + frag;

    bf16x8 fa[2][2][NPL], fb[2][2][NPL];
    auto load_frags = [&](int buf, int stage, int ks) {
        const unsigned char *a_st = abase + stage * STAGE + ks * KSTEP;
        const unsigned char *b_st = bbase + stage * STAGE + ks * KSTEP;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                const s16x4 a0 = tr_read(a_st + i * 2 * GROUP + pl * PIECE + off0), a1 = tr_read(a_st + i * 2 * GROUP + pl * PIECE + off1);
                const s16x4 b0 = tr_read(b_st + i * 2 * GROUP + pl * PIECE + off0), b1 = tr_read(b_st + i * 2 * GROUP + pl * PIECE + off1);
                fa[buf][i][pl] = __builtin_bit_cast(bf16x8, s16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]});
                fb[buf][i][pl] = __builtin_bit_cast(bf16x8, s16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]});
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mfmas = [&](int buf) {
#define TERM(PA, PB)                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)            \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][i][PA], fb[buf][j][PB], acc[i][j], 0, 0, 0);
        TERM(2, 0) TERM(1, 1) TERM(0, 2) TERM(1, 0) TERM(0, 1) TERM(0, 0)
#undef TERM
        __builtin_amdgcn_sched_barrier(0);
    };

    __builtin_amdgcn_s_barrier();
    load_frags(0, 0, 0);
    int stage = 0;
    for (int kt = 0; kt + 1 < nk; ++kt) {
        wait_lgkm0();
        load_frags(1, stage, 1);
        mfmas(0);
        int nstage = stage + 1;
        if (nstage == NST) nstage = 0;
        wait_lgkm0();
        __builtin_amdgcn_s_barrier();
        load_frags(0, nstage, 0);
        mfmas(1);
        stage = nstage;
    }
    wait_lgkm0();
    load_frags(1, stage, 1);
    mfmas(0);
    wait_lgkm0();
    mfmas(1);

    const int row0 = tm * 128 + wr * 64 + 4 * (lane >> 5), col0 = tn * 128 + wc * 64 + (lane & 31);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = col0 + j * 32;
        if (col >= p.N) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                if (row < p.M) p.C[(size_t)row * p.ldc + col] = acc[i][j][r];
            }
    }
}

// simple (unoptimised) row-major split: src [rows][cols] f32 -> pieces [64 rows][8 cols] x 3 planes
__global__ void split_rowmajor_kernel(const float *src, int rows, int cols, unsigned char *dst, size_t rbs, int rows_pad,
                                      int KC) {
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(id % KC);
    const int row = (int)(id / KC);
    if (row >= rows_pad) return;
    unsigned short hp[3][8];
    for (int e = 0; e < 8; ++e) {
        const int col = c * 8 + e;
        const float a = (row < rows && col < cols) ? src[(size_t)row * cols + col] : 0.f;
        const __bf16 b0 = (__bf16)a;
        const float r1 = a - (float)b0;
        const __bf16 b1 = (__bf16)r1;
        const float r2 = r1 - (float)b1;
        const __bf16 b2 = (__bf16)r2;
        hp[0][e] = __builtin_bit_cast(unsigned short, b0);
        hp[1][e] = __builtin_bit_cast(unsigned short, b1);
        hp[2][e] = __builtin_bit_cast(unsigned short, b2);
    }
    unsigned char *d = dst + (size_t)(row >> 6) * rbs + (size_t)c * NPL * PIECE + (row & 63) * 16;
    for (int pl = 0; pl < 3; ++pl) {
        u32x4 w;
        for (int q = 0; q < 4; ++q) w[q] = hp[pl][2 * q] | ((unsigned)hp[pl][2 * q + 1] << 16);
        *reinterpret_cast<u32x4 *>(d + pl * PIECE) = w;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Panel { unsigned char *p; size_t rbs; int rows_pad, KC; };

static int make_panel(const float *dsrc, int rows, int cols, Panel &pn) {
    pn.rows_pad = (rows + 127) / 128 * 128 + 64;          // slack: the last k-tile may start a fresh row block
    pn.KC = (cols + 127) / 128 * 16;                       // whole 128-column tiles
    pn.rbs = (size_t)pn.KC * NPL * PIECE + 4352;
    const size_t bytes = (size_t)(pn.rows_pad / 64) * pn.rbs;
    CK(hipMalloc(&pn.p, bytes));
    const size_t n = (size_t)pn.rows_pad * pn.KC;
    hipLaunchKernelGGL(split_rowmajor_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dsrc, rows, cols, pn.p,
                       pn.rbs, pn.rows_pad, pn.KC);
    CK(hipGetLastError());
    return 0;
}

static int run(int M, int N, int Kd, bool check) {
    std::vector<float> hA((size_t)Kd * M), hB((size_t)Kd * N);
    unsigned st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto &v : hA) v = rnd();
    for (auto &v : hB) v = rnd() * 3.f;
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0xFF, (size_t)M * N * 4));
    Panel pa, pb;
    if (make_panel(dA, Kd, M, pa) || make_panel(dB, Kd, N, pb)) return 1;
    Args a;
    a.Ap = pa.p; a.Bp = pb.p; a.rbs_a = pa.rbs; a.rbs_b = pb.rbs; a.C = dC; a.M = M; a.N = N; a.ldc = N;
    a.nk = (Kd + 31) / 32; a.tiles_m = (M + 127) / 128; a.tiles_n = (N + 127) / 128;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_km_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipLaunchKernelGGL(gemm_km_kernel, dim3(a.tiles_m * a.tiles_n), dim3(512), LDS_BYTES, 0, a);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    if (check) {
        std::vector<float> hC((size_t)M * N);
        CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, scale = 0;
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) {
                double ref = 0, mag = 0;
                for (int k = 0; k < Kd; ++k) {
                    ref += (double)hA[(size_t)k * M + m] * hB[(size_t)k * N + n];
                    mag += std::fabs((double)hA[(size_t)k * M + m] * hB[(size_t)k * N + n]);
                }
                const double e = std::fabs(hC[(size_t)m * N + n] - ref);
                if (!(e <= worst)) worst = e;          // NaN-catching
                if (mag > scale) scale = mag;
            }
        printf("check M=%d N=%d K=%d: max |err| %.3e / product scale %.3e = %.3e  (%s)\n", M, N, Kd, worst, scale,
               worst / scale, worst / scale < 1e-6 ? "OK" : "WRONG");
    } else {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(gemm_km_kernel, dim3(a.tiles_m * a.tiles_n), dim3(512), LDS_BYTES, 0, a);
        hipEventRecord(e1);
        CK(hipDeviceSynchronize());
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        printf("time  M=%d N=%d K=%d: %.3f ms  %.1f TF/s-equivalent (kernel only, K-major operands from row-major panels)\n",
               M, N, Kd, ms, 2.0 * M * N * Kd / ms * 1e-9);
    }
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(pa.p); hipFree(pb.p);
    return 0;
}

int main() {
    if (run(256, 384, 512, true)) return 1;          // transpose-detecting: M != N, random data
    if (run(200, 130, 100, true)) return 1;          // ragged M, N, K
    if (run(8192, 4096, 25600, false)) return 1;     // cfg3 layer-1 dW_ih
    if (run(4096, 1024, 51200, false)) return 1;     // cfg3 layer-0 dW_hh (one direction)
    return 0;
}
