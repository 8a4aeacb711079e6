// The split GEMM with a K-MAJOR left operand (and optionally a K-major right operand), gfx950.
//
//   C[M,N] = alpha * sum_k A[k][m] * B(n, k) + beta * C          A given as the ROW-major split panel of [k][m]
//
// Why: the weight gradients of an LSTM layer, dW_ih = dG^T X and dW_hh = dG^T H_prev (autograd of nn.LSTM,
// /root/reference/src/module.py:131), contract over the tokens with dG^T on the left.  gemm_split.hip wants that
// operand as a panel of [m rows][k] - a second, TRANSPOSED split pass over dG (3.3 of the 4.9 ms of split passes per
// cfg3 step) although the row-major panel of dG already exists: it is the left operand of dX = dG W.  This kernel
// reads dG straight from that row-major panel.
//
// A row-major panel stores pieces [64 k-rows][8 columns] bf16 per plane (gemm_split.hip); here k-rows are the
// contraction index and the columns are the output rows m.  A k-tile is 32 k-rows.  LDS image of a K-major
// operand: per 16-column group and plane one 1-KiB block [32 k][16 m] (32 B per k-row = two pieces side by side),
// filled by ONE LDS-DMA instruction with lane -> (k-row lane >> 1, piece lane & 1); the MFMA fragment (row m =
// lane & 31, 8 consecutive k) comes from two ds_read_b64_tr_b16 (lane i of a 16-lane group receives column i of the
// 4 x 16 tile the group addresses - tools/tr_probe.hip).  128 B of padding per 16-column group rotates the banks so
// that a 32-lane pass touches 64 distinct ones.  Everything else - 128 x 128 tile, 2 x 2 multiplying waves of
// 64 x 64, four DMA waves, 3-stage ring, one barrier per k-tile before the tile's last 16-k step, six bf16 MFMA
// products per f32 product - is gemm_bf16x6_kernel's.  Measured alone (tools/experimental/kmajor_gemm.hip):
// 8192 x 4096 x 25600 in 8.0 ms = 215 TF/s-equivalent, the rate of the [m][k]-panel kernel WITHOUT its split passes.
#include "common.h"
#include "knobs.h"

extern "C" size_t asrk_split_panel_bytes(int rows, int K, int flags);

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int PIECE = 1024, NPL = 3, NST = 3, NC = 4;
constexpr int GROUP = NPL * PIECE + 128;         // K-major operand: one 16-column group (3 plane blocks + pad)
constexpr int REGION_KM = 8 * GROUP;             // 128 columns
constexpr int RB_N = NC * NPL * PIECE;           // [m][k] operand: one 64-row block of a k-tile (12 KiB)
constexpr int REGION_N = 2 * RB_N;

struct KmArgs {
    const unsigned char *Ap, *Bp;                // Ap: first byte of the panel (K-major); Bp: see BKM
    size_t rbs_a, rbs_b;
    float *C;
    int M, N, ldc, nk, tiles_m, tiles_n;
    int a_kt0, a_cg0;                            // A: first k-tile (a_row0 / 32), first column group (a_k0 / 8)
    int b_kt0, b_cg0;                            // B K-major: same; B [n][k]: unused (folded into Bp)
    float alpha, beta;
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
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char *p) {        // k .. k+3 and k+4 .. k+7 of one column
    const s16x4 a = tr_read(p), b = tr_read(p + 128);
    return __builtin_bit_cast(bf16x8, s16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
}

// BKM: the right operand is K-major too (row-major panel of [k][n]); otherwise an [n rows][k] panel as in
// gemm_bf16x6_kernel (Bp then points at the first row block and chunk column of the call).
template <bool BKM>
__global__ __launch_bounds__(512) void gemm_km_bf16x6_kernel(KmArgs p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    constexpr int REGION_B = BKM ? REGION_KM : REGION_N;
    constexpr int STAGE = REGION_KM + REGION_B;
    constexpr int LPT = 12;                                  // LDS-DMA instructions per DMA wave and k-tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware tile order of gemm_bf16x6_kernel
    const int ntiles = p.tiles_m * p.tiles_n, bid = blockIdx.x;
    const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = bid & 7, loc = bid >> 3;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    constexpr int BAND = 8;
    const int band = tile / (BAND * p.tiles_m);
    const int band_w = min(BAND, p.tiles_n - band * BAND);
    const int in_band = tile - band * BAND * p.tiles_m;
    const int tm = in_band / band_w, tn = band * BAND + in_band % band_w;
    const int nk = p.nk;

    if (wave >= 4) {
        const int d = wave - 4;                              // DMA waves 0, 1: A groups 0-3 / 4-7; 2, 3: B
        const bool isA = d < 2;
        const unsigned char *gbase;
        unsigned char *lbase;
        size_t rbs;
        int kt0;
        const bool km = isA || BKM;
        if (km) {
            const int cg = (isA ? p.a_cg0 + tm * 16 : p.b_cg0 + tn * 16) + (d & 1) * 8 + (lane & 1);
            gbase = (isA ? p.Ap : p.Bp) + (size_t)cg * NPL * PIECE + (lane >> 1) * 16;
            lbase = lds + (isA ? 0 : REGION_KM) + (d & 1) * 4 * GROUP;
            rbs = isA ? p.rbs_a : p.rbs_b;
            kt0 = isA ? p.a_kt0 : p.b_kt0;
        } else {
            gbase = p.Bp + (size_t)(tn * 2 + (d & 1)) * p.rbs_b + lane * 16;
            lbase = lds + REGION_KM + (d & 1) * RB_N;
            rbs = 0;
            kt0 = 0;
        }
        auto issue = [&](int kt, int stage) {
            unsigned char *l = lbase + stage * STAGE;
            if (km) {
                const int q = kt0 + kt;
                const unsigned char *g = gbase + (size_t)(q >> 1) * rbs + (q & 1) * 512;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
                        glds16(g + (size_t)(j * 2 * NPL + pl) * PIECE, l + j * GROUP + pl * PIECE);
            } else {
                const unsigned char *g = gbase + (size_t)kt * RB_N;
#pragma unroll
                for (int j = 0; j < NC * NPL; ++j) glds16(g + j * PIECE, l + j * PIECE);
            }
        };
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
            if (min(NST - 2, nk - 2 - kt) == 1) wait_vm<LPT>();       // tile kt + 1 has landed
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

    // K-major fragment of row tile i, plane pl, 16-k step ks: group (w * 4 + 2 i + g16), k-row 16 ks + 8 h + (s >> 2)
    // (+ 4 for the second read), 8-byte quad s & 3
    const int g16 = (lane >> 4) & 1, h = lane >> 5, s16 = lane & 15;
    const int frag_km = g16 * GROUP + (8 * h + (s16 >> 2)) * 32 + (s16 & 3) * 8;
    const int frag_n = (h * NPL) * PIECE + (lane & 31) * 16;
    const unsigned char *abase = lds + wr * 4 * GROUP + frag_km;
    const unsigned char *bbase = lds + REGION_KM + (BKM ? wc * 4 * GROUP + frag_km : wc * RB_N + frag_n);

    bf16x8 fa[2][2][NPL], fb[2][2][NPL];
    auto load_frags = [&](int buf, int stage, int ks) {
        const unsigned char *a_st = abase + stage * STAGE + ks * 16 * 32;
        const unsigned char *b_st = bbase + stage * STAGE + (BKM ? ks * 16 * 32 : ks * 2 * NPL * PIECE);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                fa[buf][i][pl] = tr_frag(a_st + i * 2 * GROUP + pl * PIECE);
                if (BKM) fb[buf][i][pl] = tr_frag(b_st + i * 2 * GROUP + pl * PIECE);
                else fb[buf][i][pl] = *reinterpret_cast<const bf16x8 *>(b_st + pl * PIECE + i * 512);
            }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mfmas = [&](int buf) {
#define ASRK_TERM(PA, PB)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)            \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf][i][PA], fb[buf][j][PB], acc[i][j], 0, 0, 0);
        ASRK_TERM(2, 0) ASRK_TERM(1, 1) ASRK_TERM(0, 2) ASRK_TERM(1, 0) ASRK_TERM(0, 1) ASRK_TERM(0, 0)
#undef ASRK_TERM
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
        wait_lgkm0();                            // this wave's reads of the stage are done
        __builtin_amdgcn_s_barrier();            // every wave's are: the DMA waves may refill it
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
                if (row >= p.M) continue;
                float *c = p.C + (size_t)row * p.ldc + col;
                float v = p.alpha * acc[i][j][r];
                if (p.beta != 0.f) v += p.beta * *c;
                *c = v;
            }
    }
}

struct Geom { int KC, rb; size_t rbs, bytes; };
// must equal gemm_split.hip's panel_geom(rows, K, 3 planes, slack = true); checked against asrk_split_panel_bytes
Geom geom(int rows, int K) {
    const int pad = asrk_knobs_().get(asrk_knobs_().split_pad, 4352);
    Geom g;
    g.KC = (asrk_div_up(K, 32) + 1) * 4;
    g.rb = asrk_div_up(rows, 128) * 2;
    g.rbs = (size_t)g.KC * NPL * PIECE + (size_t)(pad / 16 * 16);
    g.bytes = (size_t)g.rb * g.rbs;
    return g;
}

template <bool BKM>
int launch(const KmArgs &a, hipStream_t s) {
    constexpr int lds = NST * (REGION_KM + (BKM ? REGION_KM : REGION_N));
    auto kern = gemm_km_bf16x6_kernel<BKM>;
    static AsrkLdsLatch latch;
    ASRK_HIP(asrk_max_lds_once(latch, reinterpret_cast<const void *>(kern), lds));
    hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(512), lds, s, a);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

}  // namespace

// include/asrk.h.  A_panel: asrk_split_panel_f32(trans = 0, flags = 0) of the [a_rows = contraction extent][a_K =
// output-row extent] matrix; a_row0 (multiple of 32) / a_k0 (multiple of 128): first contraction index / first
// output row used.  B: the same form when b_kmajor, else an [n rows][k] panel exactly as in asrk_gemm_panels_f32.
extern "C" int asrk_gemm_panels_km_f32(int M, int N, int K, float alpha, const void *A_panel, int a_rows, int a_K,
                                       int a_row0, int a_k0, const void *B_panel, int b_rows, int b_K, int b_row0,
                                       int b_k0, int b_kmajor, float beta, float *C, int ldc, void *stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !A_panel || !B_panel || !C || ldc < N) return ASRK_EINVAL;
    if (a_row0 < 0 || a_k0 < 0 || b_row0 < 0 || b_k0 < 0 || a_row0 % 32 || a_k0 % 128) return ASRK_EINVAL;
    if (a_row0 + K > a_rows || a_k0 + M > a_K) return ASRK_EINVAL;
    if (b_kmajor ? (b_row0 % 32 || b_k0 % 128 || b_row0 + K > b_rows || b_k0 + N > b_K)
                 : (b_row0 % 128 || b_k0 % 8 || b_row0 + N > b_rows || b_k0 + K > b_K))
        return ASRK_EINVAL;
    const Geom ga = geom(a_rows, a_K), gb = geom(b_rows, b_K);
    if (ga.bytes != asrk_split_panel_bytes(a_rows, a_K, 0) || gb.bytes != asrk_split_panel_bytes(b_rows, b_K, 0))
        return ASRK_EINVAL;                      // the two files disagree about the panel geometry
    const int nk = asrk_div_up(K, 32);
    // a ragged last k-tile must run into the zero padding of at least one operand
    const bool a_ends = a_row0 + K == a_rows;
    const bool b_ends = b_kmajor ? b_row0 + K == b_rows : b_k0 + K == b_K;
    if (K % 32 != 0 && !a_ends && !b_ends) return ASRK_ESHAPE;
    // everything a tile reads stays inside the panels (output rows beyond M / N are computed and never stored)
    if (a_row0 + nk * 32 > ga.rb * 64 || a_k0 / 8 + asrk_div_up(M, 128) * 16 > ga.KC) return ASRK_ESHAPE;
    if (b_kmajor ? (b_row0 + nk * 32 > gb.rb * 64 || b_k0 / 8 + asrk_div_up(N, 128) * 16 > gb.KC)
                 : (b_k0 / 8 + nk * NC > gb.KC))
        return ASRK_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    KmArgs a;
    a.Ap = reinterpret_cast<const unsigned char *>(A_panel);
    a.rbs_a = ga.rbs; a.rbs_b = gb.rbs;
    a.a_kt0 = a_row0 / 32; a.a_cg0 = a_k0 / 8;
    const unsigned char *B0 = reinterpret_cast<const unsigned char *>(B_panel);
    if (b_kmajor) {
        a.Bp = B0; a.b_kt0 = b_row0 / 32; a.b_cg0 = b_k0 / 8;
    } else {
        a.Bp = B0 + (size_t)(b_row0 / 64) * gb.rbs + (size_t)(b_k0 / 8) * NPL * PIECE;
        a.b_kt0 = 0; a.b_cg0 = 0;
    }
    a.C = C; a.M = M; a.N = N; a.ldc = ldc; a.nk = nk;
    a.tiles_m = asrk_div_up(M, 128); a.tiles_n = asrk_div_up(N, 128);
    a.alpha = alpha; a.beta = beta;
    asrk_prof_work_(PROF_GEMM, 2.0 * (double)M * (double)N * (double)K);
    asrk_prof_begin_(PROF_GEMM, s);
    const int rc = b_kmajor ? launch<true>(a, s) : launch<false>(a, s);
    asrk_prof_end_(PROF_GEMM, s);
    return rc;
}
