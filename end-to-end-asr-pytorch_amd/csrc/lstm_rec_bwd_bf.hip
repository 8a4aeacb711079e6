// ------------------------------------------------------------------------------------------------
// BPTT on the bf16 matrix cores (exact 3-way operand splitting; see lstm_rec_fwd_bf_kernel / gemm_split.hip).
// Plan geometry: 16 units x 16 batch rows per workgroup (UB = 16, NT = 1), wave w contracts gate w:
// rec[b, u] = sum_j dG[b, w*H + j] W_hh[w*H + j][u].  KS = H / 32 k-steps per wave, six
// v_mfma_f32_16x16x32_bf16 each (f32 kernel: 8 v_mfma_f32_16x16x4_f32 per 32 k at twice the cycles).
//  * W_hh^T slice as three bf16 planes = 6 B per weight (384 KiB per workgroup at H = 1024): the first NREG
//    k-steps of every wave live in registers (12 VGPRs per k-step), the rest in LDS in fragment order;
//  * the exchange carries dG as three bf16 planes in B-fragment order ([gate][32-k step][plane][lane][8 bf16]);
//    a cell thread owns one (unit, batch row), so its 12 plane values go through a 6-KiB LDS staging image
//    (written and read by the same wave: no barrier) and leave as 16-byte write-through stores;
//  * fragments: a ring of CH k-steps (3 loads each) in flight, refilled after the MFMAs that consumed a slot -
//    the step moves 96 KiB per wave through the CU's 64 B/clk vector-memory path, which is what bounds it.
#include "lstm_rec_common.h"

namespace asrk_rec {
namespace {

template <bool GRU, int KS, int NREG>
__global__ __launch_bounds__(256) void lstm_rec_bwd_bf_kernel(RecBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CH = 8;                        // k-steps in the fragment ring
    constexpr int KL = KS - NREG;                // LDS-resident k-steps per wave
    static_assert(KS >= CH, "ring longer than the slice");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ngroups = p.ndir * p.nbg;
    const int group = blockIdx.x % ngroups, wg = blockIdx.x / ngroups;
    const int dir = p.dir0 + group % p.ndir, bg = p.bg0 + group / p.ndir;
    const int u0 = wg * 16, b0 = bg * 16;
    const int nb = min(16, p.B - b0);
    const int H = p.H;

    unsigned char *Wl = reinterpret_cast<unsigned char *>(smem);          // [4 waves][KL][3 planes][64][16 B]
    f32x4 *red = reinterpret_cast<f32x4 *>(Wl + (size_t)4 * KL * 3 * 1024);   // [2 parity][4 waves][RED_PITCH]
    unsigned char *stage = reinterpret_cast<unsigned char *>(red + 2 * 4 * RED_PITCH);   // [4 gates][3][2][16][16 B]
    unsigned char *stage_t = stage + 4 * 3 * 2 * 16 * 16;   // transposed image [4 gates][3][2 row halves][16 units][8 rows x 2 B]
    int *abort_flag = reinterpret_cast<int *>(stage_t + 4 * 3 * 2 * 16 * 16);

    const int m16 = lane & 15, q4 = lane >> 4;
    // ---- W_hh^T slice: lane (unit m16, k-group q4) of wave w, k-step js: W_hh[w*H + js*32 + q4*8 + e][u0 + m16]
    bf16x8_t areg[NREG > 0 ? NREG : 1][3];
    {
        const float *W = p.whh[dir];
        const bool live = !GRU || wave < 3;
#pragma unroll
        for (int js = 0; js < KS; ++js) {
            unsigned h0[8], h1[8], h2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = js * 32 + q4 * 8 + e;
                split3(live ? W[((size_t)wave * H + j) * H + u0 + m16] : 0.f, h0[e], h1[e], h2[e]);
            }
            u32x4 w0, w1, w2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                w0[q] = h0[2 * q] | (h0[2 * q + 1] << 16);
                w1[q] = h1[2 * q] | (h1[2 * q + 1] << 16);
                w2[q] = h2[2 * q] | (h2[2 * q + 1] << 16);
            }
            if (js < NREG) {
                areg[js][0] = __builtin_bit_cast(bf16x8_t, w0);
                areg[js][1] = __builtin_bit_cast(bf16x8_t, w1);
                areg[js][2] = __builtin_bit_cast(bf16x8_t, w2);
            } else {
                unsigned char *d = Wl + ((size_t)((wave * KL + (js - NREG)) * 3) * 64 + lane) * 16;
                *reinterpret_cast<u32x4 *>(d) = w0;
                *reinterpret_cast<u32x4 *>(d + 1024) = w1;
                *reinterpret_cast<u32x4 *>(d + 2048) = w2;
            }
        }
        if (tid == 0) *abort_flag = 0;
    }
    __syncthreads();
    const unsigned char *a_lds = Wl + ((size_t)(wave * KL * 3) * 64 + lane) * 16;

    // the cell of this thread: (unit = tid % 16, batch row = tid / 16)
    const int ul = tid & 15, bl = tid >> 4;
    const int c_unit = u0 + ul, c_b = b0 + bl;
    const bool c_valid = bl < nb;
    const int c_red = (red_slot((ul >> 2) * 16 + bl)) * 4 + (ul & 3);
    float dc_carry = 0.f;
    float dbsum[4] = {0.f, 0.f, 0.f, 0.f};

    // exchange geometry: piece (gate, k-step, plane) = ((gate*KS + ks)*3 + plane) KiB; lane slot (q4, n)
    const size_t gate_floats = (size_t)KS * 3 * 256;
    const size_t data_floats = 4 * gate_floats;
    const size_t step_floats = data_floats + (size_t)p.canw;
    float *xgroup = p.X + (size_t)group * p.T * step_floats;
    const unsigned voff = (m16 < nb && p.dbg_steps != -1) ? (unsigned)(lane * 16) : 0x7ff00000u;
    const unsigned gate_base = (unsigned)(wave * gate_floats * 4);
    // staging image: ((gate*3 + plane)*2 + half)*256 + n*16 + (unit & 7)*2
    unsigned char *st_w = stage + (ul >> 3) * 256 + bl * 16 + (ul & 7) * 2;
    // transposed image: ((gate*3 + plane)*2 + row half)*256 + unit*16 + (row & 7)*2
    unsigned char *st_t = stage_t + (bl >> 3) * 256 + ul * 16 + (bl & 7) * 2;
    // this thread's chunk(s) of the workgroup's 384 transposed 16-byte chunks (8 batch rows of one unit, gate, plane)
    int pt_src[2];
    size_t pt_dst[2];
    int pt_half[2];
    bool pt_on[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;
        const int u = idx & 15, rest = idx >> 4, hb = rest & 1, gp = rest >> 1;   // gp = gate*3 + plane, 0..11
        const int r = gp / 3, pl = gp - r * 3;
        pt_on[i] = idx < 384;
        pt_src[i] = ((gp * 2 + hb) * 16 + u) * 16;
        const int row = (dir * 4 + r) * H + u0 + u;                               // gate column = panel row
        pt_dst[i] = (size_t)(row >> 6) * p.pt_stride + (size_t)pl * 1024 + (row & 63) * 16;
        pt_half[i] = hb;
    }
    // this lane's chunk(s) of the wave's 96 (4 batch rows x 24 (gate, plane, half)) 16-byte chunks
    const int xks = u0 >> 5, xq4 = (u0 & 31) >> 3;
    int ch_src[2], pg_row[2];
    unsigned ch_dst[2];
    size_t pg_col[2];
    bool ch_on[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = lane + 64 * i;
        const int nl = idx / 24, c = idx - nl * 24, n = wave * 4 + nl;
        const int r = c / 6, pl = (c % 6) >> 1, half = c & 1;
        ch_on[i] = idx < 96 && n < nb;
        ch_src[i] = ((r * 3 + pl) * 2 + half) * 256 + n * 16;
        ch_dst[i] = (unsigned)((((r * KS + xks) * 3 + pl) * 64 + (xq4 + half) * 16 + n) * 16);
        pg_row[i] = b0 + n;
        pg_col[i] = (size_t)((((dir * 4 + r) * H + u0 + half * 8) >> 3) * 3 + pl) * 1024;
    }

    for (int s = 0; s < p.T; ++s) {
        const int t = dir == 0 ? p.T - 1 - s : s;
        const int tp = dir == 0 ? t - 1 : t + 1;

        float vi = 0.f, vf = 0.f, vg = 0.f, vo = 0.f, vc = 0.f, vcp = 0.f, vdy = 0.f;
        if (c_valid) {
            const size_t row = (size_t)t * p.B + c_b;
            const float *g = p.G + row * p.ldg + dir * 4 * H + c_unit;
            vi = g[0];
            vf = g[(size_t)H];
            vg = g[(size_t)2 * H];
            vo = g[(size_t)3 * H];
            vc = p.C[row * p.ldy + dir * H + c_unit];
            if (p.pyr_mode == 0) {
                vdy = p.dY[row * p.ldy + dir * H + c_unit];
            } else {
                const int r = p.pyr_rate, tq = t / r, tr = t - tq * r;
                if (p.pyr_mode == 1 ? tq < p.T / r : tr == 0) {
                    const size_t ld2 = p.pyr_mode == 1 ? (size_t)r * p.ldy : (size_t)p.ldy;
                    const size_t off = p.pyr_mode == 1 ? (size_t)tr * p.ldy : 0;
                    vdy = p.dY[((size_t)tq * p.B + c_b) * ld2 + off + dir * H + c_unit];
                }
            }
            if (tp >= 0 && tp < p.T) vcp = p.C[((size_t)tp * p.B + c_b) * p.ldy + dir * H + c_unit];
        }

        f32x4 acc[3];
        acc[0] = acc[1] = acc[2] = f32x4{0.f, 0.f, 0.f, 0.f};

        // six partial products of one 32-k step on three accumulator chains
        auto mfma6 = [&](const bf16x8_t (&a)[3], const u32x4 (&b)[3]) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], __builtin_bit_cast(bf16x8_t, b[0]), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], __builtin_bit_cast(bf16x8_t, b[1]), acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], __builtin_bit_cast(bf16x8_t, b[2]), acc[2], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], __builtin_bit_cast(bf16x8_t, b[0]), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], __builtin_bit_cast(bf16x8_t, b[1]), acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], __builtin_bit_cast(bf16x8_t, b[0]), acc[2], 0, 0, 0);
        };
        auto a_frag = [&](bf16x8_t (&a)[3], int js) {      // js: compile-time after unrolling
            if (js < NREG) {
                a[0] = areg[js][0]; a[1] = areg[js][1]; a[2] = areg[js][2];
            } else {
                const unsigned char *q = a_lds + (size_t)(js - NREG) * 3 * 1024;
                a[0] = *reinterpret_cast<const bf16x8_t *>(q);
                a[1] = *reinterpret_cast<const bf16x8_t *>(q + 1024);
                a[2] = *reinterpret_cast<const bf16x8_t *>(q + 2048);
            }
        };

        REC_STAMP(0);
        if (s > 0) {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(xgroup + (size_t)(s - 1) * step_floats), 0, (int)(step_floats * 4), 0x00020000);
            u32x4 bf[CH][3];
            unsigned spins = 0;
            unsigned long long t0 = 0;
            bool ok = true;
            for (int z = (p.poll_mode >> 8) & 0xff; z > 0; z -= 8) __builtin_amdgcn_s_sleep(8);
            ok = wait_canaries(reinterpret_cast<const unsigned *>(
                                   xgroup + (size_t)(s - 1) * step_floats + data_floats),
                               4 * p.nwg, p.err, lane, p.poll_mode);
            REC_STAMP(7);
            bool bad = false;
            if (ok) {
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        bf[c][pl] = __builtin_amdgcn_raw_buffer_load_b128(
                            rs, voff, gate_base + (unsigned)((c * 3 + pl) * 1024), 0);
                __builtin_amdgcn_sched_barrier(0);
                REC_STAMP(1);
#pragma unroll
                for (int js = 0; js < KS; ++js) {
                    bf16x8_t a[3];
                    a_frag(a, js);
                    mfma6(a, bf[js % CH]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (js + CH < KS) {
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            bf[js % CH][pl] = __builtin_amdgcn_raw_buffer_load_b128(
                                rs, voff, gate_base + (unsigned)(((js + CH) * 3 + pl) * 1024), 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            bad = any_nan(acc[0] + acc[1] + acc[2]);
            if (ok && __any(bad)) {
                // slow path: L1/L2-bypassing reloads, verified against the sentinel pattern before use
                acc[0] = acc[1] = acc[2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c0 = 0; c0 < KS; c0 += CH) {
                    while (ok) {
#pragma unroll
                        for (int c = 0; c < CH; ++c)
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl)
                                bf[c][pl] = __builtin_amdgcn_raw_buffer_load_b128(
                                    rs, voff, gate_base + (unsigned)(((c0 + c) * 3 + pl) * 1024), 16);
                        __builtin_amdgcn_sched_barrier(0);
                        bool b2 = false;
#pragma unroll
                        for (int c = 0; c < CH; ++c)
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl) b2 |= has_sentinel(__builtin_bit_cast(f32x4, bf[c][pl]));
                        if (!__any(b2)) break;
                        if (!spin_ok(spins, t0, p.err, lane)) ok = false;
                    }
                    if (ok) {
#pragma unroll
                        for (int c = 0; c < CH; ++c) {
                            bf16x8_t a[3];
                            a_frag(a, c0 + c);
                            mfma6(a, bf[c]);
                        }
                    }
                }
            }
            if (!ok && lane == 0) *abort_flag = 1;
        }
        REC_STAMP(2);
        f32x4 *redw = red + (s & 1) * 4 * RED_PITCH;
        redw[wave * RED_PITCH + red_slot(lane)] = acc[0] + acc[1] + acc[2];
        REC_STAMP(3);
        __syncthreads();
        if (*abort_flag) break;
        REC_STAMP(4);

        const float *redf = reinterpret_cast<const float *>(redw);
        float *xstep = xgroup + (size_t)s * step_floats;
        float dgs[4] = {0.f, 0.f, 0.f, 0.f}, xg[4] = {0.f, 0.f, 0.f, 0.f};
        if (c_valid) {
            float rec = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) rec += redf[w * RED_PITCH * 4 + c_red];
            if (GRU) {
                const float dh = vdy + rec + dc_carry;
                const float r = vi, z = vf, n = vg, hn = vo, hp = vcp;
                const float dn = dh * (1.f - z) * (1.f - n * n);
                const float dz = dh * (hp - n) * z * (1.f - z);
                const float dr = dn * hn * r * (1.f - r);
                dc_carry = dh * z;
                dgs[0] = dr; dgs[1] = dz; dgs[2] = dn; dgs[3] = dn * r;
                xg[0] = dr; xg[1] = dz; xg[2] = dn * r; xg[3] = 0.f;
            } else {
                const float dh = vdy + rec;
                const float tc = fast_tanh(vc);
                const float dcell = dh * vo * (1.f - tc * tc) + dc_carry;
                dc_carry = dcell * vf;
                dgs[0] = dcell * vg * vi * (1.f - vi);
                dgs[1] = dcell * vcp * vf * (1.f - vf);
                dgs[2] = dcell * vi * (1.f - vg * vg);
                dgs[3] = dh * tc * vo * (1.f - vo);
#pragma unroll
                for (int r = 0; r < 4; ++r) xg[r] = dgs[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) dbsum[r] += dgs[r];
        }
        // exchange payload: 12 plane values per cell -> staging image -> 16-byte write-through stores
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            unsigned h0, h1, h2;
            split3(xg[r], h0, h1, h2);
            *reinterpret_cast<unsigned short *>(st_w + (r * 3 + 0) * 512) = (unsigned short)h0;
            *reinterpret_cast<unsigned short *>(st_w + (r * 3 + 1) * 512) = (unsigned short)h1;
            *reinterpret_cast<unsigned short *>(st_w + (r * 3 + 2) * 512) = (unsigned short)h2;
            if (!GRU && p.PT) {
                *reinterpret_cast<unsigned short *>(st_t + (r * 3 + 0) * 512) = (unsigned short)h0;
                *reinterpret_cast<unsigned short *>(st_t + (r * 3 + 1) * 512) = (unsigned short)h1;
                *reinterpret_cast<unsigned short *>(st_t + (r * 3 + 2) * 512) = (unsigned short)h2;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // same-wave LDS hand-over (no barrier needed)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
                (void *)xstep, 0, (int)(step_floats * 4), 0x00020000);
            u32x4 v[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (ch_on[i]) {
                    v[i] = *reinterpret_cast<const u32x4 *>(stage + ch_src[i]);
                    __builtin_amdgcn_raw_buffer_store_b128(v[i], xrs, ch_dst[i], 0, 16);
                }
            // publish first: the canary right behind the exchange stores, the panel image of the same chunks after it
            if (lane == 0)
                __hip_atomic_store(reinterpret_cast<unsigned *>(xstep + data_floats) + 4 * wg + wave,
                                   (unsigned)(s + 1), RLX_AGENT);
            if (!GRU && p.PG) {   // the same chunk = 8 units of one gate and row, three planes apart: dG's A panel
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    if (ch_on[i]) {
                        const int m = t * p.B + pg_row[i];
                        *reinterpret_cast<u32x4 *>(p.PG + (size_t)(m >> 6) * p.pg_stride + pg_col[i] + (m & 63) * 16) = v[i];
                    }
            }
        }
        REC_STAMP(5);
        if (c_valid) {
            float *g = p.G + ((size_t)t * p.B + c_b) * p.ldg + dir * 4 * H + c_unit;
#pragma unroll
            for (int r = 0; r < 4; ++r) g[(size_t)r * H] = dgs[r];
        }
        if (!GRU && p.PT) {
            // dG^T panel: a 16-byte slot = 8 consecutive batch rows of one gate column - two waves' values - hence the
            // barrier; it sits in the tail, behind the exchange stores and the canary (off the hand-off chain)
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (pt_on[i]) {
                    const int c = (t * p.B + b0 + pt_half[i] * 8) >> 3;           // chunk column of these 8 tokens
                    *reinterpret_cast<u32x4 *>(p.PT + pt_dst[i] + (size_t)c * 3072) =
                        *reinterpret_cast<const u32x4 *>(stage_t + pt_src[i]);
                }
        }
        if (p.rearm && s >= 2)
            rearm_region(xgroup + (size_t)(s - 2) * step_floats, step_floats, wg, p.nwg, tid, (int)blockDim.x);
        REC_STAMP(6);
    }
    if (p.db && !*abort_flag) {
        __syncthreads();
        float *sdb = reinterpret_cast<float *>(red);   // [256 cells][4] = 4 KiB <= the partial-sum buffer
#pragma unroll
        for (int r = 0; r < 4; ++r) sdb[tid * 4 + r] = c_valid ? dbsum[r] : 0.f;
        __syncthreads();
        if (tid < 64) {
            const int u = tid & 15, r = tid >> 4;
            float a = 0.f;
            for (int b = 0; b < 16; ++b) a += sdb[(b * 16 + u) * 4 + r];
            unsafeAtomicAdd(p.db + (size_t)dir * 4 * H + (size_t)r * H + u0 + u, a);
        }
    }
}

template <bool GRU, int KS, int NREG>
int launch_bwd_bf(const RecBwdArgs &a, int grid, size_t lds, hipStream_t s) {
    auto kern = lstm_rec_bwd_bf_kernel<GRU, KS, NREG>;
    ASRK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

}  // namespace

int launch_bwd_bf(bool gru, const RecBwdArgs &a, const BwdPlan &pl, int grid, hipStream_t s) {
    if (a.H == 1024) return gru ? launch_bwd_bf<true, 32, 21>(a, grid, pl.lds, s) : launch_bwd_bf<false, 32, 21>(a, grid, pl.lds, s);
    return gru ? launch_bwd_bf<true, 16, 16>(a, grid, pl.lds, s) : launch_bwd_bf<false, 16, 16>(a, grid, pl.lds, s);
}

}  // namespace asrk_rec
