// ------------------------------------------------------------------------------------------------
// Forward recurrence on the bf16 matrix cores (exact 3-way operand splitting, cf. gemm_split.hip).
//
// The f32 kernel above is bounded below by its MFMA work: a workgroup multiplies its 32 gate rows with
// h_{t-1} [32 x H] every step - 256 v_mfma_f32_16x16x4_f32 per wave = 8.2k cycles at H = 1024, half of the
// step.  An f32 number is exactly the sum of three bf16 numbers, so the same product is six
// v_mfma_f32_16x16x32_bf16 per (tile, 32 k): 192 MFMAs of ~16 cycles = 3.2k cycles per wave and step, with
// f32 accumulation and the three dropped partial products together below 2^-23 of the product (the error class of
// the f32 chain).  What changes against the f32 kernel:
//  * W_hh slice: planes 0 and 1 live in LDS in FRAGMENT order ([plane][mt][32-k step][lane][8 bf16] - every
//    ds_read_b128 is lane-linear, conflict-free), plane 2 lives in registers (64 VGPRs at H = 1024): the
//    three planes are 6 B per weight and 192 KiB would not fit the 160 KiB of LDS;
//  * the exchange carries h as three bf16 planes in B-fragment order ([32-k step][nt][plane][lane][8 bf16],
//    1-KiB pieces, one buffer_load_dwordx4 each); a producer quad holds 4 units of a batch row and writes
//    ONE 8-B write-through store per plane (lanes q = 0..2 of the quad store planes 0..2);
//  * the sentinel is still the data: 0xFFFF is a bf16 NaN and poisons the accumulator column.
// Everything else (canaries, slow path, cell update, saved tensors, fused time reduction, GRU mode) is the
// f32 kernel's.  Requires MT = 2 (8 units per workgroup = one 8-k fragment group) and H % 128 == 0.
#include "lstm_rec_common.h"

namespace asrk_rec {
namespace {

template <int MT, int NT, bool DB, bool GRU, int KSW>
__global__ __launch_bounds__(256) void lstm_rec_fwd_bf_kernel(RecFwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(MT == 2 || MT == 4, "8 or 16 units per workgroup");
    constexpr int NLP = MT == 2 ? 2 : 1;         // slice planes in LDS (the other 3 - NLP live in registers)
    constexpr int NRP = 3 - NLP;
    constexpr int U = 4 * MT;
    constexpr int CL = MT * NT * 64;
    constexpr int CW = CL / 4;
    constexpr int CPT = (CW + 63) / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = wave;                         // K quarter of this wave (= its SIMD)
    const int ngroups = p.ndir * p.nbg;
    const int group = blockIdx.x % ngroups, wg = blockIdx.x / ngroups;
    const int dir = p.dir0 + group % p.ndir, bg = p.bg0 + group / p.ndir;
    const int u0 = wg * U, b0 = bg * p.BG;
    const int nb = min(p.BG, p.B - b0);
    const int H = p.H;
    const int KS_TOT = 4 * KSW;                  // 32-k steps over the whole hidden size (H = 128 * KSW)

    unsigned char *Wl = reinterpret_cast<unsigned char *>(smem);   // [NLP planes][MT][KS_TOT][64 lanes][16 B]
    constexpr int CLP = MT * NT * RED_PITCH;
    f32x4 *red = reinterpret_cast<f32x4 *>(Wl + (size_t)NLP * MT * KS_TOT * 1024);
    int *abort_flag = reinterpret_cast<int *>(red + (DB ? 2 : 1) * 4 * CLP);

    const int m16 = lane & 15, q4 = lane >> 4;
    // ---- W_hh slice: every thread splits exactly the fragments it will multiply with
    bf16x8_t areg[MT][KSW][NRP];                // planes NLP..2
    {
        const float *W = p.whh[dir];
#pragma unroll
        for (int mtl = 0; mtl < MT; ++mtl) {
            const int mt = mtl;
            const int m = mt * 16 + m16, unit = u0 + (m >> 2), gate = m & 3;
            const bool live = !GRU || gate < 3;
            const float *wrow = W + (size_t)(gate * H + unit) * H;
#pragma unroll
            for (int j = 0; j < KSW; ++j) {
                const int g = kq * KSW + j, k = g * 32 + q4 * 8;
                unsigned h0[8], h1[8], h2[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) split3(live ? wrow[k + e] : 0.f, h0[e], h1[e], h2[e]);
                u32x4 w0, w1, w2;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    w0[q] = h0[2 * q] | (h0[2 * q + 1] << 16);
                    w1[q] = h1[2 * q] | (h1[2 * q + 1] << 16);
                    w2[q] = h2[2 * q] | (h2[2 * q + 1] << 16);
                }
                const u32x4 wp[3] = {w0, w1, w2};
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    if (pl < NLP)
                        *reinterpret_cast<u32x4 *>(Wl + ((size_t)((pl * MT + mt) * KS_TOT + g) * 64 + lane) * 16) = wp[pl];
                    else
                        areg[mtl][j][pl - NLP] = __builtin_bit_cast(bf16x8_t, wp[pl]);
                }
            }
        }
        if (tid == 0) {
            abort_flag[0] = 0;
            abort_flag[1] = 0;
        }
    }
    __syncthreads();

    // ---- static cell-lane ownership (as in the f32 kernel; wave w owns tile (mt, nt) = (w / NT, w % NT) when
    // CW == 64)
    int c_unit[CPT], c_b[CPT], c_xoff[CPT], c_pc[CPT], c_ph[CPT];
    bool c_valid[CPT];
    float c_state[CPT];
    int c_cl[CPT];
    const int xg = u0 >> 5, xq4 = (u0 & 31) >> 3;     // first 8-k fragment group this workgroup produces (U / 8 of them)
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int lw = lane + 64 * i;
        const int idx = kq * CW + (lw < CW ? lw : 0);
        const int q = idx & 3, n = (idx >> 2) & 15, blk = idx >> 6;
        const int nt = blk % NT, mt = blk / NT;
        c_cl[i] = blk * RED_PITCH + red_slot(q * 16 + n);
        c_unit[i] = u0 + mt * 4 + q;
        const int bl = nt * 16 + n;
        c_b[i] = b0 + bl;
        c_valid[i] = (lw < CW) && (bl < nb) && (c_unit[i] < H);
        // byte offset of the 8-B store of plane min(q, 2): piece (xg, nt, plane), lane slot (xq4 + mt/2, n), half mt&1
        c_xoff[i] = ((((xg * NT + nt) * 3 + min(q, 2)) * 64 + (xq4 + (mt >> 1)) * 16 + n) * 16) + (mt & 1) * 8;
        c_pc[i] = (dir * H + u0 + (mt >> 1) * 8) >> 3;     // panel chunk column of this tile's 8-unit group (t % r == 0)
        c_ph[i] = (mt & 1) * 8;                            // byte half of the 16-byte slot
        c_state[i] = 0.f;
    }

    const int k_lo = kq * KSW * 32;
    const size_t data_floats = (size_t)KS_TOT * NT * 3 * 256;
    const size_t step_floats = data_floats + (size_t)p.canw;
    float *xgroup = p.X + (size_t)group * p.T * step_floats;

    // B-fragment offsets: piece ((g*NT + nt)*3 + plane), lane-linear; OOB for padded batch rows -> 0
    unsigned xoff[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
        xoff[nt] = (nt * 16 + m16 < nb) ? (unsigned)(((kq * KSW * NT + nt) * 3 * 64 + lane) * 16) : 0x7ffffff0u;
    constexpr unsigned KS_STRIDE = NT * 3 * 1024;   // bytes per 32-k step

    const int k_hi = k_lo + KSW * 32;
    const int wg_lo = k_lo / U;
    const int wg_cnt = (k_hi - 1) / U - wg_lo + 1;
    const int can_cnt = 4 * wg_cnt;

    float gpre[CPT][4];
    int c_len[CPT];                     // steps this cell's batch row takes (p.T without per-row lengths)
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) gpre[i][r] = 0.f;
        c_len[i] = (p.lens && c_valid[i]) ? min((int)p.lens[c_b[i]], p.T) : p.T;
        if (c_valid[i] && c_len[i] > 0) {
            const int t0 = dir == 0 ? 0 : c_len[i] - 1;
            const float *g = p.G + ((size_t)t0 * p.B + c_b[i]) * p.ldg + dir * 4 * H + c_unit[i];
#pragma unroll
            for (int r = 0; r < 4; ++r) gpre[i][r] = g[(size_t)r * H];
        }
    }

    const unsigned char *a_lds = Wl + ((size_t)(kq * KSW) * 64 + lane) * 16;   // + (plane*MT + mt)*KS_TOT KiB + j KiB

    for (int s = 0; s < p.T; ++s) {
        const int t = dir == 0 ? s : p.T - 1 - s;
        f32x4 acc[MT][NT][2];
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) acc[a][b][0] = acc[a][b][1] = f32x4{0.f, 0.f, 0.f, 0.f};

        // A fragments (planes 0, 1) of 32-k step j from LDS
        auto load_a = [&](bf16x8_t (&af)[MT][NLP], int j) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int pl = 0; pl < NLP; ++pl)
                    af[mt][pl] = *reinterpret_cast<const bf16x8_t *>(
                        a_lds + ((size_t)((pl * MT + mt) * KS_TOT + j)) * 1024);
        };
        // term g of one 32-k step (g = 0..5: the six partial products, small ones first): MT*NT MFMAs on
        // MT*NT different accumulator tiles; chains alternate between terms
        auto term = [&](int g, int j, const bf16x8_t (&af)[MT][NLP], const u32x4 (&bfr)[NT][3]) {
            const int pa = g == 0 ? 2 : (g == 1 || g == 3) ? 1 : 0;           // A plane: 2 1 0 1 0 0
            const int pb = g == 0 ? 0 : g == 1 ? 1 : g == 2 ? 2 : g == 3 ? 0 : g == 4 ? 1 : 0;   // B: 0 1 2 0 1 0
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt][g & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        pa >= NLP ? areg[mt][j][pa >= NLP ? pa - NLP : 0] : af[mt][pa < NLP ? pa : 0],
                        __builtin_bit_cast(bf16x8_t, bfr[nt][pb]),
                        acc[mt][nt][g & 1], 0, 0, 0);
        };
        // plain (not interleaved) 32-k step for the slow path
        auto kstep = [&](int j, const u32x4 (&bfr)[NT][3]) {
            bf16x8_t af[MT][NLP];
            load_a(af, j);
#pragma unroll
            for (int g = 0; g < 6; ++g) term(g, j, af, bfr);
        };

        REC_STAMP_W(0);
        if (s > 0) {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(xgroup + (size_t)(s - 1) * step_floats), 0, (int)(step_floats * 4), 0x00020000);
            constexpr int PF = KSW >= 4 ? 3 : KSW;      // 32-k steps of fragments in flight before the first MFMA
            constexpr int RING = MT == 4 ? 4 : KSW;      // fragment slots (MT = 4: registers are scarce -> PF + 1)
            static_assert(KSW % RING == 0 && RING > PF - 1 + (KSW > PF ? 1 : 0), "ring too short");
            u32x4 bf[RING][NT][3];
            unsigned spins = 0;
            unsigned long long t0 = 0;
            bool ok = true;
            {
                for (int z = (p.poll_mode >> 8) & 0xff; z > 0; z -= 8) __builtin_amdgcn_s_sleep(8);
                const unsigned *cbase = reinterpret_cast<const unsigned *>(
                    xgroup + (size_t)(s - 1) * step_floats + data_floats);
                volatile int *ready = abort_flag + 1;
                if (p.poll_mode & 2) {
                    if (wave == 0) {
                        ok = wait_canaries(cbase, 4 * p.nwg, p.err, lane, p.poll_mode);
                        if (lane == 0) *ready = ok ? s : -1;
                    } else {
                        int r;
                        while ((r = *ready) != s && r != -1) __builtin_amdgcn_s_sleep(1);
                        ok = (r == s);
                    }
                } else {
                    ok = wait_canaries(cbase + 4 * wg_lo, can_cnt, p.err, lane, p.poll_mode);
                }
            }
            REC_STAMP_W(7);
            bool bad = false;
            if (ok) {
#pragma unroll
                for (int j = 0; j < PF; ++j)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            bf[j][nt][pl] = __builtin_amdgcn_raw_buffer_load_b128(rs, xoff[nt] + pl * 1024,
                                                                                  j * KS_STRIDE, 0);
                __builtin_amdgcn_sched_barrier(0);
                REC_STAMP_W(1);
                // A wave that is issuing a 1-KiB fragment load (~50-60 cycles) cannot issue MFMAs, and a block
                // of 24 MFMAs (~400 cycles) keeps it from issuing loads: the loads of step j + PF are therefore
                // interleaved ONE at a time between the six MFMA groups of step j (a group of 4 MFMAs is about
                // one load issue long), pinned with scheduling barriers; A fragments run one step ahead.
                bf16x8_t afr[2][MT][NLP];
                load_a(afr[0], 0);
#pragma unroll
                for (int j = 0; j < KSW; ++j) {
                    if (j + 1 < KSW) load_a(afr[(j + 1) & 1], j + 1);
#pragma unroll
                    for (int g = 0; g < 6; ++g) {
                        term(g, j, afr[j & 1], bf[j % RING]);
                        __builtin_amdgcn_sched_barrier(0);
                        constexpr int every = 6 / (NT * 3);         // 2 batch tiles: after every group, 1: every 2nd
                        if (j + PF < KSW && g % every == 0) {
                            const int li = g / every, nt = li / 3, pl = li % 3;
                            bf[(j + PF) % RING][nt][pl] = __builtin_amdgcn_raw_buffer_load_b128(
                                rs, xoff[nt] + pl * 1024, (j + PF) * KS_STRIDE, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bad |= any_nan(acc[mt][nt][0] + acc[mt][nt][1]);
            }
            if (ok && __any(bad)) {
                // slow path: L1/L2-bypassing reloads, one ring at a time, verified against the sentinel first
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b) acc[a][b][0] = acc[a][b][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j0 = 0; j0 < KSW; j0 += RING) {
                    while (ok) {
#pragma unroll
                        for (int j = 0; j < RING; ++j)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                                for (int pl = 0; pl < 3; ++pl)
                                    bf[j][nt][pl] = __builtin_amdgcn_raw_buffer_load_b128(
                                        rs, xoff[nt] + pl * 1024, (j0 + j) * KS_STRIDE, 16);
                        __builtin_amdgcn_sched_barrier(0);
                        bool b2 = false;
#pragma unroll
                        for (int j = 0; j < RING; ++j)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                                for (int pl = 0; pl < 3; ++pl)
                                    b2 |= has_sentinel(__builtin_bit_cast(f32x4, bf[j][nt][pl]));
                        if (!__any(b2)) break;
                        if (!spin_ok(spins, t0, p.err, lane)) ok = false;
                    }
                    if (ok) {
#pragma unroll
                        for (int j = 0; j < RING; ++j) kstep(j0 + j, bf[j]);
                    }
                }
            }
            if (!ok && lane == 0) *abort_flag = 1;
        }
        REC_STAMP_W(2);
        f32x4 *redw = red + (DB ? (s & 1) : 0) * 4 * CLP;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                redw[((kq * MT + mt) * NT + nt) * RED_PITCH + red_slot(lane)] = acc[mt][nt][0] + acc[mt][nt][1];
        REC_STAMP_W(3);
        __syncthreads();
        if (*abort_flag) break;
        REC_STAMP_W(4);

        float gi[CPT], gf[CPT], gg[CPT], go[CPT], hv[CPT];
        float *xstep = xgroup + (size_t)s * step_floats;
        __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)xstep, 0, (int)(step_floats * 4), 0x00020000);
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            hv[i] = 0.f;
            gi[i] = gf[i] = gg[i] = go[i] = 0.f;
            if (c_valid[i] && s < c_len[i]) {   // a row past its own length keeps h = 0 in the exchange
                const int cl = c_cl[i];
                f32x4 sum = redw[cl];
#pragma unroll
                for (int w = 1; w < 4; ++w) sum += redw[w * CLP + cl];
                if (GRU) {
                    gi[i] = fast_sigmoid(gpre[i][0] + sum[0]);
                    gf[i] = fast_sigmoid(gpre[i][1] + sum[1]);
                    go[i] = sum[2] + gpre[i][3];
                    gg[i] = fast_tanh(gpre[i][2] + gi[i] * go[i]);
                    hv[i] = (1.f - gf[i]) * gg[i] + gf[i] * c_state[i];
                    c_state[i] = hv[i];
                } else {
                    gi[i] = fast_sigmoid(gpre[i][0] + sum[0]);
                    gf[i] = fast_sigmoid(gpre[i][1] + sum[1]);
                    gg[i] = fast_tanh(gpre[i][2] + sum[2]);
                    go[i] = fast_sigmoid(gpre[i][3] + sum[3]);
                    c_state[i] = gf[i] * c_state[i] + gi[i] * gg[i];
                    hv[i] = go[i] * fast_tanh(c_state[i]);
                }
            }
        }
        // exchange payload: the quad's 4 units of one batch row, plane q from lane q (q = 0..2), 8 B each
        u32x2 stv[CPT];
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            unsigned h0, h1, h2;
            split3(hv[i], h0, h1, h2);
            const int p01 = (int)(h0 | (h1 << 16)), p2 = (int)h2;
            int w01[4], w2[4];
            w01[0] = __builtin_amdgcn_mov_dpp(p01, 0x00, 0xf, 0xf, true);
            w01[1] = __builtin_amdgcn_mov_dpp(p01, 0x55, 0xf, 0xf, true);
            w01[2] = __builtin_amdgcn_mov_dpp(p01, 0xAA, 0xf, 0xf, true);
            w01[3] = __builtin_amdgcn_mov_dpp(p01, 0xFF, 0xf, 0xf, true);
            w2[0] = __builtin_amdgcn_mov_dpp(p2, 0x00, 0xf, 0xf, true);
            w2[1] = __builtin_amdgcn_mov_dpp(p2, 0x55, 0xf, 0xf, true);
            w2[2] = __builtin_amdgcn_mov_dpp(p2, 0xAA, 0xf, 0xf, true);
            w2[3] = __builtin_amdgcn_mov_dpp(p2, 0xFF, 0xf, 0xf, true);
            const int q = lane & 3;
            unsigned v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                v[u] = q == 0 ? ((unsigned)w01[u] & 0xffffu) : q == 1 ? ((unsigned)w01[u] >> 16) : (unsigned)w2[u];
            u32x2 st;
            st[0] = v[0] | (v[1] << 16);
            st[1] = v[2] | (v[3] << 16);
            if (c_valid[i] && q < 3)
                __builtin_amdgcn_raw_buffer_store_b64(st, xrs, (unsigned)c_xoff[i], 0, 16);
            stv[i] = st;
        }
        // publish FIRST: the canary goes out right behind the exchange stores; everything the next step's consumers do
        // not wait for (the next layer's panel, Y, the reduced-time copy, gates, cell state) is stored after it
        if (lane == 0)
            __hip_atomic_store(reinterpret_cast<unsigned *>(xstep + data_floats) + 4 * wg + wave,
                               (unsigned)(s + 1), RLX_AGENT);
        if (p.P2) {
            // the same 8 bytes (plane q of the quad's 4 units) into the next layer's A panel: row (t / r, b), chunk
            // column = this 8-unit group's place in the (t % r, direction, unit) feature axis
            const int r = p.pyr_rate, tq = t / r, tr = t - tq * r;
            if (tq < p.T / r) {
                const int q = lane & 3;
#pragma unroll
                for (int i = 0; i < CPT; ++i) {
                    const int m = tq * p.B + c_b[i];
                    const size_t off = (size_t)(m >> 6) * p.p2_stride +
                                       (size_t)((c_pc[i] + tr * (p.ldy >> 3)) * 3 + min(q, 2)) * 1024 + (m & 63) * 16 + c_ph[i];
                    if (c_valid[i] && q < 3) *reinterpret_cast<u32x2 *>(p.P2 + off) = stv[i];
                }
            }
        }
        int c_t[CPT];                   // the frame this cell's row is at (per row with p.lens, else the uniform t)
        bool c_live[CPT];
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            c_t[i] = (p.lens && dir != 0) ? c_len[i] - 1 - s : t;
            c_live[i] = c_valid[i] && s < c_len[i];
        }
#pragma unroll
        for (int i = 0; i < CPT; ++i)
            if (c_live[i])
                p.Y[((size_t)c_t[i] * p.B + c_b[i]) * p.ldy + dir * H + c_unit[i]] = hv[i];
        if (p.Y2) {
            const int r = p.pyr_rate;
            const size_t ld2 = p.pyr_mode == 1 ? (size_t)r * p.ldy : (size_t)p.ldy;
            if (!p.lens) {              // training: one frame index for the whole step (scalar arithmetic)
                const int tq = t / r, tr = t - tq * r;
                if (p.pyr_mode == 1 ? tq < p.T / r : tr == 0) {
                    const size_t off = p.pyr_mode == 1 ? (size_t)tr * p.ldy : 0;
#pragma unroll
                    for (int i = 0; i < CPT; ++i)
                        if (c_valid[i])
                            p.Y2[((size_t)tq * p.B + c_b[i]) * ld2 + off + dir * H + c_unit[i]] = hv[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < CPT; ++i) {
                    const int tq = c_t[i] / r, tr = c_t[i] - tq * r;
                    // 'concat' trims len % r frames of every row by itself (src/module.py:147-149 on the unpadded
                    // utterance); 'drop' keeps t % r == 0
                    const bool keep = p.pyr_mode == 1 ? tq < c_len[i] / r : tr == 0;
                    const size_t off = p.pyr_mode == 1 ? (size_t)tr * p.ldy : 0;
                    if (c_live[i] && keep)
                        p.Y2[((size_t)tq * p.B + c_b[i]) * ld2 + off + dir * H + c_unit[i]] = hv[i];
                }
            }
        }
        REC_STAMP_W(5);
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            if (c_live[i]) {
                const int tn = dir == 0 ? c_t[i] + 1 : c_t[i] - 1;
                float *g = p.G + ((size_t)c_t[i] * p.B + c_b[i]) * p.ldg + dir * 4 * H + c_unit[i];
                g[0] = gi[i];
                g[(size_t)H] = gf[i];
                g[(size_t)2 * H] = gg[i];
                g[(size_t)3 * H] = go[i];
                if (!GRU) p.C[((size_t)c_t[i] * p.B + c_b[i]) * p.ldy + dir * H + c_unit[i]] = c_state[i];
                if (s + 1 < c_len[i]) {
                    const float *gn =
                        p.G + ((size_t)tn * p.B + c_b[i]) * p.ldg + dir * 4 * H + c_unit[i];
#pragma unroll
                    for (int r = 0; r < 4; ++r) gpre[i][r] = gn[(size_t)r * H];
                }
            }
        }
        if (p.rearm && s >= 2)
            rearm_region(xgroup + (size_t)(s - 2) * step_floats, step_floats, wg, p.nwg, tid, (int)blockDim.x);
        REC_STAMP_W(6);
        if (!DB) __syncthreads();
    }
}

template <int MT, int NT, bool DB, bool GRU, int KSW>
int launch_fwd_bf(const RecFwdArgs &a, int grid, size_t lds, hipStream_t s) {
    auto kern = lstm_rec_fwd_bf_kernel<MT, NT, DB, GRU, KSW>;
    ASRK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

template <bool GRU>
int launch_fwd_bf_plan(const RecFwdArgs &a, const FwdPlan &pl, int H, int grid, hipStream_t s) {
    const int ksw = H / 128;
#define ASRK_BF_CASE(NT_, DB_, KSW_)                                                   \
    if (pl.MT == 2 && pl.NT == NT_ && (pl.db != 0) == DB_ && ksw == KSW_)              \
        return launch_fwd_bf<2, NT_, DB_, GRU, KSW_>(a, grid, pl.lds, s);
    ASRK_BF_CASE(1, true, 4) ASRK_BF_CASE(1, false, 4) ASRK_BF_CASE(2, true, 4) ASRK_BF_CASE(2, false, 4)
    ASRK_BF_CASE(1, true, 8) ASRK_BF_CASE(1, false, 8) ASRK_BF_CASE(2, true, 8) ASRK_BF_CASE(2, false, 8)
#undef ASRK_BF_CASE
    if (pl.MT == 4 && pl.NT == 1 && pl.db == 0 && ksw == 8)
        return launch_fwd_bf<4, 1, false, GRU, 8>(a, grid, pl.lds, s);
    return ASRK_ESHAPE;
}

}  // namespace

int launch_fwd_bf(bool gru, const RecFwdArgs &a, const FwdPlan &pl, int H, int grid, hipStream_t s) {
    return gru ? launch_fwd_bf_plan<true>(a, pl, H, grid, s) : launch_fwd_bf_plan<false>(a, pl, H, grid, s);
}

}  // namespace asrk_rec
