// Persistent LSTM / GRU forward recurrence, exact-f32 MFMA form (design: lstm_rec.hip).
#include "lstm_rec_common.h"

namespace asrk_rec {
namespace {

// one k-group (16 k) of the forward product: j outermost so consecutive MFMAs hit different chains
template <int MT, int NT, int ACC, int KGW>
__device__ __forceinline__ void fwd_mfma_kgroup(f32x4 (&acc)[MT][NT][ACC], const f32x4 (&bf)[NT][KGW],
                                                const float *Ws, int HP, int m16, int k_lo, int kg,
                                                int q4) {
    f32x4 a[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        a[mt] = *reinterpret_cast<const f32x4 *>(Ws + (mt * 16 + m16) * HP + k_lo + kg * 16 + 4 * q4);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt][j % ACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                    a[mt][j], bf[nt][kg][j], acc[mt][nt][j % ACC], 0, 0, 0);
}

// GRU = true: the same persistent recurrence for torch.nn.GRU (gate order r, z, n) in the 4-slot-per-unit
// layout of the LSTM kernels: slot 3 has no recurrent weights (zero rows) and carries b_hn on the input
// side, so the cell sees  r = s(g0 + W_hr h), z = s(g1 + W_hz h), hn = W_hn h + g3, n = tanh(g2 + r hn),
// h' = (1-z) n + z h; the saved slots are (r, z, n, hn).
template <int MT, int NT, int KGW, bool DB, bool GRU>
__global__ __launch_bounds__(256) void lstm_rec_fwd_kernel(RecFwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int ACC = MT * NT >= 2 ? 2 : 4;  // accumulator chains per output tile
    constexpr int CL = MT * NT * 64;       // cell-lanes (one (unit,batch) cell each)
    constexpr int CW = CL / 4;             // cell-lanes per wave: every wave does cell work, so no
                                           // wave idles (and hot-spots the canary lines) meanwhile
    constexpr int CPT = (CW + 63) / 64;    // cell-lanes per thread
    // `wave` must be provably uniform: it feeds scalar operands (buffer-load soffset) and branch
    // conditions; a VGPR there costs a readfirstlane waterfall loop around EVERY load.
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ngroups = p.ndir * p.nbg;
    const int group = blockIdx.x % ngroups, wg = blockIdx.x / ngroups;
    const int dir = p.dir0 + group % p.ndir, bg = p.bg0 + group / p.ndir;
    const int u0 = wg * p.U, b0 = bg * p.BG;
    const int nb = min(p.BG, p.B - b0);
    const int H = p.H, HP = p.HP;

    float *Ws = smem;
    constexpr int CLP = MT * NT * RED_PITCH;   // padded entries per wave (see RED_PITCH)
    f32x4 *red = reinterpret_cast<f32x4 *>(smem + MT * 16 * HP);  // [2 parity][4 waves][CLP]
    int *abort_flag = reinterpret_cast<int *>(red + (DB ? 2 : 1) * 4 * CLP);

    // ---- stage this workgroup's W_hh rows: LDS row m <-> (unit u0 + m/4, gate m%4)
    {
        const float *W = p.whh[dir];
        for (int idx = tid; idx < MT * 16 * HP; idx += 256) {
            const int m = idx / HP, k = idx - m * HP;
            const int unit = u0 + (m >> 2), gate = m & 3;
            float v = 0.f;
            if (k < H && unit < H && (!GRU || gate < 3)) v = W[(size_t)(gate * H + unit) * H + k];
            Ws[idx] = v;
        }
        if (tid == 0) {
            abort_flag[0] = 0;
            abort_flag[1] = 0;   // 'canaries of step s seen' word (poll_mode bit1)
        }
    }
    __syncthreads();

    // ---- static cell-lane ownership
    int c_unit[CPT], c_b[CPT], c_xoff[CPT];
    bool c_valid[CPT];
    float c_state[CPT];
    int c_cl[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        // Cells are numbered idx = ((mt*NT + nt)*16 + n)*4 + q so that 4 ADJACENT lanes hold the 4
        // units (q) of one batch row: their h values are gathered with 3 DPP shuffles and leave
        // as ONE 16-B write-through store per row (8 rows = a full 128-B line per wave and step;
        // per-lane 4-B stores were 64 partial-line transactions per workgroup and step).
        const int lw = lane + 64 * i;             // index inside this wave's share
        const int idx = wave * CW + (lw < CW ? lw : 0);
        const int q = idx & 3, n = (idx >> 2) & 15, blk = idx >> 6;
        const int nt = blk % NT, mt = blk / NT;
        c_cl[i] = blk * RED_PITCH + red_slot(q * 16 + n);   // where the MFMA left this cell's partial sums
        c_unit[i] = u0 + mt * 4 + q;
        const int bl = nt * 16 + n;
        c_b[i] = b0 + bl;
        c_valid[i] = (lw < CW) && (bl < nb) && (c_unit[i] < H);
        // exchange layout [k4 = unit/4][nt][row n][4 units]: float offset of this row's 4-unit group
        c_xoff[i] = ((((u0 >> 2) + mt) * NT + nt) * 16 + n) * 4;
        c_state[i] = 0.f;
    }

    const int k_lo = wave * KGW * 16;  // this wave's K slice
    const int m16 = lane & 15, q4 = lane >> 4;
    // one step's exchange region: fragment-ordered data, then 4 canary words per producer
    const size_t data_floats = (size_t)p.kgp * NT * 256;
    const size_t step_floats = data_floats + (size_t)p.canw;
    float *xgroup = p.X + (size_t)group * p.T * step_floats;

    // fragment load offsets (bytes) inside one step's region; OOB offset -> hardware returns 0
    unsigned xoff[NT][KGW];
#pragma unroll
    for (int kg = 0; kg < KGW; ++kg) {
        const int k = k_lo + kg * 16 + 4 * q4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const bool v = (k < H) && (nt * 16 + m16 < nb);
            // [k4][nt][row][4]: lane (row m16, k-quad q4) of k-group kg reads block k4 = k/4
            xoff[nt][kg] =
                v ? (unsigned)((((((k_lo >> 2) + kg * 4 + q4) * NT + nt) * 16 + m16) * 4) * 4)
                  : 0x7ffffff0u;
        }
    }

    // canaries: every wave of producer workgroup j publishes word [4*j + wave] after its exchange
    // stores; this wave polls the words of the producers of ITS K slice (<= 64 words).
    const int k_hi = min(H, k_lo + KGW * 16);
    const int wg_lo = k_lo < H ? k_lo / p.U : 0;
    const int wg_cnt = k_lo < H ? (k_hi - 1) / p.U - wg_lo + 1 : 0;
    const int can_cnt = 4 * wg_cnt;

    // pre-activations of the first step
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

    for (int s = 0; s < p.T; ++s) {
        const int t = dir == 0 ? s : p.T - 1 - s;

        f32x4 acc[MT][NT][ACC];
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int h2 = 0; h2 < ACC; ++h2) acc[a][b][h2] = f32x4{0.f, 0.f, 0.f, 0.f};

        REC_STAMP(0);
        if (s > 0 && k_lo < H) {
            // h_{s-1}: B-operand fragments from the exchange buffer; re-load until sentinel-free
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(xgroup + (size_t)(s - 1) * step_floats), 0, (int)(step_floats * 4),
                0x00020000);
            f32x4 bf[NT][KGW];
            unsigned spins = 0;
            unsigned long long t0 = 0;
            bool ok = true;
            // cheap probe first (canary words, sc1 polls), bulk fragments after.
            // poll_mode bit0: pipelined polls; bit1: only wave 0 polls (all producers of the group)
            // and releases the other waves through an LDS word (4x fewer global pollers).
            {
                // Do not poll straight away: nothing can arrive sooner than one memory round trip
                // after this workgroup's own stores (all workgroups of a group run in lockstep), and
                // early polls only queue read traffic on the very lines the producers are writing
                // through (measured: 1024 idle cycles here cut the wait from 3.1k to 2.4k cycles).
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
            REC_STAMP(7);
            // FAST PATH: bulk fragments with PLAIN loads (the 32 CUs of an XCD that need the same
            // lines share one fabric fetch through their L2), all issued up front; MFMAs consume
            // them as they land and the sentinel checks ride along on the VALU.
            bool bad = false;
            if (ok) {
                // A wave that is issuing loads cannot issue MFMAs (in-order issue, ~47 cycles per 1-KiB
                // load): all KGW*NT loads up front kept the matrix pipe idle for 1.5k cycles at H = 1024
                // although the first fragment lands after ~1.2k.  Issue PF k-groups, then one k-group of
                // loads after each k-group of MFMAs.
                constexpr int PF = KGW >= 8 ? 4 : KGW;
#pragma unroll
                for (int kg = 0; kg < PF; ++kg)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, xoff[nt][kg], 0, 0);
                        bf[nt][kg] = __builtin_bit_cast(f32x4, v);
                    }
                __builtin_amdgcn_sched_barrier(0);
                REC_STAMP(1);
#pragma unroll
                for (int kg = 0; kg < KGW; ++kg) {
                    fwd_mfma_kgroup<MT, NT, ACC, KGW>(acc, bf, Ws, HP, m16, k_lo, kg, q4);
                    if (kg + PF < KGW) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, xoff[nt][kg + PF], 0, 0);
                            bf[nt][kg + PF] = __builtin_bit_cast(f32x4, v);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // the sentinel is a NaN: any unwritten word poisons its accumulator column, so the
                // check is 2*MT*NT compares after the MFMAs instead of VALU work between them
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        bad |= any_nan(acc_sum<ACC>(acc[mt][nt]));
            }
            // SLOW PATH (rare): a fragment was read before its producer's store was visible (or a
            // stale line was cached) -> redo the step from L1/L2-bypassing reloads, verified first
            if (ok && __any(bad)) {
                for (;;) {
#pragma unroll
                    for (int kg = 0; kg < KGW; ++kg)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, xoff[nt][kg], 0, 16);
                            bf[nt][kg] = __builtin_bit_cast(f32x4, v);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                    bad = false;
#pragma unroll
                    for (int kg = 0; kg < KGW; ++kg)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) bad |= has_sentinel(bf[nt][kg]);
                    if (!__any(bad)) break;
                    if (!spin_ok(spins, t0, p.err, lane)) {
                        ok = false;
                        break;
                    }
                }
                if (ok) {
#pragma unroll
                    for (int a = 0; a < MT; ++a)
#pragma unroll
                        for (int b = 0; b < NT; ++b)
#pragma unroll
                            for (int h2 = 0; h2 < ACC; ++h2) acc[a][b][h2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kg = 0; kg < KGW; ++kg)
                        fwd_mfma_kgroup<MT, NT, ACC, KGW>(acc, bf, Ws, HP, m16, k_lo, kg, q4);
                }
            }
            if (!ok && lane == 0) *abort_flag = 1;
        }
        REC_STAMP(2);
        f32x4 *redw = red + (DB ? (s & 1) : 0) * 4 * CLP;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                redw[((wave * MT + mt) * NT + nt) * RED_PITCH + red_slot(lane)] = acc_sum<ACC>(acc[mt][nt]);
        REC_STAMP(3);
        __syncthreads();  // the only barrier per step: partial sums visible
        if (*abort_flag) break;
        REC_STAMP(4);

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
                    gi[i] = fast_sigmoid(gpre[i][0] + sum[0]);            // r
                    gf[i] = fast_sigmoid(gpre[i][1] + sum[1]);            // z
                    go[i] = sum[2] + gpre[i][3];                          // hn = W_hn h + b_hn
                    gg[i] = fast_tanh(gpre[i][2] + gi[i] * go[i]);        // n
                    hv[i] = (1.f - gf[i]) * gg[i] + gf[i] * c_state[i];   // c_state carries h_{t-1}
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
        // the exchange payload for step s+1 (data == flag): lanes 4r..4r+3 hold the 4 units of one
        // row -> gather into the q == 0 lane, ONE 16-B write-through (sc1) store per row
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            f32x4 h4;
            // quad_perm broadcasts (v_mov_b32_dpp, 1 issue slot each; __shfl_down compiles to
            // ds_bpermute_b32, an LDS round trip on the serial chain): lane 0 of a quad collects 1..3
            const int hb = __builtin_bit_cast(int, hv[i]);
            h4[0] = hv[i];
            h4[1] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(hb, 0x55, 0xf, 0xf, true));
            h4[2] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(hb, 0xAA, 0xf, 0xf, true));
            h4[3] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(hb, 0xFF, 0xf, 0xf, true));
            if (c_valid[i] && (lane & 3) == 0)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h4), xrs,
                                                       (unsigned)(c_xoff[i] * 4), 0, 16);
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
        // canary: issued after this wave's exchange stores (ordering is NOT relied upon: consumers
        // verify every data word against the sentinel)
        if (lane == 0)
            __hip_atomic_store(reinterpret_cast<unsigned *>(xstep + data_floats) + 4 * wg + wave,
                               (unsigned)(s + 1), RLX_AGENT);
        REC_STAMP(5);
        // saved-for-backward tensors + next step's pre-activations (off the critical path)
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
        REC_STAMP(6);
        if (!DB) __syncthreads();  // single-buffered partial sums (LDS-tight shapes, e.g. H=1024)
    }
}

template <int MT, int NT, int KGW, bool DB, bool GRU>
int launch_fwd(const RecFwdArgs &a, int grid, size_t lds, hipStream_t s) {
    auto kern = lstm_rec_fwd_kernel<MT, NT, KGW, DB, GRU>;
    ASRK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

template <int MT, int NT, bool GRU>
int launch_fwd_k(const RecFwdArgs &a, int KGW, int db, int grid, size_t lds, hipStream_t s) {
    switch (KGW) {
        case 4: return db ? launch_fwd<MT, NT, 4, true, GRU>(a, grid, lds, s)
                          : launch_fwd<MT, NT, 4, false, GRU>(a, grid, lds, s);
        case 8: return db ? launch_fwd<MT, NT, 8, true, GRU>(a, grid, lds, s)
                          : launch_fwd<MT, NT, 8, false, GRU>(a, grid, lds, s);
        case 16: return db ? launch_fwd<MT, NT, 16, true, GRU>(a, grid, lds, s)
                           : launch_fwd<MT, NT, 16, false, GRU>(a, grid, lds, s);
    }
    return ASRK_ESHAPE;
}

template <bool GRU>
int launch_fwd_plan(const RecFwdArgs &a, const FwdPlan &pl, int grid, hipStream_t s) {
    if (pl.MT == 1 && pl.NT == 1) return launch_fwd_k<1, 1, GRU>(a, pl.KGW, pl.db, grid, pl.lds, s);
    if (pl.MT == 1 && pl.NT == 2) return launch_fwd_k<1, 2, GRU>(a, pl.KGW, pl.db, grid, pl.lds, s);
    if (pl.MT == 2 && pl.NT == 1) return launch_fwd_k<2, 1, GRU>(a, pl.KGW, pl.db, grid, pl.lds, s);
    if (pl.MT == 2 && pl.NT == 2) return launch_fwd_k<2, 2, GRU>(a, pl.KGW, pl.db, grid, pl.lds, s);
    if (pl.MT == 1 && pl.NT == 4) return launch_fwd_k<1, 4, GRU>(a, pl.KGW, pl.db, grid, pl.lds, s);
    if (pl.MT == 4 && pl.NT == 1) return launch_fwd_k<4, 1, GRU>(a, pl.KGW, pl.db, grid, pl.lds, s);
    return ASRK_ESHAPE;
}

}  // namespace

int launch_fwd_f32(bool gru, const RecFwdArgs &a, const FwdPlan &pl, int grid, hipStream_t s) {
    return gru ? launch_fwd_plan<true>(a, pl, grid, s) : launch_fwd_plan<false>(a, pl, grid, s);
}

}  // namespace asrk_rec
