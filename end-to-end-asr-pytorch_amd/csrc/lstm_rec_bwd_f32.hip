// Persistent LSTM / GRU backward recurrence (BPTT), exact-f32 MFMA form (design: lstm_rec.hip).
#include "lstm_rec_common.h"

namespace asrk_rec {
namespace {

// Fragment loads of one chunk (= one ring's worth of k-groups: primes the ring on the fast path,
// whole-chunk reloads on the slow path).  Per-lane byte offsets `voff[nt]` are loop invariant (an
// out-of-bounds value for padded batch rows -> the hardware returns 0); the k-group / gate part of
// the address is wave-uniform and goes into the scalar offset, so a load costs no VALU work.  What
// a one-KiB load does cost is ~60-90 cycles of issue with four waves loading (vector-memory path,
// 64 B/clk per CU) -- see the ring in the kernel.  `voff_tail` covers the last k-group when
// H % 16 != 0.
template <int NT, int CH, int AUX>
__device__ __forceinline__ void bwd_load_chunk(f32x4 (&bf)[NT][CH], __amdgpu_buffer_rsrc_t rs,
                                               int kg0, int kgs, const unsigned (&voff)[NT],
                                               const unsigned (&voff_tail)[NT], bool ragged_k,
                                               int gate_base) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int kg = kg0 + c;
        const bool tail = ragged_k && kg == kgs - 1;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            // padded k-groups (kg >= kgs) read far out of bounds -> zeros
            const unsigned soff = kg < kgs ? (unsigned)((gate_base + (kg * NT + nt) * 256) * 4)
                                           : 0x7ff00000u;
            u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rs, tail ? voff_tail[nt] : voff[nt], soff,
                                                            AUX);
            bf[nt][c] = __builtin_bit_cast(f32x4, x);
        }
    }
    __builtin_amdgcn_sched_barrier(0);  // the whole chunk is issued before anything else moves
}

template <int NT, int CH>
__device__ __forceinline__ bool bwd_chunk_bad(const f32x4 (&bf)[NT][CH]) {
    bool bad = false;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bad |= has_sentinel(bf[nt][c]);
    return __any(bad);
}

template <int NT, int CH, int ACC>
__device__ __forceinline__ void bwd_mfma_chunk(f32x4 (&acc)[NT][ACC], const f32x4 (&bf)[NT][CH],
                                               const float *wrow, int kg0, int q4) {
    // Straight-line: the LDS rows are zero-padded to whole chunks (rows >= UB point at a zero row)
    // and out-of-range fragments were loaded as zeros, so the ds_reads pipeline ahead of the MFMAs
    // and NO VALU work sits between them (mask multiplies + per-fragment sentinel compares cost
    // 2.4 us/step).  A sentinel (a NaN) in any fragment poisons the accumulator column instead.
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(wrow + (kg0 + c) * 16 + 4 * q4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt][j % ACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bf[nt][c][j],
                                                                        acc[nt][j % ACC], 0, 0, 0);
    }
}

// RK > 0: the first RK k-groups of every wave's W_hh^T slice live in VGPRs (4 floats per lane and
// k-group), only the rest in LDS.  At H = 1024 the whole slice of 16 units is 256 KiB: with 8 units
// per workgroup (all that fits in LDS) the 16-row MFMA tile is half padding and every workgroup
// contracts against all of dG [B,4H]; 16 units x 16 batch rows per workgroup (RK = 32: half of the
// slice in 128 VGPRs per lane) is a full tile, half the MFMAs and half the fragment bytes per step.
// GRU = true: BPTT of torch.nn.GRU in the same layout (see the forward kernel); p.C must be Y (h_{t-1} is
// read from it), the exchanged hidden-side gradients are (dr, dz, dn r, 0), the stored ones (dr, dz, dn, dn r).
template <int NT, int RK, bool GRU>
__global__ __launch_bounds__(256) void lstm_rec_bwd_kernel(RecBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CH = bwd_ring_kgroups(NT, RK);  // k-groups in the fragment ring
    static_assert(RK % CH == 0, "register-resident k-groups come in whole ring rounds");
    constexpr int ACC = NT >= 2 ? 2 : 4;     // accumulator chains per output tile (see acc_sum)
    // `wave` must be provably uniform: it feeds scalar operands (buffer-load soffset) and branch
    // conditions; a VGPR there costs a readfirstlane waterfall loop around EVERY load.
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ngroups = p.ndir * p.nbg;
    const int group = blockIdx.x % ngroups, wg = blockIdx.x / ngroups;
    const int dir = p.dir0 + group % p.ndir, bg = p.bg0 + group / p.ndir;
    const int u0 = wg * p.UB, b0 = bg * p.BG;
    const int nb = min(p.BG, p.B - b0);
    const int H = p.H, HPb = p.HPb, KP = p.KP, UB = p.UB;

    // [UB][KP]: Wt[m][gate*HPb + (j - 16*RK)] = W_hh[gate*H + j][u0 + m] for j >= 16*RK (HPb counts
    // the LDS-resident part only)
    float *Wt = smem;
    float *zrow = smem + UB * KP;  // [HPb] zeros: the A rows >= UB of the 16-row MFMA tile
    f32x4 *red = reinterpret_cast<f32x4 *>(zrow + HPb);  // [2 parity][4 waves][NT][RED_PITCH]
    int *abort_flag = reinterpret_cast<int *>(red + 2 * 4 * NT * RED_PITCH);

    {
        for (int idx = tid; idx < UB * KP + HPb; idx += 256) Wt[idx] = 0.f;   // incl. zrow
        __syncthreads();
        const float *W = p.whh[dir];
        const int total = 4 * H * UB;
        for (int idx = tid; idx < total; idx += 256) {
            const int m = idx % UB, rj = idx / UB;  // rj = gate*H + j
            const int gate = rj / H, j = rj - gate * H;
            if (u0 + m < H && j >= 16 * RK && (!GRU || gate < 3))
                Wt[m * KP + gate * HPb + j - 16 * RK] = W[(size_t)rj * H + u0 + m];
        }
        if (tid == 0) *abort_flag = 0;
    }
    __syncthreads();
    // register-resident A fragments: lane (row m16, k-quad q4) of wave w holds
    // W_hh[w*H + kg*16 + 4*q4 + 0..3][u0 + m16] for kg < RK
    f32x4 areg[RK > 0 ? RK : 1];
    if (RK > 0) {
        const float *W = p.whh[dir];
        const int m16r = tid & 15, q4r = (tid & 63) >> 4;
#pragma unroll
        for (int kg = 0; kg < RK; ++kg) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = kg * 16 + 4 * q4r + e;
                if (m16r < UB && u0 + m16r < H && j < H && (!GRU || wave < 3))
                    v[e] = W[((size_t)wave * H + j) * H + u0 + m16r];
            }
            areg[kg] = v;
        }
    }

    // cells owned by this thread: ci = tid + 256*i -> (unit = ci%16, batch = ci/16)
    int c_unit[NT], c_b[NT], c_red[NT], c_xoff[NT];
    bool c_valid[NT];
    float dc_carry[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int ci = tid + 256 * i;
        const int ul = ci & 15, bl = ci >> 4;
        c_unit[i] = u0 + ul;
        c_b[i] = b0 + bl;
        c_valid[i] = ul < UB && bl < nb && c_unit[i] < H;
        // reduction buffer address of (unit ul, batch bl): f32x4 index * 4 + component ul&3
        c_red[i] = ((bl >> 4) * RED_PITCH + red_slot((ul >> 2) * 16 + (bl & 15))) * 4 + (ul & 3);
        // exchange offset inside one gate's region: block (kg = unit/16, nt = bl/16)
        c_xoff[i] = (((c_unit[i] >> 4) * NT + (bl >> 4)) * 16 + (bl & 15)) * 16 + (c_unit[i] & 15);
        dc_carry[i] = 0.f;
    }
    float dbsum[NT][4];   // bias gradient: this thread's cells summed over time
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) dbsum[i][r] = 0.f;

    const int kgs = p.kgp;  // k-groups per gate (wave w <-> gate w)
    const int nb_ld = (p.dbg_steps == -1) ? 0 : nb;  // debug: -1 turns every fragment load into an OOB zero
    const int nch = (kgs + CH - 1) / CH;
    const int m16 = lane & 15, q4 = lane >> 4;
    const float *wrow = m16 < UB ? Wt + m16 * KP + wave * HPb : zrow;   // rows >= UB read zeros
    // loop-invariant per-lane fragment offsets (bytes); padded batch rows are out of bounds
    const bool ragged_k = (H & 15) != 0;
    unsigned voff[NT], voff_tail[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const bool rv = nt * 16 + m16 < nb_ld;
        voff[nt] = rv ? (unsigned)((m16 * 16 + 4 * q4) * 4) : 0x7ff00000u;
        voff_tail[nt] = (rv && (kgs - 1) * 16 + 4 * q4 < H) ? voff[nt] : 0x7ff00000u;
    }
    const size_t gate_floats = (size_t)kgs * NT * 256;
    const size_t data_floats = 4 * gate_floats;
    const size_t step_floats = data_floats + (size_t)p.canw;
    float *xgroup = p.X + (size_t)group * p.T * step_floats;
    const int gate_base = (int)(wave * gate_floats);

    for (int s = 0; s < p.T; ++s) {
        // dir 0 ran t = 0..T-1 forward -> backward walks T-1..0 and needs dG of t+1;
        // dir 1 ran T-1..0 -> backward walks 0..T-1 and needs dG of t-1.
        const int t = dir == 0 ? p.T - 1 - s : s;
        const int tp = dir == 0 ? t - 1 : t + 1;  // step that produced c_{prev} of t

        float vi[NT], vf[NT], vg[NT], vo[NT], vc[NT], vcp[NT], vdy[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            vi[i] = vf[i] = vg[i] = vo[i] = vc[i] = vcp[i] = vdy[i] = 0.f;
            if (c_valid[i]) {
                const size_t row = (size_t)t * p.B + c_b[i];
                const float *g = p.G + row * p.ldg + dir * 4 * H + c_unit[i];
                vi[i] = g[0];
                vf[i] = g[(size_t)H];
                vg[i] = g[(size_t)2 * H];
                vo[i] = g[(size_t)3 * H];
                vc[i] = p.C[row * p.ldy + dir * H + c_unit[i]];
                if (p.pyr_mode == 0) {
                    vdy[i] = p.dY[row * p.ldy + dir * H + c_unit[i]];
                } else {   // gradient arrives in the next layer's (time-reduced) input layout
                    const int r = p.pyr_rate, tq = t / r, tr = t - tq * r;
                    if (p.pyr_mode == 1 ? tq < p.T / r : tr == 0) {
                        const size_t ld2 = p.pyr_mode == 1 ? (size_t)r * p.ldy : (size_t)p.ldy;
                        const size_t off = p.pyr_mode == 1 ? (size_t)tr * p.ldy : 0;
                        vdy[i] = p.dY[((size_t)tq * p.B + c_b[i]) * ld2 + off + dir * H + c_unit[i]];
                    }
                }
                if (tp >= 0 && tp < p.T)
                    vcp[i] = p.C[((size_t)tp * p.B + c_b[i]) * p.ldy + dir * H + c_unit[i]];
            }
        }

        f32x4 acc[NT][ACC];
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int h2 = 0; h2 < ACC; ++h2) acc[b][h2] = f32x4{0.f, 0.f, 0.f, 0.f};

        REC_STAMP(0);
        if (s > 0) {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(xgroup + (size_t)(s - 1) * step_floats), 0, (int)(step_floats * 4),
                0x00020000);
            f32x4 bf0[NT][CH];
            unsigned spins = 0;
            unsigned long long t0 = 0;
            bool ok = true;
            // cheap probe first: 4 canary words per producer workgroup of this group (the optional
            // pre-poll pause of the forward kernel does not pay here: the loop-head loads above
            // already wait out the store acknowledgements, ~1.6k cycles)
            for (int z = (p.poll_mode >> 8) & 0xff; z > 0; z -= 8) __builtin_amdgcn_s_sleep(8);
            ok = wait_canaries(reinterpret_cast<const unsigned *>(
                                   xgroup + (size_t)(s - 1) * step_floats + data_floats),
                               4 * p.nwg, p.err, lane, p.poll_mode);
            REC_STAMP(7);
            // FAST PATH: a ring of CH k-groups of fragments in flight, refilled ONE k-group at a time
            // right after the MFMAs that consumed it.  The CU's vector-memory path moves ~64 B/clk,
            // i.e. one 1-KiB fragment load per wave every ~90 cycles with four waves loading, and a
            // wave that is stuck issuing loads cannot issue MFMAs: issuing a step's loads in bursts
            // of 16-32 left the matrix pipe idle for ~3k cycles per step.  One load per 4*NT MFMAs
            // (128*NT cycles) keeps both pipes busy.  Straight-line body, no retry loops inside, so
            // the compiler's counted vmcnt waits stay exact; sentinel checks are deferred (NaN).
            // Plain loads: the CUs of an XCD share lines in L2.
            bool bad = false;
            if (ok) {
                bwd_load_chunk<NT, CH, 0>(bf0, rs, 0, kgs, voff, voff_tail, ragged_k, gate_base);
                REC_STAMP(1);
                // The matrix pipe needs 128*NT cycles per k-group and the wave issues in order, so
                // every extra instruction between the MFMAs shows (measured: 15 instructions per
                // k-group -- range / tail selects for the refill address -- ran at 184 cycles, the
                // bare pattern of tools/mfma_ring.hip at 128).  Hence several loops: while the refill
                // is known to be a full, in-range k-group its scalar offset just advances; the
                // general form (selects, out-of-range -> zeros) only covers the last rounds.
                unsigned run = (unsigned)((gate_base + CH * NT * 256) * 4);  // offset of k-group kg0+CH
                if (RK > 0) {
                    // register-resident k-groups (the host guarantees RK + CH <= kgs, no ragged tail
                    // among the refills): A operand straight from VGPRs, no LDS read at all
#pragma unroll
                    for (int kg0 = 0; kg0 < RK; kg0 += CH) {
#pragma unroll
                        for (int r = 0; r < CH; ++r) {
                            const f32x4 ar = areg[kg0 + r];
#pragma unroll
                            for (int j = 0; j < 4; ++j)
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt)
                                    acc[nt][j % ACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                        ar[j], bf0[nt][r][j], acc[nt][j % ACC], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(
                                    rs, voff[nt], run + (unsigned)((r * NT + nt) * 1024), 0);
                                bf0[nt][r] = __builtin_bit_cast(f32x4, x);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        run += CH * NT * 1024;
                    }
                }
                // LDS-resident k-groups: LDS index kg - RK
                f32x4 a = *reinterpret_cast<const f32x4 *>(wrow + 4 * q4);
                const int kg_plain = ragged_k ? kgs - 1 : kgs;  // k-groups below this need no selects
                int kg0 = RK;
                for (; kg0 + 2 * CH <= kg_plain; kg0 += CH, run += CH * NT * 1024) {
#pragma unroll
                    for (int r = 0; r < CH; ++r) {
                        const f32x4 an =
                            *reinterpret_cast<const f32x4 *>(wrow + (kg0 - RK + r + 1) * 16 + 4 * q4);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[nt][j % ACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                    a[j], bf0[nt][r][j], acc[nt][j % ACC], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(
                                rs, voff[nt], run + (unsigned)((r * NT + nt) * 1024), 0);
                            bf0[nt][r] = __builtin_bit_cast(f32x4, x);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        a = an;
                    }
                }
                for (; kg0 < nch * CH; kg0 += CH) {
                    const bool refill = kg0 + CH < kgs;  // the last round(s) have nothing left to fetch
#pragma unroll
                    for (int r = 0; r < CH; ++r) {
                        // LDS rows are padded to whole chunks (+8 floats), so the look-ahead stays in bounds
                        const f32x4 an =
                            *reinterpret_cast<const f32x4 *>(wrow + (kg0 - RK + r + 1) * 16 + 4 * q4);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[nt][j % ACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                    a[j], bf0[nt][r][j], acc[nt][j % ACC], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if (refill) {   // slot r <- k-group kg0+CH+r (past the end: OOB -> zeros)
                            const int kg = kg0 + CH + r;
                            const bool tail = ragged_k && kg == kgs - 1;
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) {
                                const unsigned soff =
                                    kg < kgs ? (unsigned)((gate_base + (kg * NT + nt) * 256) * 4)
                                             : 0x7ff00000u;
                                u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(
                                    rs, tail ? voff_tail[nt] : voff[nt], soff, 0);
                                bf0[nt][r] = __builtin_bit_cast(f32x4, x);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        a = an;
                    }
                }
            }
            // a sentinel anywhere shows up as NaN in the accumulators
#pragma unroll
            for (int b = 0; b < NT; ++b) bad |= any_nan(acc_sum<ACC>(acc[b]));
            // SLOW PATH (rare: a fragment was read before its producer's store became visible, a
            // stale line was cached, or the data itself is NaN): start over with L1/L2-bypassing
            // reloads, verified against the sentinel bit pattern before use
            if (ok && __any(bad)) {
#pragma unroll
                for (int b = 0; b < NT; ++b)
#pragma unroll
                    for (int h2 = 0; h2 < ACC; ++h2) acc[b][h2] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (RK > 0) {
#pragma unroll
                    for (int c = 0; c < RK / CH; ++c) {
                        while (ok) {
                            bwd_load_chunk<NT, CH, 16>(bf0, rs, c * CH, kgs, voff, voff_tail, ragged_k, gate_base);
                            if (!bwd_chunk_bad<NT, CH>(bf0)) break;
                            if (!spin_ok(spins, t0, p.err, lane)) ok = false;
                        }
                        if (ok) {
#pragma unroll
                            for (int r = 0; r < CH; ++r)
#pragma unroll
                                for (int j = 0; j < 4; ++j)
#pragma unroll
                                    for (int nt = 0; nt < NT; ++nt)
                                        acc[nt][j % ACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                            areg[c * CH + r][j], bf0[nt][r][j], acc[nt][j % ACC], 0, 0, 0);
                        }
                    }
                }
                for (int c = RK / CH; c < nch && ok; ++c) {
                    for (;;) {
                        bwd_load_chunk<NT, CH, 16>(bf0, rs, c * CH, kgs, voff, voff_tail, ragged_k, gate_base);
                        if (!bwd_chunk_bad<NT, CH>(bf0)) break;
                        if (!spin_ok(spins, t0, p.err, lane)) {
                            ok = false;
                            break;
                        }
                    }
                    if (ok) bwd_mfma_chunk<NT, CH, ACC>(acc, bf0, wrow, c * CH - RK, q4);
                }
            }
            if (!ok && lane == 0) *abort_flag = 1;
        }
        REC_STAMP(2);
        f32x4 *redw = red + (s & 1) * 4 * NT * RED_PITCH;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            redw[(wave * NT + nt) * RED_PITCH + red_slot(lane)] = acc_sum<ACC>(acc[nt]);
        REC_STAMP(3);
        __syncthreads();  // the only barrier per step
        if (*abort_flag) break;
        REC_STAMP(4);

        const float *redf = reinterpret_cast<const float *>(redw);
        float *xstep = xgroup + (size_t)s * step_floats;
        float dgs[NT][4];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            if (c_valid[i]) {
                float rec = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) rec += redf[w * NT * RED_PITCH * 4 + c_red[i]];
                float xg[4];   // what the neighbours' next step contracts with W_hh (exchange payload)
                if (GRU) {
                    const float dh = vdy[i] + rec + dc_carry[i];          // carry = dh_{next} z_{next}
                    const float r = vi[i], z = vf[i], n = vg[i], hn = vo[i], hp = vcp[i];
                    const float dn = dh * (1.f - z) * (1.f - n * n);
                    const float dz = dh * (hp - n) * z * (1.f - z);
                    const float dr = dn * hn * r * (1.f - r);
                    dc_carry[i] = dh * z;
                    dgs[i][0] = dr; dgs[i][1] = dz; dgs[i][2] = dn; dgs[i][3] = dn * r;
                    xg[0] = dr; xg[1] = dz; xg[2] = dn * r; xg[3] = 0.f;
                } else {
                    const float dh = vdy[i] + rec;
                    const float tc = fast_tanh(vc[i]);
                    const float dcell = dh * vo[i] * (1.f - tc * tc) + dc_carry[i];
                    dc_carry[i] = dcell * vf[i];
                    dgs[i][0] = dcell * vg[i] * vi[i] * (1.f - vi[i]);
                    dgs[i][1] = dcell * vcp[i] * vf[i] * (1.f - vf[i]);
                    dgs[i][2] = dcell * vi[i] * (1.f - vg[i] * vg[i]);
                    dgs[i][3] = dh * tc * vo[i] * (1.f - vo[i]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) xg[r] = dgs[i][r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    __hip_atomic_store(xstep + (size_t)r * gate_floats + c_xoff[i], xg[r], RLX_AGENT);
                    dbsum[i][r] += dgs[i][r];
                }
            }
        }
        if (lane == 0)
            __hip_atomic_store(reinterpret_cast<unsigned *>(xstep + data_floats) + 4 * wg + wave,
                               (unsigned)(s + 1), RLX_AGENT);
        REC_STAMP(5);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            if (c_valid[i]) {
                float *g = p.G + ((size_t)t * p.B + c_b[i]) * p.ldg + dir * 4 * H + c_unit[i];
#pragma unroll
                for (int r = 0; r < 4; ++r) g[(size_t)r * H] = dgs[i][r];
            }
        }
        if (p.rearm && s >= 2)
            rearm_region(xgroup + (size_t)(s - 2) * step_floats, step_floats, wg, p.nwg, tid, (int)blockDim.x);
        REC_STAMP(6);
    }
    // bias gradient db[dir][gate*H + unit] = sum over time and batch of dG: the per-thread sums over
    // time meet in LDS, one thread per (unit, gate) adds the batch rows of this group; the (<= nbg)
    // batch groups of a direction combine with atomics into the zero-initialised output
    if (p.db && !*abort_flag) {
        __syncthreads();
        float *sdb = reinterpret_cast<float *>(red);   // [NT*256 cells][4]  (<= the partial-sum buffer)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) sdb[(tid + 256 * i) * 4 + r] = c_valid[i] ? dbsum[i][r] : 0.f;
        __syncthreads();
        if (tid < 64) {
            const int ul = tid & 15, r = tid >> 4;
            if (ul < UB && u0 + ul < H) {
                float acc = 0.f;
                for (int bl = 0; bl < 16 * NT; ++bl) acc += sdb[(bl * 16 + ul) * 4 + r];
                unsafeAtomicAdd(p.db + (size_t)dir * 4 * H + (size_t)r * H + u0 + ul, acc);
            }
        }
    }
}

template <int NT, int RK, bool GRU>
int launch_bwd(const RecBwdArgs &a, int grid, size_t lds, hipStream_t s) {
    auto kern = lstm_rec_bwd_kernel<NT, RK, GRU>;
    ASRK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

template <bool GRU>
int launch_bwd_plan(const RecBwdArgs &a, const BwdPlan &pl, int grid, hipStream_t s) {
    if (pl.NT == 1 && pl.RK == 32) return launch_bwd<1, 32, GRU>(a, grid, pl.lds, s);
    if (pl.NT == 1) return launch_bwd<1, 0, GRU>(a, grid, pl.lds, s);
    if (pl.NT == 2) return launch_bwd<2, 0, GRU>(a, grid, pl.lds, s);
    if (pl.NT == 4) return launch_bwd<4, 0, GRU>(a, grid, pl.lds, s);
    return ASRK_ESHAPE;
}

}  // namespace

int launch_bwd_f32(bool gru, const RecBwdArgs &a, const BwdPlan &pl, int grid, hipStream_t s) {
    return gru ? launch_bwd_plan<true>(a, pl, grid, s) : launch_bwd_plan<false>(a, pl, grid, s);
}

}  // namespace asrk_rec
