// Persistent (one launch per layer, all T steps in-kernel) LSTM recurrence for gfx950.
//
// Replaces the time loop inside ATen's lstm that the reference reaches through
// nn.LSTM(...)(input_x) (src/module.py:112-113,131): per step g = G_t + h_{t-1} W_hh^T,
// i,f,o = sigmoid, g = tanh, c = f*c + i*g, h = o*tanh(c); both directions; zero initial state;
// no sequence packing (module.py:129-132 runs over the padded frames too).
//
// Forward design
//  * W_hh is STATIONARY on chip for the whole sequence: the grid is (direction x batch-group x
//    hidden-unit slice); each workgroup (4 waves, 1 per SIMD) keeps the 4 gate rows of its U
//    hidden units in LDS ([16*MT][H] f32), c_t of its units in registers, and never re-reads
//    weights from HBM (non-stationary W_hh would be T * 4H*H*4 B of traffic: 100 GB at cfg3).
//  * exact-f32 MFMA 16x16x4: M = gate rows (unit-major, gate-minor, so one lane ends up with
//    i,f,g,o of ONE (unit,batch) cell -> the cell update is lane-local), N = batch, K = H split
//    over the 4 waves; partial sums meet in LDS (double-buffered by step parity, one barrier/step).
//  * the only inter-workgroup traffic per step is h_{t-1} [B,H].  It travels through an EXCHANGE
//    buffer laid out in MFMA-fragment order ([step][k-group][batch 16][k 16] = 1 KiB blocks, so
//    each wave load instruction reads 8 full 128-B lines) and the DATA IS THE FLAG: the buffer is
//    pre-filled with a NaN sentinel (0xFFFFFFFF, never produced by |h| < 1), producers store h
//    write-through (sc1) and consumers simply re-issue their sc1 (L1-bypassing) fragment loads
//    until no sentinel is left.  No flag word, no store drain, no second barrier, no acquire
//    fence on the critical path; no dispatch-order or XCD-placement assumption; every spin is
//    wall-clock bounded and reports ASRK_ETIMEOUT.  (v1 used per-workgroup epoch flags: its
//    measured timeline was 12k cycles/step of which ~5k were drain -> flag -> poll.)
//
// Backward design (BPTT): same stationarity with W_hh^T slices ([UB units][4H]) in LDS; the
// per-step exchange is dG_{t+1} [B,4H] (pre-activation gradients) in the same fragment-ordered,
// sentinel-tagged form; wave w contracts gate w's H rows.  dW_hh/dW_ih/dX/db are plain
// GEMMs/column sums on the finished dG (ops layer).
#include "lstm_rec_common.h"

extern "C" int asrk_cu_count_(void);
extern "C" size_t asrk_split_panel_stride_(int rows, int K);   // gemm_split.hip: bytes between 64-row blocks of a panel

using namespace asrk_rec;

namespace {

// sentinel fill of the exchange buffer (0xFFFFFFFF words): 16-B stores from every CU; it sits in front of every
// recurrence launch.  MODE 2 (default): 2048 workgroups, each filling contiguous 16-KiB runs with plain stores -
// 133 us for the two buffers of a T=800, H=1024 layer (3.6 TB/s) against 149 (plain, element-strided grid) and 195
// (nontemporal stores, the round-2 kernel; the runtime's memset: ~2 TB/s).  Since round 5 only the FIRST launch on a
// pooled buffer (and plans of several launches) pays it: ASRK_REC_REARM.
__global__ __launch_bounds__(256) void sentinel_fill_kernel(u32x4 *__restrict__ p, size_t n16) {
    const u32x4 v = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    // every workgroup fills contiguous 16-KiB runs: 4 x (256 lanes x 16 B)
    for (size_t base = (size_t)blockIdx.x * 1024; base < n16; base += (size_t)gridDim.x * 1024) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = base + u * 256 + threadIdx.x;
            if (i < n16) p[i] = v;
        }
    }
}

inline int sentinel_fill(void *xchg, size_t floats, hipStream_t s) {
    const size_t n16 = floats / 4;                      // exchange sizes are multiples of 64 floats
    if (n16 * 4 != floats || (reinterpret_cast<uintptr_t>(xchg) & 15) != 0)
        return (int)hipMemsetAsync(xchg, 0xFF, floats * 4, s);
    hipLaunchKernelGGL(sentinel_fill_kernel, dim3(2048), dim3(256), 0, s, reinterpret_cast<u32x4 *>(xchg), n16);
    return (int)hipGetLastError();
}

// ASRK_REC_REARM: regions [r0, T) of every group (the last two steps' regions, which the kernel cannot re-arm itself)
__global__ __launch_bounds__(256) void sentinel_fill_tail_kernel(u32x4 *__restrict__ x, int groups, int T, int r0,
                                                                 unsigned step16) {
    const u32x4 v = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    const unsigned n = (unsigned)(T - r0) * step16;              // 16-byte units per group
    const int g = blockIdx.y;
    u32x4 *q = x + ((size_t)g * T + r0) * step16;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) q[i] = v;
}

// what a re-arming launch leaves to do: the regions of its last two steps
inline int sentinel_fill_tail(void *xchg, int groups, int T, size_t step_floats, hipStream_t s) {
    const int r0 = std::max(0, T - 2);
    const unsigned step16 = (unsigned)(step_floats / 4);
    const unsigned n = (unsigned)(T - r0) * step16;
    hipLaunchKernelGGL(sentinel_fill_tail_kernel, dim3(std::max(1u, std::min(64u, (n + 1023) / 1024)), groups), dim3(256),
                       0, s, reinterpret_cast<u32x4 *>(xchg), groups, T, r0, step16);
    return (int)hipGetLastError();
}

// canary words per (group, step): 4 per producer workgroup, padded to whole 256-B rows
inline int canary_words(int nwg) { return ((4 * nwg + 63) / 64) * 64; }

unsigned long long *g_dbg_buf = nullptr;
int g_dbg_steps = 0;

// as fit.  gmax = groups that fit beside each other.
inline void chunk_groups(int ndir, int nbg, int gmax, int &ndir_l, int &nbg_l, int &launches) {
    ndir_l = gmax >= ndir ? ndir : 1;
    nbg_l = std::max(1, std::min(nbg, gmax / ndir_l));
    launches = ((ndir + ndir_l - 1) / ndir_l) * ((nbg + nbg_l - 1) / nbg_l);
}

FwdPlan plan_fwd(int T, int B, int H, int ndir, int ncu, int flags) {
    const AsrkKnobs &kn = asrk_knobs_();
    FwdPlan best{};
    best.ok = false;
    long best_cost = -1;
    const int kg = (H + 15) / 16;
    const int kgw_need = (kg + 3) / 4;
    const int KGW = kgw_need <= 4 ? 4 : kgw_need <= 8 ? 8 : kgw_need <= 16 ? 16 : 0;
    if (!KGW) return best;
    // LDS row pitch of the W_hh slice: rows are read with ds_read_b128 by (row = lane & 15, 16-B
    // slot = lane >> 4); the hardware serves that instruction in the lane groups {0-3,12-15,20-27},
    // {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS), and a pitch of 2 slots mod 16 (8 floats past
    // a multiple of 64) is conflict-free for all of them; the former +4 floats cost one extra LDS
    // cycle per group (half of the ~45 % SQ_LDS_BANK_CONFLICT share of both kernels).
    const int HP = KGW * 4 * 16 + 8;
    static const int combos[6][2] = {{1, 1}, {1, 2}, {2, 1}, {2, 2}, {1, 4}, {4, 1}};
    // tuning overrides (experiments): ASRK_FWD_MT / ASRK_FWD_NT force a tile
    const int oc = 1;   // workgroups per CU the persistent grids may take (oversubscription was measured and lost)
    for (auto &c : combos) {
        const int MT = c[0], NT = c[1];
        if (kn.is_set(kn.fwd_mt) && kn.fwd_mt != MT) continue;
        if (kn.is_set(kn.fwd_nt) && kn.fwd_nt != NT) continue;
        const int U = 4 * MT, BG = 16 * NT;
        const int nwg = (H + U - 1) / U, nbg = (B + BG - 1) / BG;
        const long wgs = (long)ndir * nbg * nwg;
        if (wgs > (long)ncu * oc) continue;
        const size_t red1 = (size_t)4 * MT * NT * RED_PITCH * 16;
        size_t lds = (size_t)MT * 16 * HP * 4 + 2 * red1 + 16;
        int db = 1;
        const size_t lds_cap = (size_t)158 * 1024 / (wgs > ncu ? oc : 1);
        if (lds > lds_cap) {  // fall back to single-buffered partial sums (+1 barrier/step)
            lds -= red1;
            db = 0;
        }
        if (lds > lds_cap) continue;
        // per-step MFMA work per wave; tie-break towards more (smaller) exchange groups
        const long cost = (long)MT * NT * 1000 - nbg;
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = FwdPlan{MT, NT, KGW, U, nwg, nbg, BG, HP, kg, db, lds,
                           (size_t)ndir * nbg * T * ((size_t)kg * NT * 256 + canary_words(nwg)), true,
                           ndir, nbg};
        }
    }
    if (best.ok) {
        // wide layers: the step is bounded by the f32 MFMA work -> the bf16x6 kernel (MT = 2, H = 512 / 1024)
        const bool no_bf = (flags & ASRK_REC_F32_MFMA) != 0;
        if (!no_bf && best.MT == 2 && (H == 512 || H == 1024)) {
            const size_t wl = (size_t)2 * 2 * (H / 32) * 1024;
            const size_t red1 = (size_t)4 * 2 * best.NT * RED_PITCH * 16;
            size_t lds = wl + 2 * red1 + 16;
            int db = 1;
            if (lds > (size_t)158 * 1024) { lds -= red1; db = 0; }
            if (lds <= (size_t)158 * 1024) {
                best.bf = 1;
                best.db = db;
                best.lds = lds;
                best.xfloats = (size_t)ndir * best.nbg * T *
                               ((size_t)(H / 32) * best.NT * 3 * 256 + canary_words(best.nwg));
            }
            // H = 1024: the step is bound by the fragment bytes a workgroup pulls through its 64 B/clk
            // vector-memory path (batch rows x H x 6 B): 16 units x 16 batch rows per workgroup halves them
            // for the same MFMA work (slice: plane 0 in LDS, planes 1, 2 in 256 VGPRs)
            const bool no16 = kn.get(kn.rec_bf_mt4, 1) == 0;
            const int nbg16 = (B + 15) / 16;
            if (best.bf && !no16 && H == 1024 && (long)ndir * nbg16 * (H / 16) <= ncu) {
                best.MT = 4; best.NT = 1; best.U = 16; best.BG = 16;
                best.nwg = H / 16; best.nbg = nbg16; best.ndir_l = ndir; best.nbg_l = nbg16;
                best.db = 0;
                best.lds = (size_t)4 * (H / 32) * 1024 + (size_t)4 * 4 * RED_PITCH * 16 + 16;
                best.xfloats = (size_t)ndir * nbg16 * T * ((size_t)(H / 32) * 3 * 256 + canary_words(best.nwg));
            }
        }
        return best;
    }
    // nothing fits in one launch: split the independent groups over several launches
    for (auto &c : combos) {
        const int MT = c[0], NT = c[1];
        const int U = 4 * MT, BG = 16 * NT;
        const int nwg = (H + U - 1) / U, nbg = (B + BG - 1) / BG;
        if (nwg > ncu) continue;
        const size_t red1 = (size_t)4 * MT * NT * RED_PITCH * 16;
        size_t lds = (size_t)MT * 16 * HP * 4 + 2 * red1 + 16;
        int db = 1;
        if (lds > (size_t)158 * 1024) { lds -= red1; db = 0; }
        if (lds > (size_t)158 * 1024) continue;
        int ndir_l, nbg_l, launches;
        chunk_groups(ndir, nbg, ncu / nwg, ndir_l, nbg_l, launches);
        const long cost = (long)launches * MT * NT * 1000;
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = FwdPlan{MT, NT, KGW, U, nwg, nbg, BG, HP, kg, db, lds,
                           (size_t)ndir_l * nbg_l * T * ((size_t)kg * NT * 256 + canary_words(nwg)), true,
                           ndir_l, nbg_l};
        }
    }
    return best;
}


// LDS geometry of the backward kernel for a (UB, NT, RK) choice
inline size_t bwd_lds(int kg, int UB, int NT, int RK, int &HPb, int &KP) {
    const int CH = bwd_ring_kgroups(NT, RK);         // must match the kernel ring
    HPb = ((kg - RK + CH - 1) / CH) * CH * 16;       // LDS-resident gate columns padded to whole chunks
    KP = 4 * HPb + 8;                                // pitch = 2 slots mod 16: see HP in plan_fwd
    return (size_t)UB * KP * 4 + (size_t)HPb * 4 + (size_t)2 * 4 * NT * RED_PITCH * 16 + 16;
}
// register-resident k-groups to try for a tile: none, or 32 (128 VGPRs per lane) for one-batch-tile
// plans whose slice is long enough that every ring refill of the register phase is a full k-group
inline int bwd_rk_options(int kg, int H, int NT, int (&opts)[2]) {
    opts[0] = 0;
    const int kg_plain = (H & 15) ? kg - 1 : kg;
    if (NT == 1 && 32 + bwd_ring_kgroups(1, 32) <= kg_plain) {
        opts[1] = 32;
        return 2;
    }
    return 1;
}

BwdPlan plan_bwd(int T, int B, int H, int ndir, int ncu, int flags) {
    const AsrkKnobs &kn = asrk_knobs_();
    BwdPlan best{};
    best.ok = false;
    const int kg = (H + 15) / 16;
    static const int ubs[3] = {16, 8, 4};
    static const int nts[3] = {1, 2, 4};
    const int oc = 1;   // workgroups per CU the persistent grids may take (oversubscription was measured and lost)
    for (int UB : ubs) {
        if (kn.is_set(kn.bwd_ub) && kn.bwd_ub != UB) continue;
        for (int NT : nts) {
            if (kn.is_set(kn.bwd_nt) && kn.bwd_nt != NT) continue;
            int rks[2];
            const int nrk = bwd_rk_options(kg, H, NT, rks);
            for (int ri = 0; ri < nrk; ++ri) {
                const int RK = rks[ri];
                int HPb, KP;
                const size_t lds = bwd_lds(kg, UB, NT, RK, HPb, KP);
                const int BG = 16 * NT;
                const int nwg = (H + UB - 1) / UB, nbg = (B + BG - 1) / BG;
                const long wgs = (long)ndir * nbg * nwg;
                if (wgs > (long)ncu * oc) continue;
                if (lds > (size_t)158 * 1024 / (wgs > ncu ? oc : 1)) continue;
                best = BwdPlan{NT, UB, nwg, nbg, BG, HPb, KP, kg, lds,
                               (size_t)ndir * nbg * T * ((size_t)4 * kg * NT * 256 + canary_words(nwg)),
                               true, ndir, nbg, RK};
                // wide layers: BPTT on the bf16 matrix cores (16 units x 16 batch rows per workgroup)
                const bool no_bf = (flags & ASRK_REC_F32_MFMA) != 0;
                if (!no_bf && NT == 1 && UB == 16 && BG == 16 && (H == 512 || H == 1024)) {
                    const int KS = H / 32, KL = H == 1024 ? 11 : 0;
                    best.bf = 1;
                    best.lds = (size_t)4 * KL * 3 * 1024 + (size_t)2 * 4 * RED_PITCH * 16 + 2 * 6144 + 16;
                    best.xfloats = (size_t)ndir * nbg * T * ((size_t)4 * KS * 3 * 256 + canary_words(nwg));
                }
                return best;
            }
        }
    }
    // nothing fits in one launch: split the independent groups over several launches
    long best_cost = -1;
    for (int UB : ubs) {
        for (int NT : nts) {
            int rks[2];
            const int nrk = bwd_rk_options(kg, H, NT, rks);
            for (int ri = 0; ri < nrk; ++ri) {
                const int RK = rks[ri];
                int HPb, KP;
                const size_t lds = bwd_lds(kg, UB, NT, RK, HPb, KP);
                const int BG = 16 * NT;
                const int nwg = (H + UB - 1) / UB, nbg = (B + BG - 1) / BG;
                if (nwg > ncu || lds > (size_t)158 * 1024) continue;
                int ndir_l, nbg_l, launches;
                chunk_groups(ndir, nbg, ncu / nwg, ndir_l, nbg_l, launches);
                const long cost = (long)launches * NT * 1000 + (16 / UB);
                if (best_cost < 0 || cost < best_cost) {
                    best_cost = cost;
                    best = BwdPlan{NT, UB, nwg, nbg, BG, HPb, KP, kg, lds,
                                   (size_t)ndir_l * nbg_l * T * ((size_t)4 * kg * NT * 256 + canary_words(nwg)),
                                   true, ndir_l, nbg_l, RK};
                }
            }
        }
    }
    return best;
}

int rec_fwd_impl(bool gru, float *G, const float *whh_f, const float *whh_r, float *Y, float *C, int T, int B,
                 int H, int ndir, void *xchg, int xchg_prefilled, void *ws, float *Y2, int pyr_mode,
                 int pyr_rate, int flags, void *stream, const int64_t *lens = nullptr, void *x2_panel = nullptr);
int rec_bwd_impl(bool gru, float *gates, const float *whh_f, const float *whh_r, const float *C,
                 const float *dY, int T, int B, int H, int ndir, void *xchg, int xchg_prefilled, void *ws,
                 float *db, int pyr_mode, int pyr_rate, int flags, void *stream, void *dg_panel = nullptr,
                 void *dgt_panel = nullptr);

}  // namespace


// debug: device buffer of steps*4*8 uint64 receiving workgroup 0's phase timeline (NULL = off)
extern "C" void asrk_lstm_set_debug_(void *buf, int steps) {
    g_dbg_buf = reinterpret_cast<unsigned long long *>(buf);
    g_dbg_steps = steps;
}

extern "C" size_t asrk_lstm_ws_bytes(void) { return WS_WORDS * sizeof(unsigned); }

extern "C" int asrk_lstm_plan_workgroups(int T, int B, int H, int ndir, int backward, int flags) {
    if (flags < 0 || T <= 0 || B <= 0 || H <= 0 || (ndir != 1 && ndir != 2) || H % 4 != 0) return 0;
    const int ncu = asrk_cu_count_();
    if (ncu <= 0) return 0;
    if (backward) {
        BwdPlan pl = plan_bwd(T, B, H, ndir, ncu, flags);
        return pl.ok ? pl.ndir_l * pl.nbg_l * pl.nwg : 0;
    }
    FwdPlan pl = plan_fwd(T, B, H, ndir, ncu, flags);
    return pl.ok ? pl.ndir_l * pl.nbg_l * pl.nwg : 0;
}

extern "C" size_t asrk_lstm_xchg_bytes(int T, int B, int H, int ndir, int backward, int flags) {
    if (flags < 0 || T <= 0 || B <= 0 || H <= 0 || (ndir != 1 && ndir != 2)) return 0;
    const int ncu = asrk_cu_count_();
    if (ncu <= 0) return 0;
    if (backward) {
        BwdPlan pl = plan_bwd(T, B, H, ndir, ncu, flags);
        return pl.ok ? pl.xfloats * 4 : 0;
    }
    FwdPlan pl = plan_fwd(T, B, H, ndir, ncu, flags);
    return pl.ok ? pl.xfloats * 4 : 0;
}

extern "C" int asrk_lstm_rec_fwd_f32(float *G, const float *whh_f, const float *whh_r, float *Y,
                                     float *C, int T, int B, int H, int ndir, void *xchg,
                                     int xchg_prefilled, void *ws, int flags, void *stream) {
    return asrk_lstm_rec_fwd_pyr_f32(G, whh_f, whh_r, Y, C, T, B, H, ndir, xchg, xchg_prefilled, ws, nullptr,
                                     0, 1, flags, stream);
}

extern "C" int asrk_lstm_rec_fwd_pyr_f32(float *G, const float *whh_f, const float *whh_r, float *Y,
                                         float *C, int T, int B, int H, int ndir, void *xchg,
                                         int xchg_prefilled, void *ws, float *Y2, int pyr_mode,
                                         int pyr_rate, int flags, void *stream) {
    if (!C) return ASRK_EINVAL;
    return rec_fwd_impl(false, G, whh_f, whh_r, Y, C, T, B, H, ndir, xchg, xchg_prefilled, ws, Y2, pyr_mode,
                        pyr_rate, flags, stream);
}

// The pyr form that ALSO emits the layer's output as the row-major split panel of the next layer's input (include/asrk.h).
extern "C" int asrk_lstm_rec_fwd_pyr_panel_f32(float *G, const float *whh_f, const float *whh_r, float *Y,
                                               float *C, int T, int B, int H, int ndir, void *xchg,
                                               int xchg_prefilled, void *ws, float *Y2, int pyr_mode,
                                               int pyr_rate, void *x2_panel, int flags, void *stream) {
    if (!C || !x2_panel) return ASRK_EINVAL;
    return rec_fwd_impl(false, G, whh_f, whh_r, Y, C, T, B, H, ndir, xchg, xchg_prefilled, ws, Y2, pyr_mode,
                        pyr_rate, flags, stream, nullptr, x2_panel);
}

// 1 if the launch of this shape runs on the bf16x6 recurrence kernel (the one that can emit panels)
extern "C" int asrk_lstm_plan_is_bf(int T, int B, int H, int ndir, int backward, int flags) {
    if (flags < 0 || T <= 0 || B <= 0 || H <= 0 || (ndir != 1 && ndir != 2) || H % 4 != 0) return 0;
    const int ncu = asrk_cu_count_();
    if (ncu <= 0) return 0;
    if (backward) {
        BwdPlan pl = plan_bwd(T, B, H, ndir, ncu, flags);
        return pl.ok && pl.bf ? 1 : 0;
    }
    FwdPlan pl = plan_fwd(T, B, H, ndir, ncu, flags);
    return pl.ok && pl.bf ? 1 : 0;
}

// Inference form with per-row sequence lengths (include/asrk.h): the batched beam-search encoder.
extern "C" int asrk_lstm_rec_fwd_len_f32(float *G, const float *whh_f, const float *whh_r, float *Y,
                                         float *C, const int64_t *lens, int T, int B, int H, int ndir, void *xchg,
                                         int xchg_prefilled, void *ws, float *Y2, int pyr_mode,
                                         int pyr_rate, int flags, void *stream) {
    if (!C || !lens) return ASRK_EINVAL;
    return rec_fwd_impl(false, G, whh_f, whh_r, Y, C, T, B, H, ndir, xchg, xchg_prefilled, ws, Y2, pyr_mode,
                        pyr_rate, flags, stream, lens);
}

// torch.nn.GRU recurrence over a whole sequence (src/module.py:125-156 with module='GRU', src/lm.py:20).
// G [T*B, ndir*4H]: per direction the blocks (x W_ir^T + b_ir + b_hr, x W_iz^T + b_iz + b_hz,
// x W_in^T + b_in, b_hn broadcast); whh [3H, H]; on return G holds (r, z, n, W_hn h + b_hn), Y the outputs.
extern "C" int asrk_gru_rec_fwd_f32(float *G, const float *whh_f, const float *whh_r, float *Y, int T, int B,
                                    int H, int ndir, void *xchg, int xchg_prefilled, void *ws, float *Y2,
                                    int pyr_mode, int pyr_rate, int flags, void *stream) {
    return rec_fwd_impl(true, G, whh_f, whh_r, Y, nullptr, T, B, H, ndir, xchg, xchg_prefilled, ws, Y2,
                        pyr_mode, pyr_rate, flags, stream);
}

// BPTT of the above. gates = what the forward left in G, Y = its outputs. On return gates holds
// (dr, dz, dn, dn*r): columns [0,3H) are the input-side gate gradients, columns {0..2H, 3H..4H} the
// hidden-side ones; db [ndir*4H] their column sums (db_ih = db[0:3H], db_hh = db[0:2H] ++ db[3H:4H]).
extern "C" int asrk_gru_rec_bwd_f32(float *gates, const float *whh_f, const float *whh_r, const float *Y,
                                    const float *dY, int T, int B, int H, int ndir, void *xchg,
                                    int xchg_prefilled, void *ws, float *db, int pyr_mode, int pyr_rate,
                                    int flags, void *stream) {
    return rec_bwd_impl(true, gates, whh_f, whh_r, Y, dY, T, B, H, ndir, xchg, xchg_prefilled, ws, db,
                        pyr_mode, pyr_rate, flags, stream);
}

namespace {
int rec_fwd_impl(bool gru, float *G, const float *whh_f, const float *whh_r, float *Y, float *C, int T, int B,
                 int H, int ndir, void *xchg, int xchg_prefilled, void *ws, float *Y2, int pyr_mode,
                 int pyr_rate, int flags, void *stream, const int64_t *lens, void *x2_panel) {
    const AsrkKnobs &kn = asrk_knobs_();
    if (flags < 0 || pyr_mode < 0 || pyr_mode > 2 || pyr_rate < 1) return ASRK_EINVAL;
    // a time reduction whose output is empty ('concat' with T < rate) needs no Y2
    const bool y2_empty = pyr_mode == 1 && T / pyr_rate == 0;
    if (pyr_mode != 0 && !Y2 && !y2_empty) return ASRK_EINVAL;
    if (T < 0 || B <= 0 || H <= 0 || (ndir != 1 && ndir != 2)) return ASRK_EINVAL;
    if (T == 0) return ASRK_OK;
    if (!G || !whh_f || (ndir == 2 && !whh_r) || !Y || (!gru && !C) || !ws || !xchg) return ASRK_EINVAL;
    if (H % 4 != 0) return ASRK_ESHAPE;
    if ((reinterpret_cast<uintptr_t>(xchg) & 15) != 0) return ASRK_EINVAL;
    const int ncu = asrk_cu_count_();
    if (ncu <= 0) return ASRK_EDEVICE;
    FwdPlan pl = plan_fwd(T, B, H, ndir, ncu, flags);
    if (!pl.ok) return ASRK_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    RecFwdArgs a;
    a.G = G; a.whh[0] = whh_f; a.whh[1] = ndir == 2 ? whh_r : whh_f;
    a.Y = Y; a.C = C; a.X = reinterpret_cast<float *>(xchg);
    a.err = reinterpret_cast<unsigned *>(ws);
    a.T = T; a.B = B; a.H = H; a.ldg = ndir * 4 * H; a.ldy = ndir * H;
    a.U = pl.U; a.nwg = pl.nwg; a.BG = pl.BG; a.HP = pl.HP; a.kgp = pl.kgp;
    a.canw = canary_words(pl.nwg);
    a.poll_mode = kn.get(kn.fwd_poll, 0);
    a.poll_mode |= (kn.get(kn.fwd_presleep, 8) & 0xff) << 8;  // x64 cycles
    a.dbg = g_dbg_buf; a.dbg_steps = kn.is_set(kn.dbg_noload) ? -1 : g_dbg_steps;
    a.Y2 = pyr_mode ? Y2 : nullptr; a.pyr_mode = pyr_mode; a.pyr_rate = pyr_rate;
    a.lens = lens;
    a.P2 = nullptr; a.p2_stride = 0;
    if (x2_panel) {
        // the next layer's A panel: only the bf16x6 kernel emits it, for the plain / 'concat' layouts, uniform lengths
        if (!pl.bf || pyr_mode == 2 || lens || (ndir * H) % 8 != 0 || (reinterpret_cast<uintptr_t>(x2_panel) & 15))
            return ASRK_ESHAPE;
        const int r = pyr_mode == 1 ? pyr_rate : 1;
        a.P2 = reinterpret_cast<unsigned char *>(x2_panel);
        a.p2_stride = asrk_split_panel_stride_((T / r) * B, r * ndir * H);
        if (pyr_mode == 0) { a.pyr_rate = 1; }
    }
    const bool one_launch = pl.ndir_l >= ndir && pl.nbg_l >= pl.nbg;
    a.rearm = (flags & ASRK_REC_REARM) && one_launch ? 1 : 0;
    asrk_prof_begin_(PROF_LSTM_FWD, s);
    int rc = ASRK_OK;
    bool first = true;
    // one launch covers pl.ndir_l directions x pl.nbg_l batch groups (normally everything)
    for (int d0 = 0; d0 < ndir && rc == ASRK_OK; d0 += pl.ndir_l)
        for (int g0 = 0; g0 < pl.nbg && rc == ASRK_OK; g0 += pl.nbg_l) {
            a.ndir = std::min(pl.ndir_l, ndir - d0);
            a.nbg = std::min(pl.nbg_l, pl.nbg - g0);
            a.dir0 = d0; a.bg0 = g0;
            // sentinel = "not written yet" (the caller may have filled the buffer with 0xFF bytes
            // earlier, off the critical path; later launches reuse it and must refill)
            if (!(first && xchg_prefilled)) { const int frc = sentinel_fill(xchg, pl.xfloats, s); if (frc) return frc; }
            first = false;
            const int grid = a.ndir * a.nbg * pl.nwg;
            rc = pl.bf ? launch_fwd_bf(gru, a, pl, H, grid, s) : launch_fwd_f32(gru, a, pl, grid, s);
        }
    if (rc == ASRK_OK && (flags & ASRK_REC_REARM)) {
        // hand the buffer back armed: the kernel re-armed regions 0 .. T - 3 on the fly; several launches shared the
        // buffer (more groups than CUs): one full fill instead
        const int frc = a.rearm ? sentinel_fill_tail(xchg, ndir * pl.nbg, T, pl.xfloats / ((size_t)ndir * pl.nbg * T), s)
                                : sentinel_fill(xchg, pl.xfloats, s);
        if (frc) rc = frc;
    }
    asrk_prof_end_(PROF_LSTM_FWD, s);
    return rc;
}
}  // namespace

extern "C" int asrk_lstm_rec_bwd_f32(float *gates, const float *whh_f, const float *whh_r,
                                     const float *C, const float *dY, int T, int B, int H, int ndir,
                                     void *xchg, int xchg_prefilled, void *ws, float *db, int flags, void *stream) {
    return asrk_lstm_rec_bwd_pyr_f32(gates, whh_f, whh_r, C, dY, T, B, H, ndir, xchg, xchg_prefilled, ws, db,
                                     0, 1, flags, stream);
}

extern "C" int asrk_lstm_rec_bwd_pyr_f32(float *gates, const float *whh_f, const float *whh_r,
                                         const float *C, const float *dY, int T, int B, int H, int ndir,
                                         void *xchg, int xchg_prefilled, void *ws, float *db,
                                         int pyr_mode, int pyr_rate, int flags, void *stream) {
    return rec_bwd_impl(false, gates, whh_f, whh_r, C, dY, T, B, H, ndir, xchg, xchg_prefilled, ws, db,
                        pyr_mode, pyr_rate, flags, stream);
}

extern "C" int asrk_lstm_rec_bwd_pyr_panel_f32(float *gates, const float *whh_f, const float *whh_r,
                                               const float *C, const float *dY, int T, int B, int H, int ndir,
                                               void *xchg, int xchg_prefilled, void *ws, float *db,
                                               int pyr_mode, int pyr_rate, void *dg_panel, void *dgt_panel, int flags,
                                               void *stream) {
    if (!dg_panel && !dgt_panel) return ASRK_EINVAL;
    return rec_bwd_impl(false, gates, whh_f, whh_r, C, dY, T, B, H, ndir, xchg, xchg_prefilled, ws, db,
                        pyr_mode, pyr_rate, flags, stream, dg_panel, dgt_panel);
}

namespace {
int rec_bwd_impl(bool gru, float *gates, const float *whh_f, const float *whh_r, const float *C,
                 const float *dY, int T, int B, int H, int ndir, void *xchg, int xchg_prefilled, void *ws,
                 float *db, int pyr_mode, int pyr_rate, int flags, void *stream, void *dg_panel, void *dgt_panel) {
    const AsrkKnobs &kn = asrk_knobs_();
    if (flags < 0 || pyr_mode < 0 || pyr_mode > 2 || pyr_rate < 1) return ASRK_EINVAL;
    // gradient of an EMPTY reduced tensor ('concat' with T < rate): every step sees dY = 0
    const bool dy_empty = pyr_mode == 1 && T / pyr_rate == 0;
    if (T < 0 || B <= 0 || H <= 0 || (ndir != 1 && ndir != 2)) return ASRK_EINVAL;
    if (T == 0) return ASRK_OK;
    if (!gates || !whh_f || (ndir == 2 && !whh_r) || !C || (!dY && !dy_empty) || !ws || !xchg) return ASRK_EINVAL;
    if (H % 4 != 0) return ASRK_ESHAPE;
    if ((reinterpret_cast<uintptr_t>(xchg) & 15) != 0) return ASRK_EINVAL;
    const int ncu = asrk_cu_count_();
    if (ncu <= 0) return ASRK_EDEVICE;
    BwdPlan pl = plan_bwd(T, B, H, ndir, ncu, flags);
    if (!pl.ok) return ASRK_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    RecBwdArgs a;
    a.G = gates; a.whh[0] = whh_f; a.whh[1] = ndir == 2 ? whh_r : whh_f;
    a.C = C; a.dY = dY; a.X = reinterpret_cast<float *>(xchg);
    a.err = reinterpret_cast<unsigned *>(ws);
    a.T = T; a.B = B; a.H = H; a.ldg = ndir * 4 * H; a.ldy = ndir * H;
    a.UB = pl.UB; a.nwg = pl.nwg; a.BG = pl.BG; a.HPb = pl.HPb; a.KP = pl.KP;
    a.kgp = pl.kgp; a.canw = canary_words(pl.nwg);
    a.poll_mode = kn.get(kn.bwd_poll, 1);
    if (kn.is_set(kn.bwd_presleep)) a.poll_mode |= (kn.bwd_presleep & 0xff) << 8;
    a.dbg = g_dbg_buf; a.dbg_steps = kn.is_set(kn.dbg_noload) ? -1 : g_dbg_steps;
    a.db = db;
    a.pyr_mode = pyr_mode; a.pyr_rate = pyr_rate;
    a.PG = nullptr; a.pg_stride = 0;
    if (dg_panel) {
        if (!pl.bf || gru || (reinterpret_cast<uintptr_t>(dg_panel) & 15)) return ASRK_ESHAPE;
        a.PG = reinterpret_cast<unsigned char *>(dg_panel);
        a.pg_stride = asrk_split_panel_stride_(T * B, ndir * 4 * H);
    }
    a.PT = nullptr; a.pt_stride = 0;
    if (dgt_panel) {
        if (!pl.bf || gru || B % 16 != 0 || (reinterpret_cast<uintptr_t>(dgt_panel) & 15)) return ASRK_ESHAPE;
        a.PT = reinterpret_cast<unsigned char *>(dgt_panel);
        a.pt_stride = asrk_split_panel_stride_(ndir * 4 * H, T * B);
    }
    const bool one_launch = pl.ndir_l >= ndir && pl.nbg_l >= pl.nbg;
    a.rearm = (flags & ASRK_REC_REARM) && one_launch ? 1 : 0;
    if (db) ASRK_HIP(hipMemsetAsync(db, 0, (size_t)ndir * 4 * H * sizeof(float), s));
    asrk_prof_begin_(PROF_LSTM_BWD, s);
    int rc = ASRK_OK;
    bool first = true;
    for (int d0 = 0; d0 < ndir && rc == ASRK_OK; d0 += pl.ndir_l)
        for (int g0 = 0; g0 < pl.nbg && rc == ASRK_OK; g0 += pl.nbg_l) {
            a.ndir = std::min(pl.ndir_l, ndir - d0);
            a.nbg = std::min(pl.nbg_l, pl.nbg - g0);
            a.dir0 = d0; a.bg0 = g0;
            if (!(first && xchg_prefilled)) { const int frc = sentinel_fill(xchg, pl.xfloats, s); if (frc) return frc; }
            first = false;
            const int grid = a.ndir * a.nbg * pl.nwg;
            rc = pl.bf ? launch_bwd_bf(gru, a, pl, grid, s) : launch_bwd_f32(gru, a, pl, grid, s);
        }
    if (rc == ASRK_OK && (flags & ASRK_REC_REARM)) {
        const int frc = a.rearm ? sentinel_fill_tail(xchg, ndir * pl.nbg, T, pl.xfloats / ((size_t)ndir * pl.nbg * T), s)
                                : sentinel_fill(xchg, pl.xfloats, s);
        if (frc) rc = frc;
    }
    asrk_prof_end_(PROF_LSTM_BWD, s);
    return rc;
}
}  // namespace

extern "C" int asrk_lstm_check_error(void *ws, void *stream) {
    if (!ws) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    unsigned e = 0;
    unsigned *errp = reinterpret_cast<unsigned *>(ws);
    ASRK_HIP(hipMemcpyAsync(&e, errp, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    ASRK_HIP(hipStreamSynchronize(s));
    if (e != 0) {
        ASRK_HIP(hipMemsetAsync(errp, 0, sizeof(unsigned), s));
        return ASRK_ETIMEOUT;
    }
    return ASRK_OK;
}

