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
//  * per step the only inter-workgroup traffic is h_{t-1} [B,H]: it is read straight from the
//    output tensor Y (time-major) that the producers wrote one step earlier.
//  * exact-f32 MFMA 16x16x4: M = gate rows (unit-major, gate-minor, so one lane ends up with
//    i,f,g,o of ONE (unit,batch) cell -> the cell update is lane-local), N = batch, K = H split
//    over the 4 waves; partial sums meet in LDS.
//  * in-launch hand-off follows the agent-scope recipe (MI355X guide §G16, R1): h is stored
//    write-through (sc1), every storing wave drains vmcnt, one lane publishes a per-workgroup
//    epoch flag (sc1 store); consumers poll only the flags of the producers of THEIR K-slice with
//    relaxed sc1 loads and then read h with sc1 buffer loads (L1-bypassing), so no acquire
//    fence / L1 invalidate is on the critical path.  No dispatch-order or XCD-placement
//    assumption; every spin is wall-clock bounded and reports ASRK_ETIMEOUT.
//
// Backward design (BPTT): same stationarity with W_hh^T slices ([UB units][4H]) in LDS; the
// per-step exchange is dG_{t+1} [B,4H] (pre-activation gradients, written in place over the
// saved gates); wave w contracts gate w's H rows.  dW_hh/dW_ih/dX/db are plain GEMMs/column sums
// on the finished dG (ops layer).
#include "common.h"

extern "C" int asrk_cu_count_(void);

namespace {

constexpr int WS_MAX_FLAGS = 1024;
constexpr unsigned long long TIMEOUT_TICKS = 300000000ull;  // 3 s of the 100 MHz wall clock

struct RecFwdArgs {
    float *G;
    const float *whh[2];
    float *Y, *C;
    unsigned *flags, *err;
    int T, B, H, ndir, ldg, ldy;
    int U, nwg, nbg, BG, HP;
};

struct RecBwdArgs {
    float *G;
    const float *whh[2];
    const float *C, *dY;
    unsigned *flags, *err;
    int T, B, H, ndir, ldg, ldy;
    int UB, nwg, nbg, BG, HPb, KP;
};

// Wait until flags[lo .. lo+count) >= epoch. One wave; relaxed agent-scope (sc1) polls.
__device__ __forceinline__ bool wait_flags(unsigned *flags, int lo, int count, unsigned epoch,
                                           unsigned *err, int lane) {
    unsigned long long t0 = 0;
    unsigned spins = 0;
    for (;;) {
        bool ok = true;
        for (int j = lane; j < count; j += 64) {
            const unsigned f = __hip_atomic_load(flags + lo + j, RLX_AGENT);
            ok &= (f >= epoch);
        }
        if (__all(ok)) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255u) == 0) {
            const unsigned long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            const unsigned e = __hip_atomic_load(err, RLX_AGENT);
            if (e != 0 || now - t0 > TIMEOUT_TICKS) {
                if (lane == 0) __hip_atomic_store(err, 1u, RLX_AGENT);
                return false;
            }
        }
    }
}

template <int MT, int NT, int KGW>
__global__ __launch_bounds__(256) void lstm_rec_fwd_kernel(RecFwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CL = MT * NT * 64;           // cell-lanes (one (unit,batch) cell each)
    constexpr int CPT = (CL + 255) / 256;      // cell-lanes per thread
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ngroups = p.ndir * p.nbg;
    const int group = blockIdx.x % ngroups, wg = blockIdx.x / ngroups;
    const int dir = group % p.ndir, bg = group / p.ndir;
    const int u0 = wg * p.U, b0 = bg * p.BG;
    const int nb = min(p.BG, p.B - b0);
    const int H = p.H, HP = p.HP;

    float *Ws = smem;
    f32x4 *red = reinterpret_cast<f32x4 *>(smem + MT * 16 * HP);
    int *abort_flag = reinterpret_cast<int *>(red + 4 * CL);
    unsigned *gflags = p.flags + group * p.nwg;

    // ---- stage this workgroup's W_hh rows: LDS row m <-> (unit u0 + m/4, gate m%4)
    {
        const float *W = p.whh[dir];
        for (int idx = tid; idx < MT * 16 * HP; idx += 256) {
            const int m = idx / HP, k = idx - m * HP;
            const int unit = u0 + (m >> 2), gate = m & 3;
            float v = 0.f;
            if (k < H && unit < H) v = W[(size_t)(gate * H + unit) * H + k];
            Ws[idx] = v;
        }
        if (tid == 0) *abort_flag = 0;
    }
    __syncthreads();

    // ---- static cell-lane ownership
    int c_unit[CPT], c_b[CPT];
    bool c_valid[CPT];
    float c_state[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int cl = tid + 256 * i;
        const int ln = cl & 63, nt = (cl >> 6) % NT, mt = (cl >> 6) / NT;
        c_unit[i] = u0 + mt * 4 + (ln >> 4);
        const int bl = nt * 16 + (ln & 15);
        c_b[i] = b0 + bl;
        c_valid[i] = (cl < CL) && (bl < nb) && (c_unit[i] < H) && (mt * 4 + (ln >> 4) < p.U);
        c_state[i] = 0.f;
    }

    // producers of this wave's K slice
    const int k_lo = wave * KGW * 16;
    const int k_hi = min(H, k_lo + KGW * 16);
    const int wg_lo = k_lo < H ? k_lo / p.U : 0;
    const int wg_cnt = k_lo < H ? (k_hi - 1) / p.U - wg_lo + 1 : 0;
    const int m16 = lane & 15, q4 = lane >> 4;

    for (int s = 0; s < p.T; ++s) {
        const int t = dir == 0 ? s : p.T - 1 - s;
        const int tprev = dir == 0 ? t - 1 : t + 1;

        // prefetch the input-projection pre-activations of my cells (independent of h)
        float gpre[CPT][4];
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gpre[i][r] = 0.f;
            if (c_valid[i]) {
                const float *g = p.G + ((size_t)t * p.B + c_b[i]) * p.ldg + dir * 4 * H + c_unit[i];
#pragma unroll
                for (int r = 0; r < 4; ++r) gpre[i][r] = g[(size_t)r * H];
            }
        }

        f32x4 acc[MT][NT][2];
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) acc[a][b][h2] = f32x4{0.f, 0.f, 0.f, 0.f};

        bool ok = true;
        if (s > 0 && wg_cnt > 0) {
            ok = wait_flags(gflags, wg_lo, wg_cnt, (unsigned)s, p.err, lane);
            if (ok) {
                // h_{t-1}: B operand fragments straight from Y (sc1 loads, L1 bypass)
                const float *ybase = p.Y + ((size_t)tprev * p.B + b0) * p.ldy;
                __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)ybase, 0, nb * p.ldy * 4, 0x00020000);
                f32x4 bf[NT][KGW];
#pragma unroll
                for (int kg = 0; kg < KGW; ++kg) {
                    const int k = k_lo + kg * 16 + 4 * q4;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const unsigned off =
                            k < H ? (unsigned)(((nt * 16 + m16) * p.ldy + dir * H + k) * 4)
                                  : 0x7ffffff0u;
                        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
                        bf[nt][kg] = __builtin_bit_cast(f32x4, v);
                    }
                }
                // keep ALL h loads in flight before the first MFMA: one memory round trip per
                // step instead of one per k-group (the scheduler otherwise sinks the loads)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kg = 0; kg < KGW; ++kg) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const f32x4 a = *reinterpret_cast<const f32x4 *>(
                            Ws + (mt * 16 + m16) * HP + k_lo + kg * 16 + 4 * q4);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                acc[mt][nt][j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                    a[j], bf[nt][kg][j], acc[mt][nt][j & 1], 0, 0, 0);
                        }
                    }
                }
            } else if (lane == 0) {
                *abort_flag = 1;
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                red[((wave * MT + mt) * NT + nt) * 64 + lane] = acc[mt][nt][0] + acc[mt][nt][1];
        __syncthreads();  // B1: partial sums visible
        if (*abort_flag) break;

        float gi[CPT], gf[CPT], gg[CPT], go[CPT];
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            if (c_valid[i]) {
                const int cl = tid + 256 * i;
                f32x4 sum = red[cl];
#pragma unroll
                for (int w = 1; w < 4; ++w) sum += red[w * CL + cl];
                gi[i] = sigmoidf_acc(gpre[i][0] + sum[0]);
                gf[i] = sigmoidf_acc(gpre[i][1] + sum[1]);
                gg[i] = tanhf(gpre[i][2] + sum[2]);
                go[i] = sigmoidf_acc(gpre[i][3] + sum[3]);
                c_state[i] = gf[i] * c_state[i] + gi[i] * gg[i];
                const float h = go[i] * tanhf(c_state[i]);
                // write-through (sc1) store: the exchange payload for step s+1
                __hip_atomic_store(p.Y + ((size_t)t * p.B + c_b[i]) * p.ldy + dir * H + c_unit[i], h,
                                   RLX_AGENT);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains
        __syncthreads();                                   // B2
        if (tid == 0) __hip_atomic_store(gflags + wg, (unsigned)(s + 1), RLX_AGENT);

        // saved-for-backward tensors (off the critical path)
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            if (c_valid[i]) {
                float *g = p.G + ((size_t)t * p.B + c_b[i]) * p.ldg + dir * 4 * H + c_unit[i];
                g[0] = gi[i];
                g[(size_t)H] = gf[i];
                g[(size_t)2 * H] = gg[i];
                g[(size_t)3 * H] = go[i];
                p.C[((size_t)t * p.B + c_b[i]) * p.ldy + dir * H + c_unit[i]] = c_state[i];
            }
        }
    }
}

template <int NT, int CH>
__device__ __forceinline__ void bwd_load_chunk(f32x4 (&bf)[NT][CH], __amdgpu_buffer_rsrc_t rs,
                                               int kg0, int kgs, int H, int ldg, int colbase,
                                               int m16, int q4) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int k = (kg0 + c) * 16 + 4 * q4;
        const bool v = (kg0 + c) < kgs && k < H;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const unsigned off = v ? (unsigned)(((nt * 16 + m16) * ldg + colbase + k) * 4)
                                   : 0x7ffffff0u;
            u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
            bf[nt][c] = __builtin_bit_cast(f32x4, x);
        }
    }
    __builtin_amdgcn_sched_barrier(0);  // the whole chunk is issued before anything else moves
}

template <int NT, int CH>
__device__ __forceinline__ void bwd_mfma_chunk(f32x4 (&acc)[NT][2], const f32x4 (&bf)[NT][CH],
                                               const float *wrow, bool row_ok, int kg0, int kgs,
                                               int q4) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        if (kg0 + c < kgs) {
            f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row_ok) a = *reinterpret_cast<const f32x4 *>(wrow + (kg0 + c) * 16 + 4 * q4);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[nt][j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        a[j], bf[nt][c][j], acc[nt][j & 1], 0, 0, 0);
        }
    }
}

template <int NT>
__global__ __launch_bounds__(256) void lstm_rec_bwd_kernel(RecBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int CH = 16 / NT;  // k-groups per prefetch chunk (16 float4 loads in flight / buffer)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ngroups = p.ndir * p.nbg;
    const int group = blockIdx.x % ngroups, wg = blockIdx.x / ngroups;
    const int dir = group % p.ndir, bg = group / p.ndir;
    const int u0 = wg * p.UB, b0 = bg * p.BG;
    const int nb = min(p.BG, p.B - b0);
    const int H = p.H, HPb = p.HPb, KP = p.KP, UB = p.UB;

    float *Wt = smem;  // [UB][KP]: Wt[m][gate*HPb + j] = W_hh[gate*H + j][u0 + m]
    f32x4 *red = reinterpret_cast<f32x4 *>(smem + UB * KP);
    int *abort_flag = reinterpret_cast<int *>(red + 4 * NT * 64);
    unsigned *gflags = p.flags + group * p.nwg;

    {
        for (int idx = tid; idx < UB * KP; idx += 256) Wt[idx] = 0.f;
        __syncthreads();
        const float *W = p.whh[dir];
        const int total = 4 * H * UB;
        for (int idx = tid; idx < total; idx += 256) {
            const int m = idx % UB, rj = idx / UB;  // rj = gate*H + j
            const int gate = rj / H, j = rj - gate * H;
            if (u0 + m < H) Wt[m * KP + gate * HPb + j] = W[(size_t)rj * H + u0 + m];
        }
        if (tid == 0) *abort_flag = 0;
    }
    __syncthreads();

    // cells owned by this thread: ci = tid + 256*i -> (unit = ci%16, batch = ci/16)
    int c_unit[NT], c_b[NT], c_red[NT];
    bool c_valid[NT];
    float dc_carry[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int ci = tid + 256 * i;
        const int ul = ci & 15, bl = ci >> 4;
        c_unit[i] = u0 + ul;
        c_b[i] = b0 + bl;
        c_valid[i] = ul < UB && bl < nb && c_unit[i] < H;
        // reduction buffer address of (unit ul, batch bl): f32x4 index, component ul&3
        c_red[i] = ((bl >> 4) * 64 + (ul >> 2) * 16 + (bl & 15)) * 4 + (ul & 3);
        dc_carry[i] = 0.f;
    }

    const int kgs = (H + 15) / 16;          // k-groups per gate (wave w <-> gate w)
    const int nch = (kgs + CH - 1) / CH;
    const int m16 = lane & 15, q4 = lane >> 4;
    const float *wrow = Wt + m16 * KP + wave * HPb;
    const bool row_ok = m16 < UB;
    const int colbase = dir * 4 * H + wave * H;

    for (int s = 0; s < p.T; ++s) {
        // dir 0 ran t = 0..T-1 forward -> backward walks T-1..0 and needs dG of t+1;
        // dir 1 ran T-1..0 -> backward walks 0..T-1 and needs dG of t-1.
        const int t = dir == 0 ? p.T - 1 - s : s;
        const int tn = dir == 0 ? t + 1 : t - 1;   // step whose dG feeds dh_t
        const int tp = dir == 0 ? t - 1 : t + 1;   // step that produced c_{prev} of t

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
                vdy[i] = p.dY[row * p.ldy + dir * H + c_unit[i]];
                if (tp >= 0 && tp < p.T)
                    vcp[i] = p.C[((size_t)tp * p.B + c_b[i]) * p.ldy + dir * H + c_unit[i]];
            }
        }

        f32x4 acc[NT][2];
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[b][0] = acc[b][1] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (s > 0) {
            const bool ok = wait_flags(gflags, 0, p.nwg, (unsigned)s, p.err, lane);
            if (ok) {
                const float *gb = p.G + ((size_t)tn * p.B + b0) * p.ldg;
                __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)gb, 0, nb * p.ldg * 4, 0x00020000);
                f32x4 bf0[NT][CH], bf1[NT][CH];
                bwd_load_chunk<NT, CH>(bf0, rs, 0, kgs, H, p.ldg, colbase, m16, q4);
                for (int c = 0; c < nch; c += 2) {
                    if (c + 1 < nch)
                        bwd_load_chunk<NT, CH>(bf1, rs, (c + 1) * CH, kgs, H, p.ldg, colbase, m16, q4);
                    bwd_mfma_chunk<NT, CH>(acc, bf0, wrow, row_ok, c * CH, kgs, q4);
                    if (c + 1 < nch) {
                        if (c + 2 < nch)
                            bwd_load_chunk<NT, CH>(bf0, rs, (c + 2) * CH, kgs, H, p.ldg, colbase, m16,
                                                   q4);
                        bwd_mfma_chunk<NT, CH>(acc, bf1, wrow, row_ok, (c + 1) * CH, kgs, q4);
                    }
                }
            } else if (lane == 0) {
                *abort_flag = 1;
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) red[(wave * NT + nt) * 64 + lane] = acc[nt][0] + acc[nt][1];
        __syncthreads();  // B1
        if (*abort_flag) break;

        const float *redf = reinterpret_cast<const float *>(red);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            if (c_valid[i]) {
                float rec = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) rec += redf[w * NT * 256 + c_red[i]];
                const float dh = vdy[i] + rec;
                const float tc = tanhf(vc[i]);
                const float dcell = dh * vo[i] * (1.f - tc * tc) + dc_carry[i];
                dc_carry[i] = dcell * vf[i];
                const float dgi = dcell * vg[i] * vi[i] * (1.f - vi[i]);
                const float dgf = dcell * vcp[i] * vf[i] * (1.f - vf[i]);
                const float dgg = dcell * vi[i] * (1.f - vg[i] * vg[i]);
                const float dgo = dh * tc * vo[i] * (1.f - vo[i]);
                float *g = p.G + ((size_t)t * p.B + c_b[i]) * p.ldg + dir * 4 * H + c_unit[i];
                __hip_atomic_store(g, dgi, RLX_AGENT);
                __hip_atomic_store(g + (size_t)H, dgf, RLX_AGENT);
                __hip_atomic_store(g + (size_t)2 * H, dgg, RLX_AGENT);
                __hip_atomic_store(g + (size_t)3 * H, dgo, RLX_AGENT);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // B2
        if (tid == 0) __hip_atomic_store(gflags + wg, (unsigned)(s + 1), RLX_AGENT);
    }
}

struct FwdPlan {
    int MT, NT, KGW, U, nwg, nbg, BG, HP;
    size_t lds;
    bool ok;
};

FwdPlan plan_fwd(int B, int H, int ndir, int ncu) {
    FwdPlan best{};
    best.ok = false;
    long best_cost = -1;
    const int kg = (H + 15) / 16;
    int kgw_need = (kg + 3) / 4;
    int KGW = kgw_need <= 4 ? 4 : kgw_need <= 8 ? 8 : kgw_need <= 16 ? 16 : 0;
    if (!KGW) return best;
    const int HP = KGW * 4 * 16 + 4;
    static const int combos[6][2] = {{1, 1}, {1, 2}, {2, 1}, {2, 2}, {1, 4}, {4, 1}};
    for (auto &c : combos) {
        const int MT = c[0], NT = c[1];
        const int U = 4 * MT, BG = 16 * NT;
        const int nwg = (H + U - 1) / U, nbg = (B + BG - 1) / BG;
        const long wgs = (long)ndir * nbg * nwg;
        if (wgs > ncu || wgs > WS_MAX_FLAGS) continue;
        const size_t lds = (size_t)MT * 16 * HP * 4 + (size_t)4 * MT * NT * 64 * 16 + 16;
        if (lds > 150 * 1024) continue;
        // per-step MFMA work per wave; tie-break towards more (smaller) sync groups
        const long cost = (long)MT * NT * 1000 - nbg;
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = FwdPlan{MT, NT, KGW, U, nwg, nbg, BG, HP, lds, true};
        }
    }
    return best;
}

struct BwdPlan {
    int NT, UB, nwg, nbg, BG, HPb, KP;
    size_t lds;
    bool ok;
};

BwdPlan plan_bwd(int B, int H, int ndir, int ncu) {
    BwdPlan best{};
    best.ok = false;
    const int HPb = ((H + 15) / 16) * 16;
    const int KP = 4 * HPb + 4;
    static const int ubs[3] = {16, 8, 4};
    static const int nts[3] = {1, 2, 4};
    for (int UB : ubs) {
        for (int NT : nts) {
            const size_t lds = (size_t)UB * KP * 4 + (size_t)4 * NT * 64 * 16 + 16;
            if (lds > 150 * 1024) continue;
            const int BG = 16 * NT;
            const int nwg = (H + UB - 1) / UB, nbg = (B + BG - 1) / BG;
            const long wgs = (long)ndir * nbg * nwg;
            if (wgs > ncu || wgs > WS_MAX_FLAGS) continue;
            best = BwdPlan{NT, UB, nwg, nbg, BG, HPb, KP, lds, true};
            return best;
        }
    }
    return best;
}

template <int MT, int NT, int KGW>
int launch_fwd(const RecFwdArgs &a, int grid, size_t lds, hipStream_t s) {
    auto kern = lstm_rec_fwd_kernel<MT, NT, KGW>;
    ASRK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

template <int MT, int NT>
int launch_fwd_k(const RecFwdArgs &a, int KGW, int grid, size_t lds, hipStream_t s) {
    switch (KGW) {
        case 4: return launch_fwd<MT, NT, 4>(a, grid, lds, s);
        case 8: return launch_fwd<MT, NT, 8>(a, grid, lds, s);
        case 16: return launch_fwd<MT, NT, 16>(a, grid, lds, s);
    }
    return ASRK_ESHAPE;
}

template <int NT>
int launch_bwd(const RecBwdArgs &a, int grid, size_t lds, hipStream_t s) {
    auto kern = lstm_rec_bwd_kernel<NT>;
    ASRK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

}  // namespace

extern "C" size_t asrk_lstm_ws_bytes(void) { return (WS_MAX_FLAGS + 16) * sizeof(unsigned); }

extern "C" int asrk_lstm_rec_fwd_f32(float *G, const float *whh_f, const float *whh_r, float *Y,
                                     float *C, int T, int B, int H, int ndir, void *ws,
                                     void *stream) {
    if (T < 0 || B <= 0 || H <= 0 || (ndir != 1 && ndir != 2)) return ASRK_EINVAL;
    if (T == 0) return ASRK_OK;
    if (!G || !whh_f || (ndir == 2 && !whh_r) || !Y || !C || !ws) return ASRK_EINVAL;
    if (H % 4 != 0) return ASRK_ESHAPE;
    if ((reinterpret_cast<uintptr_t>(Y) & 15) != 0) return ASRK_EINVAL;
    const int ncu = asrk_cu_count_();
    if (ncu <= 0) return ASRK_EDEVICE;
    FwdPlan pl = plan_fwd(B, H, ndir, ncu);
    if (!pl.ok) return ASRK_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    unsigned *flags = reinterpret_cast<unsigned *>(ws);
    ASRK_HIP(hipMemsetAsync(flags, 0, WS_MAX_FLAGS * sizeof(unsigned), s));

    RecFwdArgs a;
    a.G = G; a.whh[0] = whh_f; a.whh[1] = ndir == 2 ? whh_r : whh_f;
    a.Y = Y; a.C = C; a.flags = flags; a.err = flags + WS_MAX_FLAGS;
    a.T = T; a.B = B; a.H = H; a.ndir = ndir; a.ldg = ndir * 4 * H; a.ldy = ndir * H;
    a.U = pl.U; a.nwg = pl.nwg; a.nbg = pl.nbg; a.BG = pl.BG; a.HP = pl.HP;
    const int grid = ndir * pl.nbg * pl.nwg;
    asrk_prof_begin_(PROF_LSTM_FWD, s);
    int rc = ASRK_ESHAPE;
    if (pl.MT == 1 && pl.NT == 1) rc = launch_fwd_k<1, 1>(a, pl.KGW, grid, pl.lds, s);
    else if (pl.MT == 1 && pl.NT == 2) rc = launch_fwd_k<1, 2>(a, pl.KGW, grid, pl.lds, s);
    else if (pl.MT == 2 && pl.NT == 1) rc = launch_fwd_k<2, 1>(a, pl.KGW, grid, pl.lds, s);
    else if (pl.MT == 2 && pl.NT == 2) rc = launch_fwd_k<2, 2>(a, pl.KGW, grid, pl.lds, s);
    else if (pl.MT == 1 && pl.NT == 4) rc = launch_fwd_k<1, 4>(a, pl.KGW, grid, pl.lds, s);
    else if (pl.MT == 4 && pl.NT == 1) rc = launch_fwd_k<4, 1>(a, pl.KGW, grid, pl.lds, s);
    asrk_prof_end_(PROF_LSTM_FWD, s);
    return rc;
}

extern "C" int asrk_lstm_rec_bwd_f32(float *gates, const float *whh_f, const float *whh_r,
                                     const float *C, const float *dY, int T, int B, int H, int ndir,
                                     void *ws, void *stream) {
    if (T < 0 || B <= 0 || H <= 0 || (ndir != 1 && ndir != 2)) return ASRK_EINVAL;
    if (T == 0) return ASRK_OK;
    if (!gates || !whh_f || (ndir == 2 && !whh_r) || !C || !dY || !ws) return ASRK_EINVAL;
    if (H % 4 != 0) return ASRK_ESHAPE;
    if ((reinterpret_cast<uintptr_t>(gates) & 15) != 0) return ASRK_EINVAL;
    const int ncu = asrk_cu_count_();
    if (ncu <= 0) return ASRK_EDEVICE;
    BwdPlan pl = plan_bwd(B, H, ndir, ncu);
    if (!pl.ok) return ASRK_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    unsigned *flags = reinterpret_cast<unsigned *>(ws);
    ASRK_HIP(hipMemsetAsync(flags, 0, WS_MAX_FLAGS * sizeof(unsigned), s));

    RecBwdArgs a;
    a.G = gates; a.whh[0] = whh_f; a.whh[1] = ndir == 2 ? whh_r : whh_f;
    a.C = C; a.dY = dY; a.flags = flags; a.err = flags + WS_MAX_FLAGS;
    a.T = T; a.B = B; a.H = H; a.ndir = ndir; a.ldg = ndir * 4 * H; a.ldy = ndir * H;
    a.UB = pl.UB; a.nwg = pl.nwg; a.nbg = pl.nbg; a.BG = pl.BG; a.HPb = pl.HPb; a.KP = pl.KP;
    const int grid = ndir * pl.nbg * pl.nwg;
    asrk_prof_begin_(PROF_LSTM_BWD, s);
    int rc = ASRK_ESHAPE;
    if (pl.NT == 1) rc = launch_bwd<1>(a, grid, pl.lds, s);
    else if (pl.NT == 2) rc = launch_bwd<2>(a, grid, pl.lds, s);
    else if (pl.NT == 4) rc = launch_bwd<4>(a, grid, pl.lds, s);
    asrk_prof_end_(PROF_LSTM_BWD, s);
    return rc;
}

extern "C" int asrk_lstm_check_error(void *ws, void *stream) {
    if (!ws) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    unsigned e = 0;
    unsigned *errp = reinterpret_cast<unsigned *>(ws) + WS_MAX_FLAGS;
    ASRK_HIP(hipMemcpyAsync(&e, errp, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    ASRK_HIP(hipStreamSynchronize(s));
    if (e != 0) {
        ASRK_HIP(hipMemsetAsync(errp, 0, sizeof(unsigned), s));
        return ASRK_ETIMEOUT;
    }
    return ASRK_OK;
}
