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
#include "common.h"
#include "knobs.h"
#include <algorithm>
#include <cstdlib>

extern "C" int asrk_cu_count_(void);
extern "C" size_t asrk_split_panel_stride_(int rows, int K);   // gemm_split.hip: bytes between 64-row blocks of a panel

namespace {

constexpr int WS_WORDS = 1024;
constexpr unsigned SENT = 0xFFFFFFFFu;                       // NaN payload used as "not written yet"
constexpr unsigned long long TIMEOUT_TICKS = 300000000ull;  // 3 s of the 100 MHz wall clock

struct RecFwdArgs {
    float *G;
    const float *whh[2];
    float *Y, *C;
    float *X;  // exchange buffer [ndir*nbg][T][kgp][NT][16][16], sentinel-initialised
    unsigned *err;
    int T, B, H, ndir, ldg, ldy;
    int U, nwg, nbg, BG, HP, kgp, canw, poll_mode;
    int dir0, bg0;  // this launch covers directions [dir0, dir0+ndir) and batch groups [bg0, bg0+nbg)
    unsigned long long *dbg;  // optional phase timeline [steps][4 waves][8 phases] (debug only)
    int dbg_steps;
    // optional second copy of the output in the layout the NEXT layer consumes (time reduction of
    // src/module.py:141-153 fused into the store): mode 1 'concat' -> Y2[t/r][b][(t%r)*ldy + col] for
    // t < (T/r)*r; mode 2 'drop' -> Y2[t/r][b][col] for t % r == 0
    float *Y2;
    int pyr_mode, pyr_rate;
    // optional per-row sequence lengths (inference, batched beam-search encoder): row b runs steps s < lens[b] only,
    // the reverse direction starts at ITS last frame (t = lens[b] - 1 - s) - what the reference computes when it
    // encodes the utterance alone, unpadded (bin/test_asr.py / src/decode.py:88 run batch 1).  Frames t >= lens[b]
    // of Y / Y2 / G / C are not written (the caller zero-fills Y).  nullptr: every row runs all T steps (training).
    const int64_t *lens;
    int rearm;   // ASRK_REC_REARM: every workgroup refills its share of region s - 2 with the sentinel at step s
    // optional (bf16x6 kernel only): the output ALSO as the row-major split panel of the next layer's input
    // [rows = (t / r, b)][K = r * ldy] (pyr_mode 1) or [rows = (t, b)][K = ldy] (pyr_mode 0) - csrc/gemm_split.hip
    // layout: piece (row block, chunk column, plane) of [64 rows][8 bf16] - so that layer's input projection and
    // needs no split pass over this tensor: the quad's 8-byte plane stores of the exchange, once more
    unsigned char *P2;
    size_t p2_stride;   // bytes between 64-row blocks of the panel
};

struct RecBwdArgs {
    float *G;
    const float *whh[2];
    const float *C, *dY;
    float *X;  // exchange buffer [ndir*nbg][T][4 gates][kgp][NT][16][16]
    unsigned *err;
    int T, B, H, ndir, ldg, ldy;
    int UB, nwg, nbg, BG, HPb, KP, kgp, canw, poll_mode;
    int dir0, bg0;
    unsigned long long *dbg;
    int dbg_steps;
    float *db;   // optional [ndir][4H] bias gradient (sum of dG over t and batch), accumulated in-kernel
    int pyr_mode, pyr_rate;   // dY is given in the time-reduced layout of RecFwdArgs::Y2 (0: plain [T*B, ldy])
    int rearm;   // ASRK_REC_REARM (see RecFwdArgs)
    // optional (bf16x6 LSTM kernel only): dG ALSO as the row-major split panel [rows = (t, b)][K = ldg] that the
    // input-gradient GEMM dX = dG W_ih multiplies: the staged 16-byte exchange chunks, once more
    unsigned char *PG;
    size_t pg_stride;
    // optional (same kernel): dG^T ALSO as the split panel [rows = gate columns][K = tokens (t, b)] that the weight
    // gradients dW_ih = dG^T X, dW_hh = dG^T H_prev multiply (B % 16 == 0): a second, transposed staging image,
    // one more workgroup barrier in the tail of the step, 384 16-byte stores per workgroup
    unsigned char *PT;
    size_t pt_stride;
};

// debug timeline: wave-lane-0 of workgroup 0 stamps the shader clock at phase boundaries
#define REC_STAMP_W(ph)                                                                       \
    do {                                                                                      \
        if (p.dbg && blockIdx.x == 0 && lane == 0 && s < p.dbg_steps && wave < 4)             \
            p.dbg[((size_t)s * 4 + wave) * 8 + (ph)] = __builtin_readcyclecounter();          \
    } while (0)
#define REC_STAMP(ph)                                                                         \
    do {                                                                                      \
        if (p.dbg && blockIdx.x == 0 && lane == 0 && s < p.dbg_steps)                         \
            p.dbg[((size_t)s * 4 + wave) * 8 + (ph)] = __builtin_readcyclecounter();          \
    } while (0)

__device__ __forceinline__ bool has_sentinel(const f32x4 &v) {
    const u32x4 u = __builtin_bit_cast(u32x4, v);
    return (u[0] == SENT) | (u[1] == SENT) | (u[2] == SENT) | (u[3] == SENT);
}

__device__ __forceinline__ bool any_nan(const f32x4 &v) {
    return (v[0] != v[0]) | (v[1] != v[1]) | (v[2] != v[2]) | (v[3] != v[3]);
}

// bounded-spin bookkeeping shared by the poll loops; returns false when the wave must give up
__device__ __forceinline__ bool spin_ok(unsigned &spins, unsigned long long &t0, unsigned *err,
                                        int lane) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 127u) == 0) {
        const unsigned long long now = wall_clock64();
        if (t0 == 0) t0 = now;
        const unsigned e = __hip_atomic_load(err, RLX_AGENT);
        if (e != 0 || now - t0 > TIMEOUT_TICKS) {
            if (lane == 0) __hip_atomic_store(err, 1u, RLX_AGENT);
            return false;
        }
    }
    return true;
}

// Wait until none of the `cnt` canary words at `cb` is the sentinel.  The polls are PIPELINED:
// three relaxed sc1 loads are kept in flight a few hundred cycles apart and examined in order, so
// a canary is noticed ~one poll spacing after it becomes visible instead of up to a whole extra
// memory round trip later (a poll costs 1.5-2k cycles under load; the hand-off is the critical
// path of every recurrence step).
__device__ __forceinline__ bool wait_canaries(const unsigned *cb, int cnt, unsigned *err, int lane,
                                              int mode) {
    const bool pipelined = mode & 1;
    unsigned spins = 0;
    unsigned long long t0 = 0;
    if (cnt > 128 || !pipelined) {   // many producers, or pipelining disabled: plain loop
        for (;;) {
            bool good = true;
            for (int j = lane; j < cnt; j += 64) good &= (__hip_atomic_load(cb + j, RLX_AGENT) != SENT);
            if (__all(good)) return true;
            if (!spin_ok(spins, t0, err, lane)) return false;
            if (mode & 4) __builtin_amdgcn_s_sleep(2);      // experiment: back off between polls
            if (mode & 8) __builtin_amdgcn_s_sleep(8);
        }
    }
    const bool a0 = lane < cnt, a1 = lane + 64 < cnt;
    const unsigned *p0 = cb + (a0 ? lane : 0), *p1 = cb + (a1 ? lane + 64 : 0);
    unsigned x0 = a0 ? __hip_atomic_load(p0, RLX_AGENT) : 0u, y0 = a1 ? __hip_atomic_load(p1, RLX_AGENT) : 0u;
    __builtin_amdgcn_s_sleep(3);
    unsigned x1 = a0 ? __hip_atomic_load(p0, RLX_AGENT) : 0u, y1 = a1 ? __hip_atomic_load(p1, RLX_AGENT) : 0u;
    __builtin_amdgcn_s_sleep(3);
    unsigned x2 = a0 ? __hip_atomic_load(p0, RLX_AGENT) : 0u, y2 = a1 ? __hip_atomic_load(p1, RLX_AGENT) : 0u;
    for (;;) {
        const bool good = (x0 != SENT) & (y0 != SENT);   // waits for the OLDEST poll only
        if (__all(good)) return true;
        x0 = x1; y0 = y1; x1 = x2; y1 = y2;
        if (!spin_ok(spins, t0, err, lane)) return false;
        __builtin_amdgcn_s_sleep(2);
        x2 = a0 ? __hip_atomic_load(p0, RLX_AGENT) : 0u;
        y2 = a1 ? __hip_atomic_load(p1, RLX_AGENT) : 0u;
    }
}

// k-groups in the backward fragment ring: 8 loads (1 KiB each) in flight per wave at NT == 1 (the
// prologue that primes the ring sits on the serial chain, ~60 cycles per load), 16 at NT >= 2 where
// a k-group carries NT loads and 4*NT MFMAs (measured both ways at H=512/NT=1 and H=1024/NT=2).
// (The register-resident variant, RK > 0, was also measured with 16 k-groups in flight at H = 1024:
// 175 instead of 171 cycles per k-group, so the loop is not bound by fragment latency x ring depth.)
__host__ __device__ constexpr int bwd_ring_kgroups(int NT, int RK = 0) { return NT == 1 ? 8 : 16 / NT; }


// ASRK_REC_REARM: the kernel hands the exchange buffer back ARMED, so the next launch on it needs no fill pass
// (the eight sentinel fills of a cfg3 training step were 1.0 ms of 4 TB/s stores in front of latency-bound kernels).
// At loop step s a workgroup reads region s - 1 and publishes region s.  Once it is past the partial-sum barrier
// of step s, its four waves together have seen the step-(s - 1) canaries of EVERY producer of the group, and a
// producer publishes step s - 1 only after its own reads of region s - 2 have been consumed by its MFMAs: nobody
// will touch region s - 2 again in this launch.  Each workgroup then overwrites its 1/nwg share of it (1.5-6 KiB:
// one or two 16-byte stores per thread, issued in the tail of the step).  Regions T - 2 and T - 1 are left to a
// small fill behind the launch (sentinel_fill_tail); a launch that aborts (hand-off timeout) leaves the buffer dirty,
// the host drops it.  The next launch sees the sentinels through the kernel-boundary release / acquire like those of
// a fill kernel.
__device__ __forceinline__ void rearm_region(float *region, size_t floats, int wg, int nwg, int tid, int nthreads) {
    const unsigned n16 = (unsigned)(floats >> 2);
    const unsigned per = (n16 + (unsigned)nwg - 1u) / (unsigned)nwg;
    const unsigned lo = (unsigned)wg * per, hi = min(lo + per, n16);
    u32x4 *q = reinterpret_cast<u32x4 *>(region);
    const u32x4 v = {SENT, SENT, SENT, SENT};
    for (unsigned i = lo + (unsigned)tid; i < hi; i += (unsigned)nthreads) q[i] = v;
}

__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));  // v_rcp_f32 (1 ulp), not the IEEE divide sequence
}
__device__ __forceinline__ float fast_tanh(float x) {
    // 1 - 2/(exp(2x)+1); exact limits for |x| large (exp -> inf/0), abs error ~1e-7
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f);
}

// Partial-sum buffer: one f32x4 per MFMA lane and 64-lane block, padded to 72 entries (2 after every
// 16 lanes).  The cell update reads it with the 4 units of a batch row on adjacent lanes, i.e. 16
// entries = 256 B apart: unpadded that is a 4-way bank conflict on every read (SQ_LDS_BANK_CONFLICT
// was ~48 % of the LDS-active cycles of both kernels), with the padding 8 consecutive lanes cover
// 8 different 16-B bank groups.
constexpr int RED_PITCH = 72;
__device__ __forceinline__ int red_slot(int lane) { return lane + 2 * (lane >> 4); }

// A dependent v_mfma_f32_16x16x4_f32 (same accumulator) can only issue ~90 cycles after its
// predecessor, an independent one after 32: every wave therefore rotates over >= 4 accumulator
// chains (measured on the backward kernel with 2 chains: 46 cycles per MFMA instead of 32).
template <int ACC>
__device__ __forceinline__ f32x4 acc_sum(const f32x4 (&a)[ACC]) {
    f32x4 r = a[0];
#pragma unroll
    for (int i = 1; i < ACC; ++i) r += a[i];
    return r;
}

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
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split3(float a, unsigned &b0, unsigned &b1, unsigned &b2) {
    const __bf16 h0 = (__bf16)a;
    const float r1 = a - (float)h0;
    const __bf16 h1 = (__bf16)r1;
    const __bf16 h2 = (__bf16)(r1 - (float)h1);
    b0 = __builtin_bit_cast(unsigned short, h0);
    b1 = __builtin_bit_cast(unsigned short, h1);
    b2 = __builtin_bit_cast(unsigned short, h2);
}

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
        if (lane == 0)
            __hip_atomic_store(reinterpret_cast<unsigned *>(xstep + data_floats) + 4 * wg + wave,
                               (unsigned)(s + 1), RLX_AGENT);
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
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (ch_on[i]) {
                    const u32x4 v = *reinterpret_cast<const u32x4 *>(stage + ch_src[i]);
                    __builtin_amdgcn_raw_buffer_store_b128(v, xrs, ch_dst[i], 0, 16);
                    if (!GRU && p.PG) {   // the same chunk = 8 units of one gate and row, three planes apart: dG's A panel
                        const int m = t * p.B + pg_row[i];
                        *reinterpret_cast<u32x4 *>(p.PG + (size_t)(m >> 6) * p.pg_stride + pg_col[i] + (m & 63) * 16) = v;
                    }
                }
        }
        if (lane == 0)
            __hip_atomic_store(reinterpret_cast<unsigned *>(xstep + data_floats) + 4 * wg + wave,
                               (unsigned)(s + 1), RLX_AGENT);
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

struct FwdPlan {
    int MT, NT, KGW, U, nwg, nbg, BG, HP, kgp, db;
    size_t lds, xfloats;
    bool ok;
    int ndir_l, nbg_l;   // directions / batch groups per launch (== ndir, nbg when one launch suffices)
    int bf;              // 1: lstm_rec_fwd_bf_kernel (bf16x6 operand splitting), lds / xfloats are that kernel's
};

// When (directions x batch groups x unit slices) exceeds the CU count the independent groups are run
// as several launches of the same persistent kernel: all directions or one, and as many batch groups
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
    // tuning overrides (experiments): ASRK_FWD_MT / ASRK_FWD_NT force a tile, ASRK_WG_PER_CU > 1
    // lets the grid oversubscribe the CUs (co-resident workgroups hide each other's latency)
    const int oc = kn.get(kn.wg_per_cu, 1);
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

struct BwdPlan {
    int NT, UB, nwg, nbg, BG, HPb, KP, kgp;
    size_t lds, xfloats;
    bool ok;
    int ndir_l, nbg_l;
    int RK;   // k-groups of every wave's W_hh^T slice kept in VGPRs instead of LDS (0 or 32)
    int bf;   // 1: lstm_rec_bwd_bf_kernel (bf16x6 operand splitting); lds / xfloats are that kernel's
};

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
    if (asrk_knobs_().get(asrk_knobs_().bwd_rk, 1) == 0) return 1;
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
    const int oc = kn.get(kn.wg_per_cu, 1);
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
                const int BG = (kn.is_set(kn.bwd_bg) && NT == 1) ? kn.bwd_bg : 16 * NT;  // experiment: half-filled tile
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

template <int NT, int RK, bool GRU>
int launch_bwd(const RecBwdArgs &a, int grid, size_t lds, hipStream_t s) {
    auto kern = lstm_rec_bwd_kernel<NT, RK, GRU>;
    ASRK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
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

template <bool GRU>
int launch_bwd_plan(const RecBwdArgs &a, const BwdPlan &pl, int grid, hipStream_t s) {
    if (pl.bf) return a.H == 1024 ? launch_bwd_bf<GRU, 32, 21>(a, grid, pl.lds, s)
                                  : launch_bwd_bf<GRU, 16, 16>(a, grid, pl.lds, s);
    if (pl.NT == 1 && pl.RK == 32) return launch_bwd<1, 32, GRU>(a, grid, pl.lds, s);
    if (pl.NT == 1) return launch_bwd<1, 0, GRU>(a, grid, pl.lds, s);
    if (pl.NT == 2) return launch_bwd<2, 0, GRU>(a, grid, pl.lds, s);
    if (pl.NT == 4) return launch_bwd<4, 0, GRU>(a, grid, pl.lds, s);
    return ASRK_ESHAPE;
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
    a.poll_mode |= (kn.get(kn.fwd_presleep, 16) & 0xff) << 8;  // x64 cycles
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
            if (pl.bf) rc = gru ? launch_fwd_bf_plan<true>(a, pl, H, grid, s) : launch_fwd_bf_plan<false>(a, pl, H, grid, s);
            else rc = gru ? launch_fwd_plan<true>(a, pl, grid, s) : launch_fwd_plan<false>(a, pl, grid, s);
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
            rc = gru ? launch_bwd_plan<true>(a, pl, grid, s) : launch_bwd_plan<false>(a, pl, grid, s);
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
