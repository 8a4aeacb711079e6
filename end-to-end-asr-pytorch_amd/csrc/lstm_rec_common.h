// Shared by the translation units of the persistent recurrence (lstm_rec.hip: plans + C entry points;
// lstm_rec_{fwd,bwd}_{f32,bf}.hip: one kernel family each): argument blocks, plan records, the sentinel / canary
// hand-off helpers and the per-family launch functions.  Round 6 split one 2 500-line unit into five so that they
// compile in parallel and a change to one kernel rebuilds one object.
#pragma once
#include "common.h"
#include "knobs.h"
#include <algorithm>
#include <cstdlib>

namespace asrk_rec {

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

struct FwdPlan {
    int MT, NT, KGW, U, nwg, nbg, BG, HP, kgp, db;
    size_t lds, xfloats;
    bool ok;
    int ndir_l, nbg_l;   // directions / batch groups per launch (== ndir, nbg when one launch suffices)
    int bf;              // 1: lstm_rec_fwd_bf_kernel (bf16x6 operand splitting), lds / xfloats are that kernel's
};

// When (directions x batch groups x unit slices) exceeds the CU count the independent groups are run
// as several launches of the same persistent kernel: all directions or one, and as many batch groups

struct BwdPlan {
    int NT, UB, nwg, nbg, BG, HPb, KP, kgp;
    size_t lds, xfloats;
    bool ok;
    int ndir_l, nbg_l;
    int RK;   // k-groups of every wave's W_hh^T slice kept in VGPRs instead of LDS (0 or 32)
    int bf;   // 1: lstm_rec_bwd_bf_kernel (bf16x6 operand splitting); lds / xfloats are that kernel's
};


// one launch function per kernel family (defined beside the kernels): picks the instantiation the plan names
int launch_fwd_f32(bool gru, const RecFwdArgs &a, const FwdPlan &pl, int grid, hipStream_t s);
int launch_fwd_bf(bool gru, const RecFwdArgs &a, const FwdPlan &pl, int H, int grid, hipStream_t s);
int launch_bwd_f32(bool gru, const RecBwdArgs &a, const BwdPlan &pl, int grid, hipStream_t s);
int launch_bwd_bf(bool gru, const RecBwdArgs &a, const BwdPlan &pl, int grid, hipStream_t s);

}  // namespace asrk_rec
