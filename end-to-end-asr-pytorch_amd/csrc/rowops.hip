// HBM-bound helpers around the GEMM / recurrence kernels: strided 3-D copies (time-major <->
// batch-major, pyramid concat/drop of src/module.py:141-153), column sums (bias gradients),
// tanh epilogues (src/asr.py:280,290; src/module.py:155-156) and the row log-softmax of the CTC
// head / decoder (src/asr.py:96).  All are one-pass-per-byte streaming kernels: coalesced 16-B
// lanes, wave64 shuffles for the row reductions, no LDS round trips.
#include "common.h"
#include "knobs.h"

namespace {

template <bool VEC, bool ACC>
__global__ void copy3d_kernel(const float *__restrict__ src, float *__restrict__ dst, int n1, int n2,
                              int64_t ss0, int64_t ss1, int64_t ds0, int64_t ds1, int64_t total) {
    const int per_row = VEC ? (n2 >> 2) : n2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / per_row;
        const int e = (int)(i - row * per_row);
        const int64_t i0 = row / n1;
        const int i1 = (int)(row - i0 * n1);
        const float *s = src + i0 * ss0 + i1 * ss1;
        float *d = dst + i0 * ds0 + i1 * ds1;
        if (VEC) {
            f32x4 v = reinterpret_cast<const f32x4 *>(s)[e];
            if (ACC) {
                f32x4 o = reinterpret_cast<f32x4 *>(d)[e];
                v += o;
            }
            reinterpret_cast<f32x4 *>(d)[e] = v;
        } else {
            float v = s[e];
            if (ACC) v += d[e];
            d[e] = v;
        }
    }
}

// Wide rows (>= 64 vector elements): a thread owns one element column and walks `rpb` rows, so the
// (i0, i1) decomposition costs one 32-bit division per thread instead of two 64-bit ones per
// element (the generic kernel above moved the 65-MB pyramid copies at 1-1.8 TB/s), and 4 rows are
// in flight per lane.
template <bool ACC>
__global__ __launch_bounds__(256) void copy3d_rows_kernel(const float *__restrict__ src,
                                                          float *__restrict__ dst, int rows, int n1,
                                                          int per_row, int64_t ss0, int64_t ss1,
                                                          int64_t ds0, int64_t ds1, int rpb) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= per_row) return;
    int row = blockIdx.y * rpb;
    const int row_end = min(rows, row + rpb);
    int i0 = row / n1, i1 = row - i0 * n1;
    const f32x4 *sp = reinterpret_cast<const f32x4 *>(src + i0 * ss0 + i1 * ss1) + e;
    f32x4 *dp = reinterpret_cast<f32x4 *>(dst + i0 * ds0 + i1 * ds1) + e;
    const int64_t s_wrap = (ss0 - (int64_t)n1 * ss1) / 4, d_wrap = (ds0 - (int64_t)n1 * ds1) / 4;
    const int64_t s1 = ss1 / 4, d1 = ds1 / 4;
    while (row < row_end) {
        const f32x4 *sq[4];
        f32x4 *dq[4];
        f32x4 v[4];
        int n = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (row < row_end) {
                sq[u] = sp; dq[u] = dp; n = u + 1;
                sp += s1; dp += d1; ++row;
                if (++i1 == n1) { i1 = 0; sp += s_wrap; dp += d_wrap; }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (u < n) v[u] = *sq[u];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (u < n) {
            if (ACC) v[u] += *dq[u];
            *dq[u] = v[u];
        }
    }
}

// 64 columns x 4 row-lanes per block; grid.y row chunks; atomics combine chunks
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ X, int M, int N,
                                                     int ldx, float *__restrict__ out,
                                                     int rows_per_chunk) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = blockIdx.y * rows_per_chunk;
    const int r1 = min(M, r0 + rows_per_chunk);
    float s = 0.f;
    if (c < N) {
        for (int r = r0 + rl; r < r1; r += 4) s += X[(size_t)r * ldx + c];
    }
    part[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < N) {
        const int l = threadIdx.x;
        unsafeAtomicAdd(out + c, part[0][l] + part[1][l] + part[2][l] + part[3][l]);
    }
}

// vector variant (N % 4 == 0, ldx % 4 == 0, 16-B aligned): a lane owns 4 adjacent columns (one
// 16-B load per row), a wave 256 columns, the 4 waves of a block take rows r0+w, r0+w+4, ...;
// 4 independent accumulators keep 4 row loads in flight per lane (the scalar kernel above has one
// 4-B load in flight per lane and reads at ~1.2 TB/s)
__global__ __launch_bounds__(256) void colsum_vec_kernel(const float *__restrict__ X, int M, int N,
                                                         int ldx, float *__restrict__ out,
                                                         int rows_per_chunk) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ f4 part[4][64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = (blockIdx.x * 64 + lane) * 4;
    const int r0 = blockIdx.y * rows_per_chunk;
    const int r1 = min(M, r0 + rows_per_chunk);
    f4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (c < N) {
        const float *xp = X + (size_t)(r0 + w) * ldx + c;
        const size_t st = (size_t)4 * ldx;
        int r = r0 + w;
        for (; r + 12 < r1; r += 16, xp += 4 * st) {
            const f4 a = *reinterpret_cast<const f4 *>(xp);
            const f4 b = *reinterpret_cast<const f4 *>(xp + st);
            const f4 d = *reinterpret_cast<const f4 *>(xp + 2 * st);
            const f4 e = *reinterpret_cast<const f4 *>(xp + 3 * st);
            s0 += a; s1 += b; s2 += d; s3 += e;
        }
        for (; r < r1; r += 4, xp += st) s0 += *reinterpret_cast<const f4 *>(xp);
    }
    part[w][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0 && c < N) {
        const f4 t = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
#pragma unroll
        for (int j = 0; j < 4; ++j) unsafeAtomicAdd(out + c + j, t[j]);
    }
}

// narrow contiguous matrices (ldx == N, N/4 divides 256: N = 4..128): the vector kernel above would leave 1 - N/256 of
// every wave idle (a conv activation [512k x 64] summed at 0.6 TB/s).  Here X is a flat stream of 16-B pieces; a
// thread's pieces are 256 apart, so they all belong to column group tid % (N/4); 4 loads in flight per lane
__global__ __launch_bounds__(256) void colsum_narrow_kernel(const float *__restrict__ X, int64_t total4, int G,
                                                            float *__restrict__ out, int64_t per_block) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ f4 part[256];
    const int tid = threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.x * per_block, i1 = min(total4, i0 + per_block);
    const f4 *xp = reinterpret_cast<const f4 *>(X);
    f4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int64_t i = i0 + tid;
    for (; i + 1792 < i1; i += 2048) {
        const f4 a = xp[i], b = xp[i + 256], d = xp[i + 512], e = xp[i + 768];
        const f4 a2 = xp[i + 1024], b2 = xp[i + 1280], d2 = xp[i + 1536], e2 = xp[i + 1792];
        s0 += a; s1 += b; s2 += d; s3 += e;
        s0 += a2; s1 += b2; s2 += d2; s3 += e2;
    }
    for (; i < i1; i += 256) s0 += xp[i];
    part[tid] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int w = 128; w >= G; w >>= 1) {
        if (tid < w) part[tid] += part[tid + w];
        __syncthreads();
    }
    if (tid < G) {
        const f4 t = part[tid];
#pragma unroll
        for (int j = 0; j < 4; ++j) unsafeAtomicAdd(out + tid * 4 + j, t[j]);
    }
}

__global__ void tanh_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        y[i] = tanhf(x[i]);
}

__global__ void tanh_bwd_kernel(const float *__restrict__ y, const float *__restrict__ dy,
                                float *__restrict__ dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float t = y[i];
        dx[i] = dy[i] * (1.f - t * t);
    }
}

// one wave per row; 4 rows per block
template <bool VEC>
__global__ __launch_bounds__(256) void log_softmax_fwd_kernel(const float *__restrict__ x,
                                                              float *__restrict__ y, int rows,
                                                              int cols, int ld) {
    const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *xr = x + (size_t)row * ld;
    float *yr = y + (size_t)row * ld;
    float m = -INFINITY;
    if (VEC) {
        const int nv = cols >> 2;
        for (int i = lane; i < nv; i += 64) {
            f32x4 v = reinterpret_cast<const f32x4 *>(xr)[i];
            m = fmaxf(m, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
        }
        m = wave_max(m);
        float s = 0.f;
        for (int i = lane; i < nv; i += 64) {
            f32x4 v = reinterpret_cast<const f32x4 *>(xr)[i];
            s += expf(v[0] - m) + expf(v[1] - m) + expf(v[2] - m) + expf(v[3] - m);
        }
        s = wave_sum(s);
        const float lse = m + logf(s);
        for (int i = lane; i < nv; i += 64) {
            f32x4 v = reinterpret_cast<const f32x4 *>(xr)[i];
            v[0] -= lse; v[1] -= lse; v[2] -= lse; v[3] -= lse;
            reinterpret_cast<f32x4 *>(yr)[i] = v;
        }
    } else {
        for (int i = lane; i < cols; i += 64) m = fmaxf(m, xr[i]);
        m = wave_max(m);
        float s = 0.f;
        for (int i = lane; i < cols; i += 64) s += expf(xr[i] - m);
        s = wave_sum(s);
        const float lse = m + logf(s);
        for (int i = lane; i < cols; i += 64) yr[i] = xr[i] - lse;
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const float *__restrict__ y,
                                                              const float *__restrict__ dy,
                                                              float *__restrict__ dx, int rows,
                                                              int cols, int ld) {
    const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *yr = y + (size_t)row * ld;
    const float *gr = dy + (size_t)row * ld;
    float *dr = dx + (size_t)row * ld;
    float s = 0.f;
    if (VEC) {
        const int nv = cols >> 2;
        for (int i = lane; i < nv; i += 64) {
            f32x4 g = reinterpret_cast<const f32x4 *>(gr)[i];
            s += g[0] + g[1] + g[2] + g[3];
        }
        s = wave_sum(s);
        for (int i = lane; i < nv; i += 64) {
            f32x4 g = reinterpret_cast<const f32x4 *>(gr)[i];
            f32x4 v = reinterpret_cast<const f32x4 *>(yr)[i];
            g[0] -= expf(v[0]) * s; g[1] -= expf(v[1]) * s;
            g[2] -= expf(v[2]) * s; g[3] -= expf(v[3]) * s;
            reinterpret_cast<f32x4 *>(dr)[i] = g;
        }
    } else {
        for (int i = lane; i < cols; i += 64) s += gr[i];
        s = wave_sum(s);
        for (int i = lane; i < cols; i += 64) dr[i] = gr[i] - expf(yr[i]) * s;
    }
}

// top-k of each row (k small: beam widths), one wave per row.  Selection order is total:
// descending value, ascending index among equal values (so k = 1 is "first arg-max").  Iteration j
// finds the best element strictly after the (j-1)-th pick in that order; no marking, no sort.
__global__ __launch_bounds__(256) void topk_kernel(const float *__restrict__ x, int rows, int cols,
                                                   int ld, int k, float *__restrict__ vals,
                                                   int64_t *__restrict__ idxs) {
    const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *xr = x + (size_t)row * ld;
    float pv = INFINITY;
    int pi = -1;
    for (int j = 0; j < k; ++j) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = lane; i < cols; i += 64) {
            const float v = xr[i];
            const bool after = (v < pv) || (v == pv && i > pi);
            if (after && (v > bv || (v == bv && i < bi))) { bv = v; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) {
            vals[(size_t)row * k + j] = bv;
            idxs[(size_t)row * k + j] = bi == 0x7fffffff ? -1 : bi;
        }
        pv = bv;
        pi = bi;
    }
}

// The same selection for rows of up to 8192 columns with one 512-thread workgroup per row: the row is
// read ONCE into registers (16 elements per thread), every pick is a register scan + one workgroup
// arg-max (the one-wave kernel above re-reads the row k times: 313 us per call at 16 x 5000, k = 24 -
// half of a beam-search step).
constexpr int TK_T = 512, TK_E = 16, TK_NONE = 0x7fffffff;
__device__ __forceinline__ bool tk_better(float v, int i, float bv, int bi) {
    return bi == TK_NONE || v > bv || (v == bv && i < bi);
}
__global__ __launch_bounds__(TK_T) void topk_block_kernel(const float *__restrict__ x, int cols, int ld, int k,
                                                          float *__restrict__ vals,
                                                          int64_t *__restrict__ idxs) {
    __shared__ float s_v[2][TK_T / 64];
    __shared__ int s_i[2][TK_T / 64];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float *xr = x + (size_t)row * ld;
    float v[TK_E];
    unsigned live = 0;       // bit e: element tid + TK_T*e is selectable (in range, not NaN, not yet picked)
#pragma unroll
    for (int e = 0; e < TK_E; ++e) {
        const int i = tid + TK_T * e;
        v[e] = i < cols ? xr[i] : 0.f;
        if (i < cols && v[e] == v[e]) live |= 1u << e;
    }
    for (int j = 0; j < k; ++j) {
        float bv = -INFINITY;
        int bi = TK_NONE;
#pragma unroll
        for (int e = 0; e < TK_E; ++e)
            if (((live >> e) & 1u) && tk_better(v[e], tid + TK_T * e, bv, bi)) { bv = v[e]; bi = tid + TK_T * e; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (oi != TK_NONE && tk_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { s_v[j & 1][wave] = bv; s_i[j & 1][wave] = bi; }
        __syncthreads();
        bv = s_v[j & 1][0];
        bi = s_i[j & 1][0];
#pragma unroll
        for (int w = 1; w < TK_T / 64; ++w) {
            const float ov = s_v[j & 1][w];
            const int oi = s_i[j & 1][w];
            if (oi != TK_NONE && tk_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        if (bi != TK_NONE && (bi % TK_T) == tid) live &= ~(1u << (bi / TK_T));
        if (tid == 0) {
            vals[(size_t)row * k + j] = bi == TK_NONE ? -INFINITY : bv;
            idxs[(size_t)row * k + j] = bi == TK_NONE ? -1 : bi;
        }
    }
}

// The same selection by RADIX SELECT (k <= 64, up to 8192 columns, one 512-thread workgroup per row, the row in
// registers): the k-th largest key is found digit by digit (8 bits per pass over an LDS histogram, starting below the
// bits all selectable elements share), ties at the threshold go to the smallest indices (two more passes over the
// index), and the <= 64 winners are ranked in LDS.  ~8 rounds of barriers in all, against one workgroup-wide arg-max
// per pick in topk_block_kernel (1.7 us each: 45 us per call at 16 x 5000, k = 24 - two calls per beam-search step).
constexpr int RK_CAP = 64;
__device__ __forceinline__ unsigned tk_key(float v) {
    unsigned u = __float_as_uint(v);
    if (u == 0x80000000u) u = 0u;                       // -0 == +0
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // larger float <=> larger key
}
__global__ __launch_bounds__(TK_T) void topk_radix_kernel(const float *__restrict__ x, int cols, int ld, int k,
                                                          float *__restrict__ vals, int64_t *__restrict__ idxs) {
    __shared__ int hist[256];
    __shared__ unsigned s_mx[TK_T / 64], s_mn[TK_T / 64];
    __shared__ int s_cnt[TK_T / 64];
    __shared__ int s_digit, s_krem, c_n;
    __shared__ unsigned c_key[RK_CAP];
    __shared__ int c_idx[RK_CAP];
    __shared__ float c_val[RK_CAP];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float *xr = x + (size_t)row * ld;
    float v[TK_E];
    unsigned key[TK_E];
    unsigned live = 0;
    unsigned mx = 0u, mn = 0xFFFFFFFFu;
#pragma unroll
    for (int e = 0; e < TK_E; ++e) {
        const int i = tid + TK_T * e;
        v[e] = i < cols ? xr[i] : 0.f;
        key[e] = tk_key(v[e]);
        if (i < cols && v[e] == v[e]) {
            live |= 1u << e;
            mx = max(mx, key[e]);
            mn = min(mn, key[e]);
        }
    }
    int cnt = __popc(live);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mx = max(mx, (unsigned)__shfl_xor((int)mx, o, 64));
        mn = min(mn, (unsigned)__shfl_xor((int)mn, o, 64));
        cnt += __shfl_xor(cnt, o, 64);
    }
    if (lane == 0) { s_mx[wave] = mx; s_mn[wave] = mn; s_cnt[wave] = cnt; }
    if (tid == 0) c_n = 0;
    __syncthreads();
    mx = s_mx[0]; mn = s_mn[0]; cnt = s_cnt[0];
#pragma unroll
    for (int w = 1; w < TK_T / 64; ++w) { mx = max(mx, s_mx[w]); mn = min(mn, s_mn[w]); cnt += s_cnt[w]; }
    const int kk = min(k, cnt);                          // fewer selectable elements than k: all of them
    unsigned T = 0u;                                     // threshold key: the kk-th largest
    int krem = kk;                                       // ... of which `krem` are ties AT the threshold
    if (kk > 0) {
        const unsigned diff = mx ^ mn;
        const int nbits = diff ? 32 - __clz(diff) : 0;   // low bits in which the selectable keys differ
        const int top = (nbits + 7) / 8 * 8;             // the passes cover bits [0, top); the bits above are shared
        unsigned prefix = top >= 32 ? 0u : (mx >> top);
        for (int shift = top - 8; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
#pragma unroll
            for (int e = 0; e < TK_E; ++e)
                if (((live >> e) & 1u) && (shift + 8 >= 32 || (key[e] >> (shift + 8)) == prefix))
                    atomicAdd(&hist[(key[e] >> shift) & 255u], 1);
            __syncthreads();
            if (tid < 256) {
                int suf = 0;
                for (int j = tid; j < 256; ++j) suf += hist[j];
                const int nxt = suf - hist[tid];
                if (suf >= krem && nxt < krem) { s_digit = tid; s_krem = krem - nxt; }
            }
            __syncthreads();
            prefix = (prefix << 8) | (unsigned)s_digit;
            krem = s_krem;
        }
        T = nbits == 0 ? mx : prefix;
    }
    // ties at T: the krem smallest indices (index < 8192: digits idx >> 8, idx & 255)
    int Ti = 0x7fffffff;
    if (kk > 0) {
        int hi = 0;
        for (int pass = 0; pass < 2; ++pass) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
#pragma unroll
            for (int e = 0; e < TK_E; ++e) {
                const int i = tid + TK_T * e;
                if (((live >> e) & 1u) && key[e] == T && (pass == 0 || (i >> 8) == hi))
                    atomicAdd(&hist[pass == 0 ? (i >> 8) : (i & 255)], 1);
            }
            __syncthreads();
            if (tid < 256) {
                int pre = 0;
                for (int j = 0; j <= tid; ++j) pre += hist[j];
                const int prv = pre - hist[tid];
                if (pre >= krem && prv < krem) { s_digit = tid; s_krem = krem - prv; }
            }
            __syncthreads();
            if (pass == 0) hi = s_digit;
            else Ti = (hi << 8) | s_digit;
            krem = s_krem;
        }
    }
#pragma unroll
    for (int e = 0; e < TK_E; ++e) {
        const int i = tid + TK_T * e;
        if (((live >> e) & 1u) && kk > 0 && (key[e] > T || (key[e] == T && i <= Ti))) {
            const int pos = atomicAdd(&c_n, 1);
            if (pos < RK_CAP) { c_key[pos] = key[e]; c_idx[pos] = i; c_val[pos] = v[e]; }
        }
    }
    __syncthreads();
    if (tid < k) {
        if (tid < kk) {
            const unsigned mk = c_key[tid];
            const int mi = c_idx[tid];
            int rank = 0;
            for (int m = 0; m < kk; ++m) rank += (c_key[m] > mk || (c_key[m] == mk && c_idx[m] < mi)) ? 1 : 0;
            vals[(size_t)row * k + rank] = c_val[tid];
            idxs[(size_t)row * k + rank] = mi;
        } else {
            vals[(size_t)row * k + tid] = -INFINITY;
            idxs[(size_t)row * k + tid] = -1;
        }
    }
}

// fused softmax cross-entropy rows (CrossEntropyLoss(ignore_index) of bin/train_asr.py:47,130-131)
// one wave per row: lse_r = logsumexp(x_r); loss_r = lse_r - x_r[tgt]; sums[0]+=loss, sums[1]+=1
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float *__restrict__ x, int rows, int V,
                                                     int ld, const int64_t *__restrict__ tgt,
                                                     int ignore_index, float *__restrict__ row_lse,
                                                     float *__restrict__ sums) {
    const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *xr = x + (size_t)row * ld;
    float m = -INFINITY;
    for (int i = lane; i < V; i += 64) m = fmaxf(m, xr[i]);
    m = wave_max(m);
    float s = 0.f;
    for (int i = lane; i < V; i += 64) s += expf(xr[i] - m);
    s = wave_sum(s);
    const float lse = m + logf(s);
    if (lane == 0) {
        row_lse[row] = lse;
        const int64_t t = tgt[row];
        if (sums && t != ignore_index && t >= 0 && t < V) {   // sums == nullptr: ce_sum_kernel adds in a fixed order
            unsafeAtomicAdd(sums, lse - xr[t]);
            unsafeAtomicAdd(sums + 1, 1.0f);
        }
    }
}

// deterministic cross-entropy reduction: one workgroup, thread t adds rows t, t + 256, ... then a fixed LDS tree
__global__ __launch_bounds__(256) void ce_sum_kernel(const float *__restrict__ x, int rows, int V, int ld,
                                                     const int64_t *__restrict__ tgt, int ignore_index,
                                                     const float *__restrict__ row_lse, float *__restrict__ sums) {
    __shared__ float sl[256], sc[256];
    float a = 0.f, c = 0.f;
    for (int r = threadIdx.x; r < rows; r += 256) {
        const int64_t t = tgt[r];
        if (t != ignore_index && t >= 0 && t < V) { a += row_lse[r] - x[(size_t)r * ld + t]; c += 1.f; }
    }
    sl[threadIdx.x] = a;
    sc[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { sl[threadIdx.x] += sl[threadIdx.x + o]; sc[threadIdx.x] += sc[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { sums[0] = sl[0]; sums[1] = sc[0]; }
}

// dlogits = (softmax - onehot) * gscale for counted rows, 0 for ignored rows
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float *__restrict__ x, int rows, int V,
                                                     int ld, const int64_t *__restrict__ tgt,
                                                     int ignore_index,
                                                     const float *__restrict__ row_lse,
                                                     const float *__restrict__ gscale,
                                                     float *__restrict__ dx) {
    const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *xr = x + (size_t)row * ld;
    float *dr = dx + (size_t)row * ld;
    const int64_t t = tgt[row];
    const bool live = (t != ignore_index && t >= 0 && t < V);
    const float g = live ? gscale[0] : 0.f;
    const float lse = row_lse[row];
    for (int i = lane; i < V; i += 64) {
        float v = 0.f;
        if (live) v = (expf(xr[i] - lse) - (i == (int)t ? 1.f : 0.f)) * g;
        dr[i] = v;
    }
}

inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline unsigned grid_for(int64_t n, int block) {
    int64_t g = asrk_div_up64(n, block);
    if (g > 256 * 16) g = 256 * 16;  // grid-stride beyond ~16 blocks/CU
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

extern "C" int asrk_copy3d_f32(const float *src, float *dst, int n0, int n1, int n2,
                               int64_t ss0, int64_t ss1, int64_t ds0, int64_t ds1,
                               int accumulate, void *stream) {
    if (n0 < 0 || n1 < 0 || n2 < 0) return ASRK_EINVAL;
    if (n0 == 0 || n1 == 0 || n2 == 0) return ASRK_OK;
    if (!src || !dst) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = al16(src) && al16(dst) && (n2 % 4 == 0) && (ss0 % 4 == 0) && (ss1 % 4 == 0) &&
                     (ds0 % 4 == 0) && (ds1 % 4 == 0);
    const int64_t total = (int64_t)n0 * n1 * (vec ? n2 / 4 : n2);
    const unsigned grid = grid_for(total, 256);
    asrk_prof_begin_(PROF_ROWOPS, s);
    const int64_t rows64 = (int64_t)n0 * n1;
    if (vec && n2 / 4 >= 64 && rows64 < (1ll << 30)) {
        const int rows = (int)rows64, per_row = n2 / 4;
        const int gx = asrk_div_up(per_row, 256);
        int rpb = 16;                                   // rows per block: >= ~2k blocks when possible
        while (rpb > 4 && (int64_t)gx * asrk_div_up(rows, rpb) < 2048) rpb >>= 1;
        if (asrk_div_up(rows, rpb) > 65535) rpb = asrk_div_up(rows, 65535);
        const dim3 g(gx, asrk_div_up(rows, rpb));
        if (g.y <= 65535u) {
            if (accumulate)
                hipLaunchKernelGGL((copy3d_rows_kernel<true>), g, dim3(256), 0, s, src, dst, rows, n1,
                                   per_row, ss0, ss1, ds0, ds1, rpb);
            else
                hipLaunchKernelGGL((copy3d_rows_kernel<false>), g, dim3(256), 0, s, src, dst, rows, n1,
                                   per_row, ss0, ss1, ds0, ds1, rpb);
            asrk_prof_end_(PROF_ROWOPS, s);
            ASRK_LAUNCH_CHECK();
            return ASRK_OK;
        }
    }
    if (vec) {
        if (accumulate)
            hipLaunchKernelGGL((copy3d_kernel<true, true>), dim3(grid), dim3(256), 0, s, src, dst,
                               n1, n2, ss0, ss1, ds0, ds1, total);
        else
            hipLaunchKernelGGL((copy3d_kernel<true, false>), dim3(grid), dim3(256), 0, s, src, dst,
                               n1, n2, ss0, ss1, ds0, ds1, total);
    } else {
        if (accumulate)
            hipLaunchKernelGGL((copy3d_kernel<false, true>), dim3(grid), dim3(256), 0, s, src, dst,
                               n1, n2, ss0, ss1, ds0, ds1, total);
        else
            hipLaunchKernelGGL((copy3d_kernel<false, false>), dim3(grid), dim3(256), 0, s, src,
                               dst, n1, n2, ss0, ss1, ds0, ds1, total);
    }
    asrk_prof_end_(PROF_ROWOPS, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_colsum_f32(const float *X, int M, int N, int ldx, float *out, int accumulate,
                               void *stream) {
    if (M < 0 || N < 0 || ldx < N) return ASRK_EINVAL;
    if (N == 0) return ASRK_OK;
    if (!out || (!X && M > 0)) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) ASRK_HIP(hipMemsetAsync(out, 0, (size_t)N * 4, s));
    if (M == 0) return ASRK_OK;
    const bool vec = al16(X) && (N % 4 == 0) && (ldx % 4 == 0);
    if (vec && ldx == N && N <= 128 && 256 % (N / 4) == 0 && M >= 4096) {
        const int64_t total4 = (int64_t)M * (N / 4);
        // out[c] takes one atomic per block: 2048 blocks on 64 addresses took longer than the 131-MB read itself
        int blocks = asrk_knobs_().get(asrk_knobs_().deterministic, 0) ? 1 : 512;
        const int64_t per_block = asrk_div_up64(asrk_div_up64(total4, blocks), 2048) * 2048;
        blocks = (int)asrk_div_up64(total4, per_block);
        asrk_prof_begin_(PROF_ROWOPS, s);
        hipLaunchKernelGGL(colsum_narrow_kernel, dim3(blocks), dim3(256), 0, s, X, total4, N / 4, out, per_block);
        asrk_prof_end_(PROF_ROWOPS, s);
        ASRK_LAUNCH_CHECK();
        return ASRK_OK;
    }
    const int gx = asrk_div_up(N, vec ? 256 : 64);
    int chunks = asrk_div_up(vec ? 2048 : 1024, gx);
    if (chunks > asrk_div_up(M, 32)) chunks = asrk_div_up(M, 32);
    // deterministic: ONE row chunk per column block - the block's fixed-order sum is the only value added to out[c]
    if (chunks < 1 || asrk_knobs_().get(asrk_knobs_().deterministic, 0)) chunks = 1;
    const int rpc = asrk_div_up(M, chunks);
    chunks = asrk_div_up(M, rpc);
    asrk_prof_begin_(PROF_ROWOPS, s);
    if (vec)
        hipLaunchKernelGGL(colsum_vec_kernel, dim3(gx, chunks), dim3(256), 0, s, X, M, N, ldx, out, rpc);
    else
        hipLaunchKernelGGL(colsum_kernel, dim3(gx, chunks), dim3(256), 0, s, X, M, N, ldx, out, rpc);
    asrk_prof_end_(PROF_ROWOPS, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_tanh_fwd_f32(const float *x, float *y, int64_t n, void *stream) {
    if (n < 0) return ASRK_EINVAL;
    if (n == 0) return ASRK_OK;
    if (!x || !y) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tanh_fwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, x, y, n);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_tanh_bwd_f32(const float *y, const float *dy, float *dx, int64_t n,
                                 void *stream) {
    if (n < 0) return ASRK_EINVAL;
    if (n == 0) return ASRK_OK;
    if (!y || !dy || !dx) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, y, dy, dx, n);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_log_softmax_fwd_f32(const float *x, float *y, int rows, int cols, int ld,
                                        void *stream) {
    if (rows < 0 || cols <= 0 || ld < cols) return ASRK_EINVAL;
    if (rows == 0) return ASRK_OK;
    if (!x || !y) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = al16(x) && al16(y) && (cols % 4 == 0) && (ld % 4 == 0);
    const unsigned grid = (unsigned)asrk_div_up(rows, 4);
    asrk_prof_begin_(PROF_ROWOPS, s);
    if (vec)
        hipLaunchKernelGGL((log_softmax_fwd_kernel<true>), dim3(grid), dim3(256), 0, s, x, y, rows,
                           cols, ld);
    else
        hipLaunchKernelGGL((log_softmax_fwd_kernel<false>), dim3(grid), dim3(256), 0, s, x, y,
                           rows, cols, ld);
    asrk_prof_end_(PROF_ROWOPS, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_log_softmax_bwd_f32(const float *y, const float *dy, float *dx, int rows,
                                        int cols, int ld, void *stream) {
    if (rows < 0 || cols <= 0 || ld < cols) return ASRK_EINVAL;
    if (rows == 0) return ASRK_OK;
    if (!y || !dy || !dx) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = al16(y) && al16(dy) && al16(dx) && (cols % 4 == 0) && (ld % 4 == 0);
    const unsigned grid = (unsigned)asrk_div_up(rows, 4);
    asrk_prof_begin_(PROF_ROWOPS, s);
    if (vec)
        hipLaunchKernelGGL((log_softmax_bwd_kernel<true>), dim3(grid), dim3(256), 0, s, y, dy, dx,
                           rows, cols, ld);
    else
        hipLaunchKernelGGL((log_softmax_bwd_kernel<false>), dim3(grid), dim3(256), 0, s, y, dy, dx,
                           rows, cols, ld);
    asrk_prof_end_(PROF_ROWOPS, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_cross_entropy_fwd_f32(const float *logits, int rows, int V, int ld,
                                          const int64_t *targets, int ignore_index, float *row_lse,
                                          float *sums, void *stream) {
    if (rows < 0 || V <= 0 || ld < V) return ASRK_EINVAL;
    if (!sums) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ASRK_HIP(hipMemsetAsync(sums, 0, 2 * sizeof(float), s));
    if (rows == 0) return ASRK_OK;
    if (!logits || !targets || !row_lse) return ASRK_EINVAL;
    asrk_prof_begin_(PROF_ROWOPS, s);
    // ASRK_DETERMINISTIC: no atomics in the row kernel; one workgroup adds the row losses in a fixed order
    const bool det = asrk_knobs_().get(asrk_knobs_().deterministic, 0) != 0;
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(asrk_div_up(rows, 4)), dim3(256), 0, s, logits, rows, V, ld,
                       targets, ignore_index, row_lse, det ? nullptr : sums);
    if (det)
        hipLaunchKernelGGL(ce_sum_kernel, dim3(1), dim3(256), 0, s, logits, rows, V, ld, targets, ignore_index,
                           row_lse, sums);
    asrk_prof_end_(PROF_ROWOPS, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_cross_entropy_bwd_f32(const float *logits, int rows, int V, int ld,
                                          const int64_t *targets, int ignore_index,
                                          const float *row_lse, const float *gscale, float *dlogits,
                                          void *stream) {
    if (rows < 0 || V <= 0 || ld < V) return ASRK_EINVAL;
    if (rows == 0) return ASRK_OK;
    if (!logits || !targets || !row_lse || !gscale || !dlogits) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_ROWOPS, s);
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(asrk_div_up(rows, 4)), dim3(256), 0, s, logits, rows, V, ld,
                       targets, ignore_index, row_lse, gscale, dlogits);
    asrk_prof_end_(PROF_ROWOPS, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_topk_f32(const float *x, int rows, int cols, int ld, int k, float *values,
                             int64_t *indices, void *stream) {
    if (rows < 0 || cols <= 0 || ld < cols || k <= 0 || k > cols) return ASRK_EINVAL;
    if (rows == 0) return ASRK_OK;
    if (!x || !values || !indices) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_ROWOPS, s);
    if (cols <= TK_T * TK_E && k <= RK_CAP)
        hipLaunchKernelGGL(topk_radix_kernel, dim3((unsigned)rows), dim3(TK_T), 0, s, x, cols, ld, k, values, indices);
    else if (cols <= TK_T * TK_E)
        hipLaunchKernelGGL(topk_block_kernel, dim3((unsigned)rows), dim3(TK_T), 0, s, x, cols, ld, k, values,
                           indices);
    else
        hipLaunchKernelGGL(topk_kernel, dim3((unsigned)asrk_div_up(rows, 4)), dim3(256), 0, s, x, rows, cols,
                           ld, k, values, indices);
    asrk_prof_end_(PROF_ROWOPS, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}
