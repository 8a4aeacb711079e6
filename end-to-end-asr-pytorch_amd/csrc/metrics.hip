// Validation read-out on the device (reference: bin/train_asr.py:169-217 -> src/util.py:113-127 cal_er,
// src/text.py:61-71 decode(ignore_repeat)): the hypothesis crop every reference text encoder applies to a
// row of arg-max token ids (stop at <eos>, drop <pad>, merge CTC repeats) and the Levenshtein distance the
// reference takes from the third-party `editdistance` package (absent here; its published algorithm is the
// classic unit-cost dynamic programme).  Both are batch kernels: one wave per utterance, nothing but the
// compacted ids / the B distances ever crosses PCIe.  Integer work, HBM / latency bound; bit-exact.
#include "common.h"

namespace {

// One wave per row.  ids [B, ld] int64 -> out [B, ld_out] (compacted, rest untouched), out_len [B].
// keep(t) = before the first <eos>  &&  id != pad  &&  !(ignore_repeat && t > 0 && id == ids[t-1])
// (the RAW previous element, pads included: src/text.py:65 compares with idxs[t-1]).
__global__ __launch_bounds__(256) void token_crop_kernel(const int64_t *__restrict__ ids, int64_t ld, int B, int T,
                                                         int64_t pad, int64_t eos, int ignore_repeat,
                                                         int64_t *__restrict__ out, int64_t ld_out,
                                                         int32_t *__restrict__ out_len) {
    const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const int64_t *x = ids + (int64_t)row * ld;
    int64_t *o = out + (int64_t)row * ld_out;
    int n = 0;                                         // kept so far (wave-uniform)
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        const bool in = t < T;
        const int64_t v = in ? x[t] : pad;
        const int64_t pv = (in && t > 0) ? x[t - 1] : -1;
        const unsigned long long eos_mask = __ballot(in && v == eos);
        const int first_eos = eos_mask ? __builtin_ctzll(eos_mask) : 64;
        const bool keep = in && lane < first_eos && v != pad && !(ignore_repeat && t > 0 && v == pv);
        const unsigned long long km = __ballot(keep);
        if (keep) o[n + __builtin_popcountll(km & ((1ull << lane) - 1ull))] = v;
        n += __builtin_popcountll(km);
        if (eos_mask) break;
    }
    if (lane == 0) out_len[row] = n;
}

// Levenshtein distance of B independent pairs; one wave per pair, DP rows in wave-private LDS.
//   cur[j] = min(prev[j] + 1, prev[j-1] + (a_i != b_j), cur[j-1] + 1)
// The cur[j-1] term is a min-plus prefix scan: with u_j = min(prev[j] + 1, prev[j-1] + cost) - j,
// cur[j] = j + min(carry, min_{k <= j} u_k), carry = cur[chunk start - 1] - (chunk start - 1).
constexpr int ED_MAXB = 4096;                          // longest second sequence (LDS: 2 rows x 4 waves x 16 KiB)

__global__ __launch_bounds__(256) void edit_distance_kernel(const int64_t *__restrict__ a, int64_t lda,
                                                            const int32_t *__restrict__ a_len,
                                                            const int64_t *__restrict__ b, int64_t ldb,
                                                            const int32_t *__restrict__ b_len, int B, int rowlen,
                                                            int32_t *__restrict__ dist) {
    extern __shared__ int ed_rows[];                   // [4 waves][2][rowlen]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= B) return;
    const int la = a_len[pair], lb = b_len[pair];
    const int64_t *pa = a + (int64_t)pair * lda, *pb = b + (int64_t)pair * ldb;
    int *r0 = ed_rows + (size_t)wave * 2 * rowlen, *r1 = r0 + rowlen;
    for (int j = lane; j <= lb; j += 64) r0[j] = j;    // row 0: distance from the empty prefix
    int *prev = r0, *cur = r1;
    for (int i = 1; i <= la; ++i) {
        const int64_t ai = pa[i - 1];
        int carry = i;                                 // cur[0] - 0
        for (int j0 = 1; j0 <= lb; j0 += 64) {
            const int j = j0 + lane;
            int u = 0x3fffffff;
            if (j <= lb) u = min(prev[j] + 1, prev[j - 1] + (ai != pb[j - 1] ? 1 : 0)) - j;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(u, o, 64);
                if (lane >= o) u = min(u, v);
            }
            u = min(u, carry);
            if (j <= lb) cur[j] = u + j;
            carry = __shfl(u, 63, 64);                 // lanes past lb hold the running minimum as well
        }
        if (lane == 0) cur[0] = i;
        int *t = prev; prev = cur; cur = t;
    }
    if (lane == 0) dist[pair] = prev[lb];
}

}  // namespace

extern "C" int asrk_token_crop_i64(const int64_t *ids, int64_t ld, int B, int T, int64_t pad_idx, int64_t eos_idx,
                                   int ignore_repeat, int64_t *out, int64_t ld_out, int32_t *out_len,
                                   void *stream) {
    if (B < 0 || T < 0 || ld < T || ld_out < T) return ASRK_EINVAL;
    if (B == 0) return ASRK_OK;
    if (!out_len || (T > 0 && (!ids || !out))) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(token_crop_kernel, dim3(asrk_div_up(B, 4)), dim3(256), 0, s, ids, ld, B, T, pad_idx, eos_idx,
                       ignore_repeat, out, ld_out, out_len);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_edit_distance_i64(const int64_t *a, int64_t lda, const int32_t *a_len, const int64_t *b,
                                      int64_t ldb, const int32_t *b_len, int B, int max_b_len, int32_t *dist,
                                      void *stream) {
    if (B < 0 || max_b_len < 0 || lda < 0 || ldb < max_b_len) return ASRK_EINVAL;
    if (B == 0) return ASRK_OK;
    if (!a_len || !b_len || !dist || (!a && lda > 0) || (!b && max_b_len > 0)) return ASRK_EINVAL;
    if (max_b_len > ED_MAXB) return ASRK_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int rowlen = max_b_len + 1;
    const size_t lds = (size_t)4 * 2 * rowlen * sizeof(int);
    // per call (per-device attribute; no latched flag that a second GPU or thread would trip over)
    ASRK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(edit_distance_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * (ED_MAXB + 1) * 4));
    hipLaunchKernelGGL(edit_distance_kernel, dim3(asrk_div_up(B, 4)), dim3(256), lds, s, a, lda, a_len, b, ldb, b_len,
                       B, rowlen, dist);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}
