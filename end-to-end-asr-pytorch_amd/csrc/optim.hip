// Fused optimiser updates (bin/train_asr.py:115-137 + src/solver.py:76-91: clip_grad_norm_ followed by
// torch.optim.<Adadelta|Adam>.step()).  torch runs these as ~8 multi-tensor passes over parameters,
// gradients and state; here each parameter is updated in ONE streaming pass (read p, g, 2 state
// tensors; write p and the state), with the gradient-clipping coefficient folded in as a device
// scalar so the separate "scale all gradients" pass disappears.  Arithmetic order follows
// torch/optim/adadelta.py / adam.py (single-tensor path, weight_decay = 0, maximize = False).
#include "common.h"
#include <algorithm>

namespace {

__device__ __forceinline__ float clip_of(const float *coef) { return coef ? fminf(*coef, 1.0f) : 1.0f; }
// The NaN guard of src/solver.py:85-89 (`if math.isnan(grad_norm): skip the step`) as a device-side predicate:
// a NaN gradient norm makes the coefficient max_norm / (norm + 1e-6) NaN, and a kernel that sees a NaN
// coefficient leaves parameters and state untouched - the host never has to read the norm back to decide.
__device__ __forceinline__ bool skip_of(const float *coef) { return coef && (*coef != *coef); }

template <bool VEC>
__global__ __launch_bounds__(256) void adadelta_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                       float *__restrict__ sq, float *__restrict__ acc,
                                                       int64_t n, float lr, float rho, float omr,
                                                       float eps, const float *coef) {
    if (skip_of(coef)) return;
    const float c = clip_of(coef);
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    if (VEC) {
        const int64_t nv = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += step) {
            f32x4 pv = reinterpret_cast<f32x4 *>(p)[i];
            const f32x4 gv = reinterpret_cast<const f32x4 *>(g)[i];
            f32x4 sv = reinterpret_cast<f32x4 *>(sq)[i], av = reinterpret_cast<f32x4 *>(acc)[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gr = gv[j] * c;
                sv[j] = sv[j] * rho + omr * gr * gr;
                const float stdv = sqrtf(sv[j] + eps);
                const float delta = sqrtf(av[j] + eps) / stdv * gr;
                av[j] = av[j] * rho + omr * delta * delta;
                pv[j] = pv[j] - lr * delta;
            }
            reinterpret_cast<f32x4 *>(p)[i] = pv;
            reinterpret_cast<f32x4 *>(sq)[i] = sv;
            reinterpret_cast<f32x4 *>(acc)[i] = av;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
            const float gr = g[i] * c;
            const float s = sq[i] * rho + omr * gr * gr;
            const float delta = sqrtf(acc[i] + eps) / sqrtf(s + eps) * gr;
            sq[i] = s;
            acc[i] = acc[i] * rho + omr * delta * delta;
            p[i] = p[i] - lr * delta;
        }
    }
}

// torch.optim.Adam (amsgrad = False): m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
// p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v,
                                                   int64_t n, float step_size, float omb1, float b2,
                                                   float omb2, float eps, float sqrt_bc2,
                                                   const float *coef) {
    if (skip_of(coef)) return;
    const float c = clip_of(coef);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float gr = g[i] * c;
        const float mi = m[i] + (gr - m[i]) * omb1;                  // lerp, as torch
        const float vi = v[i] * b2 + omb2 * gr * gr;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - step_size * (mi / (sqrtf(vi) / sqrt_bc2 + eps));
    }
}

// ---- multi-tensor form: up to MT_MAX parameter tensors per launch.  At 18-60 tensors per model the
// per-tensor launches were host-bound (one ctypes call + launch every ~20 us with the GPU idle in
// between: 0.36 ms per cfg2 step); the table travels by value in the kernel arguments.
constexpr int MT_MAX = 24;
struct MultiArgs {
    float *p[MT_MAX];
    const float *g[MT_MAX];
    float *s0[MT_MAX];
    float *s1[MT_MAX];
    long long n[MT_MAX];
    int blk0[MT_MAX + 1];   // first block of tensor t; blk0[count] = grid size
    int count;
    float h[8];             // hyper-parameters (meaning depends on OP)
    const float *coef;
};

template <int OP>  // 0 adadelta (h: lr, rho, 1-rho, eps)   1 adam (h: lr/bc1, 1-b1, b2, 1-b2, eps, sqrt(bc2))
__global__ __launch_bounds__(256) void multi_step_kernel(MultiArgs a) {
    if (skip_of(a.coef)) return;
    int t = 0;
    while (t + 1 < a.count && (int)blockIdx.x >= a.blk0[t + 1]) ++t;
    const int nb = a.blk0[t + 1] - a.blk0[t], b = blockIdx.x - a.blk0[t];
    float *__restrict__ p = a.p[t];
    const float *__restrict__ g = a.g[t];
    float *__restrict__ s0 = a.s0[t];
    float *__restrict__ s1 = a.s1[t];
    const long long n = a.n[t];
    const float c = clip_of(a.coef);
    const long long stride = (long long)nb * 256;
    // element-wise and order-identical to the single-tensor kernels above
    for (long long i = (long long)b * 256 + threadIdx.x; i < n; i += stride) {
        const float gr = g[i] * c;
        if (OP == 0) {
            const float s = s0[i] * a.h[1] + a.h[2] * gr * gr;
            const float delta = sqrtf(s1[i] + a.h[3]) / sqrtf(s + a.h[3]) * gr;
            s0[i] = s;
            s1[i] = s1[i] * a.h[1] + a.h[2] * delta * delta;
            p[i] = p[i] - a.h[0] * delta;
        } else {
            const float mi = s0[i] + (gr - s0[i]) * a.h[1];
            const float vi = s1[i] * a.h[2] + a.h[3] * gr * gr;
            s0[i] = mi;
            s1[i] = vi;
            p[i] = p[i] - a.h[0] * (mi / (sqrtf(vi) / a.h[5] + a.h[4]));
        }
    }
}

template <int OP>
int multi_launch(int count, float *const *params, const float *const *grads, float *const *st0,
                 float *const *st1, const int64_t *numel, const float (&h)[8], const float *coef,
                 hipStream_t s) {
    // `t` is the consumed-tensor cursor: empty tensors are skipped without taking a table slot, so a
    // chunk may span more than MT_MAX entries and the next chunk starts exactly where this one ended
    double bytes = 0.0;
    for (int t = 0; t < count; ++t) bytes += numel[t] > 0 ? 28.0 * (double)numel[t] : 0.0;   // read p, g, 2 states; write p, 2 states
    asrk_prof_work_(PROF_OPTIM, bytes);
    asrk_prof_begin_(PROF_OPTIM, s);
    for (int t = 0; t < count;) {
        MultiArgs a;
        a.count = 0;
        a.coef = coef;
        for (int j = 0; j < 8; ++j) a.h[j] = h[j];
        int blocks = 0;
        for (; t < count && a.count < MT_MAX; ++t) {
            if (numel[t] < 0) return ASRK_EINVAL;
            if (numel[t] == 0) continue;
            if (!params[t] || !grads[t] || !st0[t] || !st1[t]) return ASRK_EINVAL;
            const int k = a.count++;
            a.p[k] = params[t]; a.g[k] = grads[t]; a.s0[k] = st0[t]; a.s1[k] = st1[t];
            a.n[k] = numel[t];
            a.blk0[k] = blocks;
            // ~4 elements per thread, at most 2048 blocks per tensor
            blocks += (int)std::max<int64_t>(1, std::min<int64_t>(asrk_div_up64(numel[t], 1024), 2048));
        }
        if (a.count == 0) continue;
        a.blk0[a.count] = blocks;
        hipLaunchKernelGGL((multi_step_kernel<OP>), dim3(blocks), dim3(256), 0, s, a);
        ASRK_LAUNCH_CHECK();
    }
    asrk_prof_end_(PROF_OPTIM, s);
    return ASRK_OK;
}

// ---- global gradient norm (clip_grad_norm_, src/solver.py:84): sum of squares of a list of tensors in two
// deterministic stages - per-block partial sums (float per thread, double across the block), then ONE block adds
// the partials in a fixed order and writes norm = sqrt(sum) and the clipping coefficient max_norm / (norm + 1e-6).
struct NormArgs {
    const float *g[MT_MAX];
    long long n[MT_MAX];
    int blk0[MT_MAX + 1];
    int count;
    double *partials;          // [grid] of this launch, already offset
};

__global__ __launch_bounds__(256) void sqnorm_partial_kernel(NormArgs a) {
    __shared__ double red[4];
    int t = 0;
    while (t + 1 < a.count && (int)blockIdx.x >= a.blk0[t + 1]) ++t;
    const int nb = a.blk0[t + 1] - a.blk0[t], b = blockIdx.x - a.blk0[t];
    const float *__restrict__ g = a.g[t];
    const long long n = a.n[t], stride = (long long)nb * 256;
    float acc = 0.f;
    for (long long i = (long long)b * 256 + threadIdx.x; i < n; i += stride) acc += g[i] * g[i];
    double d = (double)acc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) a.partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void sqnorm_final_kernel(const double *__restrict__ partials, int n, float max_norm,
                                                           float *__restrict__ norm_out, float *__restrict__ coef_out) {
    __shared__ double red[256];
    double d = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) d += partials[i];       // fixed order per thread
    red[threadIdx.x] = d;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {                                 // fixed tree
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(red[0]);
        if (norm_out) norm_out[0] = norm;
        if (coef_out) coef_out[0] = max_norm / (norm + 1e-6f);
    }
}

__global__ void fill_kernel(float *__restrict__ p, long long n, float v) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

inline int norm_blocks_of(int64_t n) {
    return (int)std::max<int64_t>(1, std::min<int64_t>(asrk_div_up64(n, 2048), 1024));
}

inline unsigned grid_for(int64_t n) {
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>(asrk_div_up64(n, 256), 4096));
}
inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int asrk_adadelta_step_f32(float *param, const float *grad, float *square_avg,
                                      float *acc_delta, int64_t n, double lr, double rho, double eps,
                                      const float *clip_coef, void *stream) {
    if (n < 0) return ASRK_EINVAL;
    if (n == 0) return ASRK_OK;
    if (!param || !grad || !square_avg || !acc_delta) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = al16(param) && al16(grad) && al16(square_avg) && al16(acc_delta) && (n % 4 == 0);
    if (vec)
        hipLaunchKernelGGL((adadelta_kernel<true>), dim3(grid_for(n / 4)), dim3(256), 0, s, param, grad,
                           square_avg, acc_delta, n, (float)lr, (float)rho, (float)(1.0 - rho), (float)eps, clip_coef);
    else
        hipLaunchKernelGGL((adadelta_kernel<false>), dim3(grid_for(n)), dim3(256), 0, s, param, grad,
                           square_avg, acc_delta, n, (float)lr, (float)rho, (float)(1.0 - rho), (float)eps, clip_coef);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq,
                                  int64_t n, double lr, double beta1, double beta2, double eps,
                                  int64_t step, const float *clip_coef, void *stream) {
    if (n < 0 || step < 1) return ASRK_EINVAL;
    if (n == 0) return ASRK_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq,
                       n, (float)(lr / bc1), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                       (float)sqrt(bc2), clip_coef);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_adadelta_multi_f32(int count, float *const *params, const float *const *grads,
                                       float *const *square_avg, float *const *acc_delta,
                                       const int64_t *numel, double lr, double rho, double eps,
                                       const float *clip_coef, void *stream) {
    if (count < 0) return ASRK_EINVAL;
    if (count == 0) return ASRK_OK;
    if (!params || !grads || !square_avg || !acc_delta || !numel) return ASRK_EINVAL;
    const float h[8] = {(float)lr, (float)rho, (float)(1.0 - rho), (float)eps, 0.f, 0.f, 0.f, 0.f};
    return multi_launch<0>(count, params, grads, square_avg, acc_delta, numel, h, clip_coef,
                           (hipStream_t)stream);
}

extern "C" int asrk_adam_multi_f32(int count, float *const *params, const float *const *grads,
                                   float *const *exp_avg, float *const *exp_avg_sq,
                                   const int64_t *numel, double lr, double beta1, double beta2,
                                   double eps, int64_t step, const float *clip_coef, void *stream) {
    if (count < 0 || step < 1) return ASRK_EINVAL;
    if (count == 0) return ASRK_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel) return ASRK_EINVAL;
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const float h[8] = {(float)(lr / bc1), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                        (float)eps, (float)sqrt(bc2), 0.f, 0.f};
    return multi_launch<1>(count, params, grads, exp_avg, exp_avg_sq, numel, h, clip_coef,
                           (hipStream_t)stream);
}

extern "C" size_t asrk_grad_norm_ws_bytes(int count, const int64_t *numel) {
    if (count < 0 || (count > 0 && !numel)) return 0;
    size_t blocks = 0;
    for (int t = 0; t < count; ++t)
        if (numel[t] > 0) blocks += (size_t)norm_blocks_of(numel[t]);
    return (blocks + 1) * sizeof(double);
}

extern "C" int asrk_grad_norm_multi_f32(int count, const float *const *grads, const int64_t *numel, float max_norm,
                                        float *norm_out, float *coef_out, void *ws, size_t ws_bytes, void *stream) {
    if (count < 0 || (count > 0 && (!grads || !numel)) || (!norm_out && !coef_out)) return ASRK_EINVAL;
    if (!ws || ws_bytes < asrk_grad_norm_ws_bytes(count, numel) || (reinterpret_cast<uintptr_t>(ws) & 7) != 0)
        return ASRK_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    double *partials = reinterpret_cast<double *>(ws);
    int total = 0;
    double bytes = 0.0;
    for (int t = 0; t < count; ++t) bytes += numel[t] > 0 ? 4.0 * (double)numel[t] : 0.0;     // one read of every gradient
    asrk_prof_work_(PROF_OPTIM, bytes);
    asrk_prof_begin_(PROF_OPTIM, s);
    for (int t = 0; t < count;) {
        NormArgs a;
        a.count = 0;
        int blocks = 0;
        for (; t < count && a.count < MT_MAX; ++t) {
            if (numel[t] < 0) return ASRK_EINVAL;
            if (numel[t] == 0) continue;
            if (!grads[t]) return ASRK_EINVAL;
            const int k = a.count++;
            a.g[k] = grads[t]; a.n[k] = numel[t]; a.blk0[k] = blocks;
            blocks += norm_blocks_of(numel[t]);
        }
        if (a.count == 0) continue;
        a.blk0[a.count] = blocks;
        a.partials = partials + total;
        hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(blocks), dim3(256), 0, s, a);
        ASRK_LAUNCH_CHECK();
        total += blocks;
    }
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, s, partials, total, max_norm, norm_out, coef_out);
    asrk_prof_end_(PROF_OPTIM, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_fill_f32(float *x, int64_t n, float value, void *stream) {
    if (n < 0) return ASRK_EINVAL;
    if (n == 0) return ASRK_OK;
    if (!x) return ASRK_EINVAL;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, (long long)n, value);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}
