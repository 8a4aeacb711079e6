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

template <bool VEC>
__global__ __launch_bounds__(256) void adadelta_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                       float *__restrict__ sq, float *__restrict__ acc,
                                                       int64_t n, float lr, float rho, float omr,
                                                       float eps, const float *coef) {
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
