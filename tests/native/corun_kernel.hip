// TEST INFRASTRUCTURE ONLY (tests/test_parallel_gpu.py): a kernel shaped like an RCCL ring-reduce step - one
// 256-thread workgroup per CU, ~96 VGPRs per lane, no LDS to speak of, streaming a 32-MiB buffer
// (read two sources, add, write) `iters` times - to be launched on a side stream WHILE the persistent bf16x6
// recurrence kernels run, the situation of a gradient all-reduce under data parallelism on a real 8-GPU node
// (which this box does not have).  Not part of libasrk; built by __graft_entry__.build() for gfx950.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void corun_reduce_kernel(float *__restrict__ dst, const float *__restrict__ src,
                                                           size_t n16, int iters) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int it = 0; it < iters; ++it) {
        for (size_t base = (size_t)blockIdx.x * blockDim.x + threadIdx.x; base < n16; base += stride * 12) {
            f32x4 a[12], b[12];                          // 96 VGPRs of payload in flight, like a collective's FIFO slots
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const size_t i = base + (size_t)k * stride;
                if (i < n16) {
                    a[k] = reinterpret_cast<const f32x4 *>(src)[i];
                    b[k] = reinterpret_cast<const f32x4 *>(dst)[i];
                }
            }
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const size_t i = base + (size_t)k * stride;
                if (i < n16) reinterpret_cast<f32x4 *>(dst)[i] = a[k] + b[k] * 0.5f;
            }
        }
    }
}

extern "C" int corun_reduce(float *dst, const float *src, size_t n_floats, int iters, int workgroups, void *stream) {
    hipLaunchKernelGGL(corun_reduce_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, dst, src,
                       n_floats / 4, iters);
    return (int)hipGetLastError();
}
