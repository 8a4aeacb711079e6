// Effective shader clock under load: clock64() (shader cycles) vs wall_clock64() (100 MHz) inside
// MFMA-only and MFMA+LDS-read kernels on every CU.
// build: hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int LDS_READS>  // ds_read_b128 per 16 MFMAs
__global__ __launch_bounds__(256) void k(float *out, long long *clk, int iters) {
    __shared__ __attribute__((aligned(16))) float sm[128 * 36 * 2];
    for (int i = threadIdx.x; i < 128 * 36 * 2; i += 256) sm[i] = (float)i * 1e-6f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        f32x4 f[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < LDS_READS)
                f[q] = *reinterpret_cast<const f32x4 *>(sm + ((lane & 31) + 32 * (q & 1)) * 36 + ((it + q) & 7) * 4);
            else
                f[q] = f32x4{1.f + it, 2.f, 3.f, 4.f};
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f[i][s], f[2 + j][s], acc[i * 2 + j], 0, 0, 0);
    }
    long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int L>
void run(int wg_per_cu, int ncu, float *d, long long *clk) {
    const int iters = 40000, grid = ncu * wg_per_cu;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<L>, dim3(grid), dim3(256), 0, 0, d, clk, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<L>, dim3(grid), dim3(256), 0, 0, d, clk, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const double flops = (double)grid * 4 * iters * 16 * 4096.0;
    printf("LDS reads/16 MFMA = %d, %d WG/CU: %.1f TFLOP/s  shader clock %.0f MHz  (%.0f cycles per 16 MFMA, ideal 1024)\n",
           L, wg_per_cu, flops / ms * 1e-9, (double)h[0] / ((double)h[1] / 100.0), (double)h[0] / iters);
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    float *d; long long *clk; (void)hipMalloc(&d, 1 << 24); (void)hipMalloc(&clk, 1 << 16);
    for (int rep = 0; rep < 2; ++rep)
        for (int w = 1; w <= 2; ++w) { run<0>(w, p.multiProcessorCount, d, clk); run<4>(w, p.multiProcessorCount, d, clk); }
    return 0;
}
