// Sustained f32 MFMA issue rate on gfx950 (no memory traffic): every wave runs `iters` rounds of
// 4 independent v_mfma_f32_32x32x2_f32 (or 16x16x4).  Prints TFLOP/s for 1 and 2 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    float a = seed + threadIdx.x, b = seed - threadIdx.x;
    if (SHAPE == 32) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

template <int SHAPE>
void run(int wg_per_cu, int ncu, float *d) {
    const int iters = 20000;
    const int grid = ncu * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<SHAPE>, dim3(grid), dim3(256), 0, 0, d, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<SHAPE>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop_per_mfma = SHAPE == 32 ? 32.0 * 32 * 2 * 2 : 16.0 * 16 * 4 * 2;
    const double flops = (double)grid * 4 * iters * 16 * flop_per_mfma;
    printf("mfma_f32_%s  %d WG/CU (%d waves/SIMD): %.1f TFLOP/s  (%.2f ms)\n",
           SHAPE == 32 ? "32x32x2" : "16x16x4", wg_per_cu, wg_per_cu, flops / ms * 1e-9, ms);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, clock %d MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    float *d; hipMalloc(&d, 256 * 1024 * 16);
    for (int w = 1; w <= 2; ++w) { run<32>(w, p.multiProcessorCount, d); run<16>(w, p.multiProcessorCount, d); }
    run<32>(1, p.multiProcessorCount, d);
    return 0;
}
