// Debug microbenchmark (GPU): the inner loop of the backward LSTM recurrence in isolation --
// per k-group 4 x v_mfma_f32_16x16x4_f32 + one ds_read_b128 (A operand) + one 1-KiB buffer load
// (B operand ring refill) -- to separate matrix-pipe issue limits from memory-path limits.
//   mode 0: no refill loads        mode 1: refill from a private 128-KiB window per workgroup (L2 hits)
//   mode 2: all workgroups share one 128-KiB window      mode 3: refill loads are out of bounds (no traffic)
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_ring.hip -o tools/mfma_ring
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int ACC>
__global__ __launch_bounds__(256) void ring_kernel(float *buf, float *out, long long *cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 16 * 2052; i += 256) smem[i] = 1e-3f * (i & 15);
    __syncthreads();
    const int m16 = lane & 15, q4 = lane >> 4;
    const float *wrow = smem + m16 * 2052 + wave * 512;
    const size_t win = MODE == 2 ? 0 : (size_t)blockIdx.x * 32768;  // floats
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(buf + win), 0, 128 * 1024, 0x00020000);
    const unsigned voff = MODE == 3 ? 0x7ff00000u : (unsigned)((m16 * 16 + 4 * q4) * 4);
    f32x4 bf[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (unsigned)((wave * 32 + r) * 1024), 0);
        bf[r] = __builtin_bit_cast(f32x4, x);
    }
    f32x4 acc[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
    const long long t0 = __builtin_readcyclecounter();
    f32x4 a = *reinterpret_cast<const f32x4 *>(wrow + 4 * q4);
    for (int it = 0; it < iters; ++it) {
        for (int kg0 = 0; kg0 < 32; kg0 += 16) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const f32x4 an = *reinterpret_cast<const f32x4 *>(wrow + ((kg0 + r + 1) & 31) * 16 + 4 * q4);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j % ACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bf[r][j], acc[j % ACC], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (MODE != 0) {
                    const int kg = (kg0 + 16 + r) & 31;
                    u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (unsigned)((wave * 32 + kg) * 1024), 0);
                    bf[r] = __builtin_bit_cast(f32x4, x);
                }
                __builtin_amdgcn_sched_barrier(0);
                a = an;
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < ACC; ++i) s += acc[i];
    out[(size_t)blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int ACC>
void run(const char *tag, int grid, float *buf, float *out, long long *cyc, int iters) {
    auto k = ring_kernel<MODE, ACC>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<grid, 256, 140 * 1024>>>(buf, out, cyc, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<grid, 256, 140 * 1024>>>(buf, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (long long v : h) mean += (double)v;
    mean /= grid;
    const double groups = (double)iters * 32;
    printf("%-34s grid %3d acc %d: %7.1f cycles/k-group (ideal 128), %6.1f ns/k-group, counter %.0f MHz\n", tag, grid,
           ACC, mean / groups, ms * 1e6 / groups, mean / (ms * 1e3));
}

int main() {
    float *buf, *out;
    long long *cyc;
    hipMalloc(&buf, (size_t)256 * 32768 * 4 + (1 << 20));
    hipMemset(buf, 0, (size_t)256 * 32768 * 4 + (1 << 20));
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    for (int grid : {128, 256}) {
        run<0, 2>("no refill loads", grid, buf, out, cyc, iters);
        run<0, 4>("no refill loads", grid, buf, out, cyc, iters);
        run<3, 4>("out-of-bounds refill loads", grid, buf, out, cyc, iters);
        run<1, 4>("private 128 KiB window (L2)", grid, buf, out, cyc, iters);
        run<2, 4>("shared 128 KiB window (L2)", grid, buf, out, cyc, iters);
        run<1, 2>("private 128 KiB window (L2)", grid, buf, out, cyc, iters);
    }
    return 0;
}
