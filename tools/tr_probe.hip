// What does ds_read_b64_tr_b16 deliver?  LDS holds element i at 16-bit slot i (value = i); every lane reads the
// 8 bytes at byte offset `off(lane)` with the transposing read and with a plain ds_read_b64; prints, per lane, the
// four 16-bit values of both.  Two address patterns: (a) lane-linear (lane * 8), (b) the [32 k][16 m] block image
// of DESIGN.md section 7 (row = k, 32 B per row): lane s of a 16-lane group reads row s >> 2, 8-byte quad s & 3.
// build: hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o tools/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void k(int pattern, unsigned short *out_tr, unsigned short *out_plain) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    int off;
    if (pattern == 0) off = lane * 8;
    else {
        const int g = lane >> 4, s = lane & 15;
        // group g: rows 4 * g .. 4 * g + 3 of a [rows][16 elements = 32 B] image
        off = (4 * g + (s >> 2)) * 32 + (s & 3) * 8;
    }
    auto *p = (__attribute__((address_space(3))) s16x4 *)(reinterpret_cast<char *>(lds) + off);
    s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    s16x4 q = *reinterpret_cast<s16x4 *>(reinterpret_cast<char *>(lds) + off);
    for (int j = 0; j < 4; ++j) {
        out_tr[lane * 4 + j] = (unsigned short)t[j];
        out_plain[lane * 4 + j] = (unsigned short)q[j];
    }
}

int main() {
    unsigned short *a, *b, ha[256], hb[256];
    hipMalloc(&a, 512); hipMalloc(&b, 512);
    for (int pattern = 0; pattern < 2; ++pattern) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, pattern, a, b);
        hipMemcpy(ha, a, 512, hipMemcpyDeviceToHost);
        hipMemcpy(hb, b, 512, hipMemcpyDeviceToHost);
        printf("pattern %d (%s)\n", pattern, pattern ? "[rows][16] image: row = element / 16, column = element % 16" : "lane-linear");
        for (int l = 0; l < 64; ++l)
            printf(" lane %2d  plain %4d %4d %4d %4d   tr %4d %4d %4d %4d\n", l, hb[l * 4], hb[l * 4 + 1], hb[l * 4 + 2],
                   hb[l * 4 + 3], ha[l * 4], ha[l * 4 + 1], ha[l * 4 + 2], ha[l * 4 + 3]);
    }
    return 0;
}
