// Micro-benchmark (debug tool, not part of the library): one-word ping-pong latency between two
// workgroups on gfx950 for several store/load flavours, same-XCD vs cross-XCD.
//   hipcc --offload-arch=gfx950 -O3 -o pingpong tools/pingpong.hip && ./pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define SPIN_LIMIT 400000
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define RLX_SYS __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM
#define RLX_WG __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

template <int ST, int LD>
__device__ __forceinline__ void do_store(unsigned *p, unsigned v) {
    if (ST == 0) __hip_atomic_store(p, v, RLX_AGENT);           // sc1 (write-through)
    else if (ST == 1) *(volatile unsigned *)p = v;              // plain
    else if (ST == 2) __hip_atomic_store(p, v, RLX_SYS);        // sc0 sc1
    else if (ST == 3) __hip_atomic_exchange(p, v, RLX_AGENT);   // L2 atomic RMW
    else if (ST == 4) __hip_atomic_store(p, v, RLX_WG);         // sc0
}
template <int ST, int LD>
__device__ __forceinline__ unsigned do_load(unsigned *p) {
    if (LD == 0) return __hip_atomic_load(p, RLX_AGENT);        // sc1
    else if (LD == 1) return __builtin_nontemporal_load(p);     // nt
    else if (LD == 2) return __hip_atomic_load(p, RLX_SYS);     // sc0 sc1
    else if (LD == 3) return __hip_atomic_fetch_add(p, 0u, RLX_AGENT);  // atomic RMW read
    else if (LD == 4) return __hip_atomic_load(p, RLX_WG);      // sc0
    else if (LD == 5) return __hip_atomic_fetch_add(p, 0u, RLX_WG);   // RMW executed in the XCD's L2, no sc1
    else {                                                      // sc0 sc1 nt buffer load (streaming: L1 / L2 miss-evict?)
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, 4, 0x00020000);
        return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 1);     // aux bit 0 = sc0 (glc-like)
    }
}

// blocks a and b ping-pong on two words (each written by one side only)
template <int ST, int LD>
__global__ void pingpong(unsigned *words, int a, int b, int iters, unsigned long long *out,
                         unsigned *xcc) {
    const int bid = blockIdx.x;
    if (threadIdx.x == 0) xcc[bid] = xcc_id();
    if (bid != a && bid != b) return;
    if (threadIdx.x != 0) return;
    unsigned *mine = words + (bid == a ? 0 : 64);
    unsigned *theirs = words + (bid == a ? 64 : 0);
    unsigned long long t0 = 0;
    for (int i = 1; i <= iters; ++i) {
        if (i == 11) t0 = wall_clock64();
        if (bid == a) {
            do_store<ST, LD>(mine, (unsigned)i);
            long spins = 0;
            while (do_load<ST, LD>(theirs) < (unsigned)i) {
                if (++spins > SPIN_LIMIT) { out[2] = 1; return; }
            }
        } else {
            long spins = 0;
            while (do_load<ST, LD>(theirs) < (unsigned)i) {
                if (++spins > SPIN_LIMIT) { out[2] = 1; return; }
            }
            do_store<ST, LD>(mine, (unsigned)i);
        }
    }
    if (bid == a) {
        out[0] = wall_clock64() - t0;
        out[1] = iters - 10;
    }
}

// fan: N producers each store their word; one consumer wave polls all N words (allgather-like),
// then the consumer signals next round via a broadcast word that all producers poll.
template <int ST, int LD>
__global__ void fan(unsigned *words, int nprod, int cons, int iters, unsigned long long *out) {
    const int bid = blockIdx.x;
    if (bid > nprod) return;
    unsigned *bc = words;  // broadcast word (written by the consumer)
    unsigned *slots = words + 64;
    if (bid == cons) {
        if (threadIdx.x >= 64) return;
        unsigned long long t0 = 0;
        for (int i = 1; i <= iters; ++i) {
            if (i == 11) t0 = wall_clock64();
            if (threadIdx.x == 0) do_store<ST, LD>(bc, (unsigned)i);
            long spins = 0;
            for (;;) {
                bool ok = true;
                for (int j = threadIdx.x; j < nprod; j += 64)
                    ok &= do_load<ST, LD>(slots + j * 32) >= (unsigned)i;
                if (__all(ok)) break;
                if (++spins > SPIN_LIMIT) { out[2] = 1; return; }
            }
        }
        if (threadIdx.x == 0) { out[0] = wall_clock64() - t0; out[1] = iters - 10; }
    } else {
        if (threadIdx.x != 0) return;
        const int j = bid < cons ? bid : bid - 1;
        for (int i = 1; i <= iters; ++i) {
            long spins = 0;
            while (do_load<ST, LD>(bc) < (unsigned)i) {
                if (++spins > SPIN_LIMIT) { out[2] = 1; return; }
            }
            do_store<ST, LD>(slots + j * 32, (unsigned)i);
        }
    }
}

template <int ST, int LD>
void run(const char *name, unsigned *words, unsigned long long *out, unsigned *xcc, int nblk) {
    const int iters = 2010;
    int pairs[3][2] = {{0, 8}, {0, 1}, {0, 4}};
    printf("%-34s", name);
    for (auto &pr : pairs) {
        hipMemset(words, 0, 4096 * 4);
        hipMemset(out, 0, 64);
        hipLaunchKernelGGL((pingpong<ST, LD>), dim3(nblk), dim3(64), 0, 0, words, pr[0], pr[1], iters,
                           out, xcc);
        hipDeviceSynchronize();
        unsigned long long h[3];
        std::vector<unsigned> hx(nblk);
        hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
        hipMemcpy(hx.data(), xcc, nblk * 4, hipMemcpyDeviceToHost);
        if (h[2]) printf("  (%d:x%u,%d:x%u) TIMEOUT   ", pr[0], hx[pr[0]], pr[1], hx[pr[1]]);
        else
            printf("  (%d:x%u,%d:x%u) %6.0f ns/rt", pr[0], hx[pr[0]], pr[1], hx[pr[1]],
                   (double)h[0] * 10.0 / (double)h[1]);
    }
    // fan-in/out with 31 producers on blocks 1..31 (spread over XCDs) and with same-XCD producers
    for (int np : {31, 127}) {
        hipMemset(words, 0, 8192 * 4);
        hipMemset(out, 0, 64);
        hipLaunchKernelGGL((fan<ST, LD>), dim3(np + 1), dim3(64), 0, 0, words, np, 0, iters, out);
        hipDeviceSynchronize();
        unsigned long long h[3];
        hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
        if (h[2]) printf("  fan%d TIMEOUT", np);
        else printf("  fan%d %6.0f ns/rt", np, (double)h[0] * 10.0 / (double)h[1]);
    }
    printf("\n");
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const bool only_new = argc > 1;
    unsigned *words, *xcc;
    unsigned long long *out;
    const int nblk = 16;
    hipMalloc(&words, 8192 * 4);
    hipMalloc(&out, 64);
    hipMalloc(&xcc, 1024 * 4);
    printf("round-trip = A stores, B sees it, B stores, A sees it (2 one-way hops); wall clock 100 MHz\n");
    run<0, 0>("st sc1      / ld sc1", words, out, xcc, nblk);
    if (!only_new) {
    run<1, 0>("st plain    / ld sc1", words, out, xcc, nblk);
    run<1, 1>("st plain    / ld nt", words, out, xcc, nblk);
    run<0, 1>("st sc1      / ld nt", words, out, xcc, nblk);
    run<2, 2>("st sc0sc1   / ld sc0sc1", words, out, xcc, nblk);
    run<3, 0>("st atomicxchg / ld sc1", words, out, xcc, nblk);
    run<3, 3>("st atomicxchg / ld atomic(add 0)", words, out, xcc, nblk);
    run<1, 3>("st plain    / ld atomic(add 0)", words, out, xcc, nblk);
    run<4, 4>("st sc0      / ld sc0", words, out, xcc, nblk);
    }
    // round 5: is there a poll that bypasses the CU's L1 but is served by the XCD's L2 (same-XCD hand-off at L2 speed)?
    run<1, 5>("st plain    / ld atomic-add-0 WG scope", words, out, xcc, nblk);
    run<0, 5>("st sc1      / ld atomic-add-0 WG scope", words, out, xcc, nblk);
    run<4, 5>("st sc0      / ld atomic-add-0 WG scope", words, out, xcc, nblk);
    run<0, 6>("st sc1      / buffer_load sc0", words, out, xcc, nblk);
    return 0;
}
