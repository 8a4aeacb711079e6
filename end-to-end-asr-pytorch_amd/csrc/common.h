// Shared device/host helpers for the asrk (ASR kernels) library — gfx950 / CDNA4 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/asrk.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define ASRK_LAUNCH_CHECK()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return (int)e__;               \
    } while (0)

#define ASRK_HIP(call)                                        \
    do {                                                      \
        hipError_t e__ = (call);                              \
        if (e__ != hipSuccess) return (int)e__;               \
    } while (0)

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to ONE device's copy of a kernel: latch "already set" per device
// (bit d of the mask), never process-wide - a second GPU or a second host thread must not launch with the default limit.
struct AsrkLdsLatch { unsigned long long done = 0; };
static inline hipError_t asrk_max_lds_once(AsrkLdsLatch &l, const void *fn, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(&l.done, __ATOMIC_ACQUIRE) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) __atomic_fetch_or(&l.done, bit, __ATOMIC_RELEASE);
    return e;
}

static inline int asrk_div_up(int a, int b) { return (a + b - 1) / b; }
static inline int64_t asrk_div_up64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// wave64 reductions (DPP/shuffle based; all 64 lanes participate)
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// log(exp(a)+exp(b)) that tolerates -inf on either side (CTC lattice)
__device__ __forceinline__ float log_add(float a, float b) {
    float m = fmaxf(a, b);
    if (m == -INFINITY) return -INFINITY;
    return m + log1pf(expf(-fabsf(a - b)));
}

__device__ __forceinline__ float sigmoidf_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

// optional per-kernel timing hooks (profile.cpp)
extern "C" void asrk_prof_begin_(int id, hipStream_t s);
extern "C" void asrk_prof_end_(int id, hipStream_t s);
extern "C" void asrk_prof_work_(int id, double flops);
enum AsrkProfId {
    PROF_GEMM = 0,
    PROF_LSTM_FWD = 1,
    PROF_LSTM_BWD = 2,
    PROF_CTC = 3,
    PROF_ROWOPS = 4,
    PROF_ATTN = 5,
    PROF_CELL = 6,
    PROF_FBANK = 7,
    PROF_GEMM_BG = 8,   // GEMM launches carrying the background launch hint (one workgroup per CU)
    PROF_SPELLER = 9,   // the fused attention-decoder loop (speller.hip)
    PROF_CONV = 10,     // prenet convolutions: im2col / col2im / ReLU / max-pool (conv.hip; their GEMMs count as GEMM)
    PROF_SPLIT = 11,    // split passes (f32 -> three bf16 planes); nested inside the GEMM family's time; work = HBM bytes
    PROF_OPTIM = 12,    // fused optimiser steps + the gradient-norm passes; work = HBM bytes
    PROF_CONV_MFMA = 13,  // implicit-GEMM convolutions (conv3x3.hip: forward / data gradient / weight gradient on the f32 matrix cores)
    PROF_NUM = 14
};
