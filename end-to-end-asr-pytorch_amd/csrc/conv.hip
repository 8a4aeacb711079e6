// Convolutional prenets of the encoder (src/module.py:7-90): VGGExtractor (4 x Conv2d 3x3 + ReLU,
// 2 x MaxPool 2x2) and CNNExtractor (2 x Conv1d k=4 stride 2).
//
// Layout is channels-LAST in HBM ([B, H=time, W=freq, C]); a convolution is
//   im2col (this file, HBM-bound gather)  ->  one MFMA GEMM  [B*Ho*Wo, Cin*KH*KW] x W^T (+bias)
// whose [M, Cout] result IS the channels-last activation of the next layer — no NCHW<->NHWC
// shuffles between layers.  The K axis is ordered (cin, kh, kw), i.e. exactly
// weight.view(Cout, Cin*KH*KW) of the reference's nn.Conv2d/Conv1d parameter, so checkpoints load
// unchanged and dW comes out of the TN GEMM in parameter layout.  Input/grad-input take explicit
// element strides so the first layer reads the [B,T,C*F] feature tensor in place (view_input,
// src/module.py:44-57) and the last pool writes the [B,T/4,C*F/4] layout the RNN expects
// (src/module.py:62-65) without a separate transpose pass.
#include "common.h"
#include <algorithm>

namespace {

struct ConvGeom {
    int B, H, W, C;          // input extent
    int KH, KW, SH, SW, PH, PW;
    int Ho, Wo;
    int64_t sb, sh, sw, sc;  // input element strides
};

// col[m, k] = x[b, ho*SH+kh-PH, wo*SW+kw-PW, cin]  (0 outside), m = (b*Ho+ho)*Wo+wo,
// k = (cin*KH+kh)*KW+kw.  One thread per element, k fastest -> coalesced 4-B stores.  Rows are `ld` >= K floats apart
// and columns K .. ld-1 are written as zeros (a K that is not a multiple of 4 padded up for the 16-byte GEMM paths).
__global__ __launch_bounds__(256) void im2col_kernel(const float *__restrict__ x,
                                                     float *__restrict__ col, ConvGeom g, int ld,
                                                     int64_t total) {
    const int KK = g.KH * g.KW;
    const int K = g.C * KK;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / ld;
        const int k = (int)(i - m * ld);
        if (k >= K) {
            col[i] = 0.f;
            continue;
        }
        const int cin = k / KK;
        const int r = k - cin * KK;
        const int kh = r / g.KW, kw = r - kh * g.KW;
        const int wo = (int)(m % g.Wo);
        const int64_t t = m / g.Wo;
        const int ho = (int)(t % g.Ho);
        const int b = (int)(t / g.Ho);
        const int h = ho * g.SH + kh - g.PH;
        const int w = wo * g.SW + kw - g.PW;
        float v = 0.f;
        if (h >= 0 && h < g.H && w >= 0 && w < g.W)
            v = x[b * g.sb + h * g.sh + w * g.sw + cin * g.sc];
        col[i] = v;
    }
}

// dx[b,h,w,c] = sum over (kh,kw) with (h+PH-kh) % SH == 0, (w+PW-kw) % SW == 0 of
//               dcol[(b,ho,wo), (c*KH+kh)*KW+kw]          (gather form: no atomics)
__global__ __launch_bounds__(256) void col2im_kernel(const float *__restrict__ dcol,
                                                     float *__restrict__ dx, ConvGeom g,
                                                     int64_t total) {
    const int KK = g.KH * g.KW;
    const int K = g.C * KK;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % g.C);
        int64_t t = i / g.C;
        const int w = (int)(t % g.W);
        t /= g.W;
        const int h = (int)(t % g.H);
        const int b = (int)(t / g.H);
        float s = 0.f;
        for (int kh = 0; kh < g.KH; ++kh) {
            const int hn = h + g.PH - kh;
            if (hn < 0 || hn % g.SH != 0) continue;
            const int ho = hn / g.SH;
            if (ho >= g.Ho) continue;
            for (int kw = 0; kw < g.KW; ++kw) {
                const int wn = w + g.PW - kw;
                if (wn < 0 || wn % g.SW != 0) continue;
                const int wo = wn / g.SW;
                if (wo >= g.Wo) continue;
                const int64_t m = ((int64_t)b * g.Ho + ho) * g.Wo + wo;
                s += dcol[m * K + (c * g.KH + kh) * g.KW + kw];
            }
        }
        dx[b * g.sb + h * g.sh + w * g.sw + c * g.sc] = s;
    }
}

__global__ void relu_fwd_kernel(float *__restrict__ x, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        x[i] = fmaxf(x[i], 0.f);
}

// dx = y > 0 ? dy : 0  (y = relu output)
__global__ void relu_bwd_kernel(const float *__restrict__ y, const float *__restrict__ dy,
                                float *__restrict__ dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// 2x2 / stride-2 max pool (floor mode) on contiguous channels-last x [B,H,W,C]; the pooled value
// goes to y[b*osb + ho*osh + wo*osw + c*osc]; idx (contiguous [B,Ho,Wo,C]) keeps the winner 0..3
// (first maximum in (dh,dw) row-major order, as ATen's max_pool2d backward picks).
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float *__restrict__ x,
                                                          float *__restrict__ y,
                                                          uint8_t *__restrict__ idx, int B, int H,
                                                          int W, int C, int Ho, int Wo, int64_t osb,
                                                          int64_t osh, int64_t osw, int64_t osc,
                                                          int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const float *p = x + (((int64_t)b * H + 2 * ho) * W + 2 * wo) * C + c;
        float best = p[0];
        int bi = 0;
        const float v1 = p[C], v2 = p[(int64_t)W * C], v3 = p[(int64_t)W * C + C];
        if (v1 > best || v1 != v1) { best = v1; bi = 1; }
        if (v2 > best || v2 != v2) { best = v2; bi = 2; }
        if (v3 > best || v3 != v3) { best = v3; bi = 3; }
        y[b * osb + ho * osh + wo * osw + c * osc] = best;
        idx[i] = (uint8_t)bi;
    }
}

// dx (contiguous [B,H,W,C], every element written: rows/cols beyond 2*Ho / 2*Wo get 0)
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float *__restrict__ dy,
                                                          const uint8_t *__restrict__ idx,
                                                          float *__restrict__ dx, int B, int H, int W,
                                                          int C, int Ho, int Wo, int64_t osb,
                                                          int64_t osh, int64_t osw, int64_t osc,
                                                          int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t t = i / C;
        const int w = (int)(t % W);
        t /= W;
        const int h = (int)(t % H);
        const int b = (int)(t / H);
        const int ho = h >> 1, wo = w >> 1;
        float v = 0.f;
        if (ho < Ho && wo < Wo) {
            const int me = ((h & 1) << 1) | (w & 1);
            const int64_t o = (((int64_t)b * Ho + ho) * Wo + wo) * C + c;
            if (idx[o] == me) v = dy[b * osb + ho * osh + wo * osw + c * osc];
        }
        dx[i] = v;
    }
}


// the same in 16-byte pieces along the channels (C % 4 == 0, dx 16-byte and idx 4-byte aligned): 131 MB of dx written at
// 2 TB/s by the scalar kernel above became the second-largest non-MFMA item of the VGG prenet's backward
template <bool DYV>
__global__ __launch_bounds__(256) void maxpool_bwd_vec_kernel(const float *__restrict__ dy, const uint8_t *__restrict__ idx,
                                                              float *__restrict__ dx, int B, int H, int W, int C, int Ho,
                                                              int Wo, int64_t osb, int64_t osh, int64_t osw, int64_t osc,
                                                              unsigned total /* B*H*W*C/4 */) {
    const unsigned C4 = (unsigned)C >> 2;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned pos = i / C4, c4 = i - pos * C4;
        const unsigned t = pos / (unsigned)W, w = pos - t * (unsigned)W;
        const unsigned b = t / (unsigned)H, h = t - b * (unsigned)H;
        const int ho = (int)(h >> 1), wo = (int)(w >> 1);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ho < Ho && wo < Wo) {
            const unsigned me = ((h & 1) << 1) | (w & 1);
            const int64_t o = (((int64_t)b * Ho + ho) * Wo + wo) * C + c4 * 4;
            const unsigned win = *reinterpret_cast<const unsigned *>(idx + o);
            const float *dp = dy + (int64_t)b * osb + ho * osh + wo * osw + (int64_t)(c4 * 4) * osc;
            f32x4 d;
            if (DYV) {
                d = *reinterpret_cast<const f32x4 *>(dp);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = dp[e * osc];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ((win >> (8 * e)) & 0xff) == me ? d[e] : 0.f;
        }
        *reinterpret_cast<f32x4 *>(dx + (int64_t)i * 4) = v;
    }
}


// ---- channels-innermost K order (kh, kw, cin) for inputs whose channels are contiguous (sc == 1, C % 4 == 0): a patch
// row is KH*KW runs of C contiguous floats, so im2col / col2im are plain strided copies in whole 16-byte pieces - every
// load and store instruction covers contiguous 256-byte (C = 64) runs - instead of one 4-byte gather per element through
// three integer divisions (the (cin, kh, kw) kernels above: 1.2 / 1.6 TB/s at the shipped VGG sizes).  The weight is
// re-ordered to match (Cout x K floats, asrk_conv_weight_reorder_f32) and dW is re-ordered back: both tiny.
//   col[m, (kh*KW + kw)*C + c] = x[b, ho*SH+kh-PH, wo*SW+kw-PW, c];   grid.y = kh*KW + kw, threads over (m, c/4)
__global__ __launch_bounds__(256) void im2col_cl_kernel(const float *__restrict__ x, float *__restrict__ col,
                                                        ConvGeom g, unsigned total /* M * C/4 */) {
    const int seg = blockIdx.y, kh = seg / g.KW, kw = seg - kh * g.KW;
    const unsigned C4 = (unsigned)g.C >> 2;
    const int64_t K = (int64_t)g.C * g.KH * g.KW;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned m = i / C4, c4 = i - m * C4;
        const unsigned t = m / (unsigned)g.Wo, wo = m - t * (unsigned)g.Wo;
        const unsigned b = t / (unsigned)g.Ho, ho = t - b * (unsigned)g.Ho;
        const int h = (int)ho * g.SH + kh - g.PH, w = (int)wo * g.SW + kw - g.PW;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (h >= 0 && h < g.H && w >= 0 && w < g.W)
            v = *reinterpret_cast<const f32x4 *>(x + (int64_t)b * g.sb + (int64_t)h * g.sh + (int64_t)w * g.sw + c4 * 4);
        *reinterpret_cast<f32x4 *>(col + (int64_t)m * K + (int64_t)seg * g.C + c4 * 4) = v;
    }
}

// dx[b,h,w,c] = sum over (kh,kw) of dcol[(b,ho,wo), (kh*KW+kw)*C + c]   (gather form, 16-byte pieces along c)
__global__ __launch_bounds__(256) void col2im_cl_kernel(const float *__restrict__ dcol, float *__restrict__ dx,
                                                        ConvGeom g, unsigned total /* B*H*W * C/4 */) {
    const unsigned C4 = (unsigned)g.C >> 2;
    const int64_t K = (int64_t)g.C * g.KH * g.KW;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned p = i / C4, c4 = i - p * C4;
        const unsigned t = p / (unsigned)g.W, w = p - t * (unsigned)g.W;
        const unsigned b = t / (unsigned)g.H, h = t - b * (unsigned)g.H;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int kh = 0; kh < g.KH; ++kh) {
            const int hn = (int)h + g.PH - kh;
            if (hn < 0 || hn % g.SH != 0) continue;
            const int ho = hn / g.SH;
            if (ho >= g.Ho) continue;
            for (int kw = 0; kw < g.KW; ++kw) {
                const int wn = (int)w + g.PW - kw;
                if (wn < 0 || wn % g.SW != 0) continue;
                const int wo = wn / g.SW;
                if (wo >= g.Wo) continue;
                const int64_t m = ((int64_t)b * g.Ho + ho) * g.Wo + wo;
                s += *reinterpret_cast<const f32x4 *>(dcol + m * K + (int64_t)(kh * g.KW + kw) * g.C + c4 * 4);
            }
        }
        *reinterpret_cast<f32x4 *>(dx + (int64_t)b * g.sb + (int64_t)h * g.sh + (int64_t)w * g.sw + c4 * 4) = s;
    }
}

// weight.view(Cout, Cin, KK) <-> [Cout][KK][Cin]  (inverse != 0: back to the parameter's order)
__global__ __launch_bounds__(256) void conv_weight_reorder_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                                  int Cout, int Cin, int KK, int inverse) {
    const int K = Cin * KK, total = Cout * K;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int co = i / K, r = i - co * K;
        const int ci = r / KK, kk = r - ci * KK;           // r in parameter order (ci, kk)
        const int j = co * K + kk * Cin + ci;              // the same element in (kk, ci) order
        if (inverse) dst[i] = src[j];
        else dst[j] = src[i];
    }
}

inline bool cl_ok(const ConvGeom &g, const void *a, const void *b) {
    return g.sc == 1 && g.C % 4 == 0 && g.sb % 4 == 0 && g.sh % 4 == 0 && g.sw % 4 == 0 &&
           ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
}

inline unsigned grid_for(int64_t total) {
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>(asrk_div_up64(total, 256), 1 << 16));
}

inline bool geom_ok(const ConvGeom &g) {
    return g.B >= 0 && g.H > 0 && g.W > 0 && g.C > 0 && g.KH > 0 && g.KW > 0 && g.SH > 0 &&
           g.SW > 0 && g.PH >= 0 && g.PW >= 0 && g.Ho > 0 && g.Wo > 0 &&
           g.Ho == (g.H + 2 * g.PH - g.KH) / g.SH + 1 && g.Wo == (g.W + 2 * g.PW - g.KW) / g.SW + 1;
}

}  // namespace

extern "C" int asrk_conv_out_size(int in, int k, int stride, int pad) {
    if (in <= 0 || k <= 0 || stride <= 0 || pad < 0 || in + 2 * pad < k) return 0;
    return (in + 2 * pad - k) / stride + 1;
}

extern "C" int asrk_im2col_ld_f32(const float *x, float *col, int ldcol, int B, int H, int W, int C, int KH, int KW,
                                  int SH, int SW, int PH, int PW, int64_t sb, int64_t sh, int64_t sw,
                                  int64_t sc, void *stream) {
    ConvGeom g{B, H, W, C, KH, KW, SH, SW, PH, PW, asrk_conv_out_size(H, KH, SH, PH),
               asrk_conv_out_size(W, KW, SW, PW), sb, sh, sw, sc};
    if (!geom_ok(g) || (int64_t)ldcol < (int64_t)C * KH * KW) return ASRK_EINVAL;
    if (B == 0) return ASRK_OK;
    if (!x || !col) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * g.Ho * g.Wo * ldcol;
    asrk_prof_begin_(PROF_CONV, s);
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, col, g, ldcol, total);
    asrk_prof_end_(PROF_CONV, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_im2col_f32(const float *x, float *col, int B, int H, int W, int C, int KH, int KW,
                               int SH, int SW, int PH, int PW, int64_t sb, int64_t sh, int64_t sw,
                               int64_t sc, void *stream) {
    if ((int64_t)C * KH * KW >= (int64_t)1 << 31) return ASRK_EINVAL;
    return asrk_im2col_ld_f32(x, col, C * KH * KW, B, H, W, C, KH, KW, SH, SW, PH, PW, sb, sh, sw, sc, stream);
}

extern "C" int asrk_col2im_f32(const float *dcol, float *dx, int B, int H, int W, int C, int KH, int KW,
                               int SH, int SW, int PH, int PW, int64_t sb, int64_t sh, int64_t sw,
                               int64_t sc, void *stream) {
    ConvGeom g{B, H, W, C, KH, KW, SH, SW, PH, PW, asrk_conv_out_size(H, KH, SH, PH),
               asrk_conv_out_size(W, KW, SW, PW), sb, sh, sw, sc};
    if (!geom_ok(g)) return ASRK_EINVAL;
    if (B == 0) return ASRK_OK;
    if (!dcol || !dx) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * H * W * C;
    asrk_prof_begin_(PROF_CONV, s);
    hipLaunchKernelGGL(col2im_kernel, dim3(grid_for(total)), dim3(256), 0, s, dcol, dx, g, total);
    asrk_prof_end_(PROF_CONV, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}


// The same pair with the K axis ordered (kh, kw, cin) - for channels-contiguous inputs (sc == 1, C % 4 == 0, strides
// multiples of 4 floats, 16-byte aligned pointers; ASRK_ESHAPE otherwise): multiply against a weight re-ordered with
// asrk_conv_weight_reorder_f32 and re-order dW back.
extern "C" int asrk_im2col_cl_f32(const float *x, float *col, int B, int H, int W, int C, int KH, int KW,
                                  int SH, int SW, int PH, int PW, int64_t sb, int64_t sh, int64_t sw,
                                  int64_t sc, void *stream) {
    ConvGeom g{B, H, W, C, KH, KW, SH, SW, PH, PW, asrk_conv_out_size(H, KH, SH, PH),
               asrk_conv_out_size(W, KW, SW, PW), sb, sh, sw, sc};
    if (!geom_ok(g)) return ASRK_EINVAL;
    if (B == 0) return ASRK_OK;
    if (!x || !col) return ASRK_EINVAL;
    const int64_t total = (int64_t)B * g.Ho * g.Wo * (C / 4);
    if (!cl_ok(g, x, col) || total >= (int64_t)1 << 31 || KH * KW > 65535) return ASRK_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_CONV, s);
    hipLaunchKernelGGL(im2col_cl_kernel, dim3(std::min(grid_for(total), 16384u), KH * KW), dim3(256), 0, s, x, col, g,
                       (unsigned)total);
    asrk_prof_end_(PROF_CONV, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_col2im_cl_f32(const float *dcol, float *dx, int B, int H, int W, int C, int KH, int KW,
                                  int SH, int SW, int PH, int PW, int64_t sb, int64_t sh, int64_t sw,
                                  int64_t sc, void *stream) {
    ConvGeom g{B, H, W, C, KH, KW, SH, SW, PH, PW, asrk_conv_out_size(H, KH, SH, PH),
               asrk_conv_out_size(W, KW, SW, PW), sb, sh, sw, sc};
    if (!geom_ok(g)) return ASRK_EINVAL;
    if (B == 0) return ASRK_OK;
    if (!dcol || !dx) return ASRK_EINVAL;
    const int64_t total = (int64_t)B * H * W * (C / 4);
    if (!cl_ok(g, dcol, dx) || total >= (int64_t)1 << 31) return ASRK_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_CONV, s);
    hipLaunchKernelGGL(col2im_cl_kernel, dim3(grid_for(total)), dim3(256), 0, s, dcol, dx, g, (unsigned)total);
    asrk_prof_end_(PROF_CONV, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_conv_weight_reorder_f32(const float *src, float *dst, int Cout, int Cin, int KK, int inverse,
                                            void *stream) {
    if (Cout <= 0 || Cin <= 0 || KK <= 0 || !src || !dst || src == dst) return ASRK_EINVAL;
    if ((int64_t)Cout * Cin * KK >= (int64_t)1 << 31) return ASRK_ESHAPE;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_CONV, s);
    hipLaunchKernelGGL(conv_weight_reorder_kernel, dim3(grid_for((int64_t)Cout * Cin * KK)), dim3(256), 0, s, src, dst,
                       Cout, Cin, KK, inverse);
    asrk_prof_end_(PROF_CONV, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_relu_fwd_f32(float *x, int64_t n, void *stream) {
    if (n < 0) return ASRK_EINVAL;
    if (n == 0) return ASRK_OK;
    if (!x) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_CONV, s);
    hipLaunchKernelGGL(relu_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, n);
    asrk_prof_end_(PROF_CONV, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_relu_bwd_f32(const float *y, const float *dy, float *dx, int64_t n, void *stream) {
    if (n < 0) return ASRK_EINVAL;
    if (n == 0) return ASRK_OK;
    if (!y || !dy || !dx) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    asrk_prof_begin_(PROF_CONV, s);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, y, dy, dx, n);
    asrk_prof_end_(PROF_CONV, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_maxpool2x2_fwd_f32(const float *x, float *y, uint8_t *idx, int B, int H, int W,
                                       int C, int64_t osb, int64_t osh, int64_t osw, int64_t osc,
                                       void *stream) {
    if (B < 0 || H < 2 || W < 2 || C <= 0) return ASRK_EINVAL;
    if (B == 0) return ASRK_OK;
    if (!x || !y || !idx) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)B * Ho * Wo * C;
    asrk_prof_begin_(PROF_CONV, s);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, y, idx, B, H, W,
                       C, Ho, Wo, osb, osh, osw, osc, total);
    asrk_prof_end_(PROF_CONV, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}

extern "C" int asrk_maxpool2x2_bwd_f32(const float *dy, const uint8_t *idx, float *dx, int B, int H,
                                       int W, int C, int64_t osb, int64_t osh, int64_t osw, int64_t osc,
                                       void *stream) {
    if (B < 0 || H < 2 || W < 2 || C <= 0) return ASRK_EINVAL;
    if (B == 0) return ASRK_OK;
    if (!dy || !idx || !dx) return ASRK_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)B * H * W * C;
    asrk_prof_begin_(PROF_CONV, s);
    auto a16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (C % 4 == 0 && a16(dx) && (reinterpret_cast<uintptr_t>(idx) & 3) == 0 && total / 4 < ((int64_t)1 << 31)) {
        const bool dyv = osc == 1 && a16(dy) && osb % 4 == 0 && osh % 4 == 0 && osw % 4 == 0;
        if (dyv)
            hipLaunchKernelGGL(maxpool_bwd_vec_kernel<true>, dim3(grid_for(total / 4)), dim3(256), 0, s, dy, idx, dx, B, H, W,
                               C, Ho, Wo, osb, osh, osw, osc, (unsigned)(total / 4));
        else
            hipLaunchKernelGGL(maxpool_bwd_vec_kernel<false>, dim3(grid_for(total / 4)), dim3(256), 0, s, dy, idx, dx, B, H, W,
                               C, Ho, Wo, osb, osh, osw, osc, (unsigned)(total / 4));
    } else {
        hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, dy, idx, dx, B, H, W,
                           C, Ho, Wo, osb, osh, osw, osc, total);
    }
    asrk_prof_end_(PROF_CONV, s);
    ASRK_LAUNCH_CHECK();
    return ASRK_OK;
}
