/*
 * asrk.h — C ABI of libasrk.so: the MI355X (gfx950 / CDNA4) kernels underneath the
 * LAS/CTC training + decode hot path of Alexander-H-Liu/End-to-end-ASR-Pytorch.
 *
 * The reference has NO FFI of its own (pure Python over torch/ATen); each entry
 * point below therefore cites the reference call site (file:line under /root/reference) whose
 * third-party arithmetic it replaces.  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *  - plain C, raw device pointers, explicit sizes/strides in ELEMENTS (floats) unless stated;
 *  - `stream` is a hipStream_t passed as void*; every call only ENQUEUES work (no device sync);
 *  - the library never allocates/frees device memory: outputs + workspaces are caller-owned
 *    (every entry point that needs scratch has a *_bytes query next to it);
 *  - no mutable mode state: arithmetic / launch variants are `flags` ARGUMENTS of the calls they affect;
 *    the ASRK_* tuning environment variables are read ONCE, at the first asrk_init();
 *    the only process-wide state is the cached device query and the optional profiling hooks below;
 *  - return 0 on success, negative ASRK_E* for argument errors, positive = hipError_t;
 *  - no C++ exceptions cross the boundary, no abort().
 */
#ifndef ASRK_H
#define ASRK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASRK_OK 0
#define ASRK_EINVAL (-1)   /* bad argument (null pointer, negative size, unsupported combo) */
#define ASRK_ESHAPE (-2)   /* shape not supported by the persistent kernels (see message)   */
#define ASRK_EWORKSPACE (-3) /* workspace too small                                          */
#define ASRK_EDEVICE (-4)  /* device lacks the CUs/LDS the persistent kernel needs            */
#define ASRK_ETIMEOUT (-5) /* in-kernel grid sync gave up (reported by asrk_lstm_check_error) */

int asrk_version(void);
const char *asrk_strerror(int rc);
/* Query + cache device properties (CU count, LDS/CU). Call once per device before persistent
 * kernels; returns the CU count (>0) or a negative error. */
int asrk_init(int device);

/* ---- optional per-kernel hipEvent timing (bench.py roofline) --------------------------- */
void asrk_profile_enable(int on);
/* Families (bit id = ASRK_PROF_* id) that record events while profiling is on; default all.  An event pair costs ~3 us of
 * stream time (a marker packet waits for the kernel in front of it), so a timed region that only needs the dominant
 * kernel's durations enables that family alone. */
void asrk_profile_families(unsigned mask);
void asrk_profile_reset(void);
/* Resolves pending events (synchronises them) and returns total ms + launch count for a
 * kernel family id (see ASRK_PROF_*). */
int asrk_profile_get(int id, double *total_ms, int64_t *launches);
/* Algorithmic work the family was asked to do while profiling was on: flops for the GEMM families (2*M*N*K per
 * call), HBM bytes for the streaming families (CTC: dense gradient read + write; SPLIT: 4 B read + 6 B written per
 * element; OPTIM: 28 B per parameter for the update + 4 B per gradient element for the norm). */
int asrk_profile_get_work(int id, double *flops);
#define ASRK_PROF_GEMM 0
#define ASRK_PROF_LSTM_FWD 1
#define ASRK_PROF_LSTM_BWD 2
#define ASRK_PROF_CTC 3
#define ASRK_PROF_ROWOPS 4
#define ASRK_PROF_ATTN 5
#define ASRK_PROF_CELL 6
#define ASRK_PROF_FBANK 7
#define ASRK_PROF_GEMM_BG 8 /* asrk_gemm_f32 calls whose flags carry ASRK_GEMM_LDS_HINT(> 80) */
#define ASRK_PROF_SPELLER 9 /* asrk_speller_* (one event pair per call; launches = kernels enqueued) */
#define ASRK_PROF_CONV 10   /* prenet convolutions: im2col / col2im / ReLU / max-pool (their GEMMs count as GEMM) */
#define ASRK_PROF_SPLIT 11  /* split passes of the bf16x6 GEMM path (their time is ALSO inside ASRK_PROF_GEMM) */
#define ASRK_PROF_OPTIM 12  /* fused optimiser steps and gradient-norm passes */

/* ---- dense f32 GEMM on v_mfma_f32_32x32x2_f32 (exact f32) ------------------------------
 * C[M,N] = alpha * op(A) * op(B) + beta * C + bias[n] + bias2[n]     (row-major, ld in floats)
 *   transA=0: A stored [M,K] (lda>=K);  transA=1: A stored [K,M] (lda>=M)
 *   transB=0: B stored [K,N] (ldb>=N);  transB=1: B stored [N,K] (ldb>=K)
 * Supported: NT (0,1), NN (0,0), TN (1,0).  bias/bias2 may be NULL.
 * splitk<=0 -> heuristic; splitk>1 accumulates partials with f32 atomics (beta must be 0 or 1).
 * Replaces: torch Linear / the LSTM input projection inside nn.LSTM (src/module.py:131,
 * src/asr.py:96,220,280,290) and their autograd GEMMs. */
int asrk_gemm_f32(int transA, int transB, int M, int N, int K, float alpha,
                  const float *A, int lda, const float *B, int ldb, float beta,
                  float *C, int ldc, const float *bias, const float *bias2, int splitk,
                  int flags, void *ws, size_t ws_bytes, void *stream);

/* `flags` of asrk_gemm_f32 / asrk_gemm_ws_bytes / asrk_gemm_takes_split (per CALL - the library keeps no
 * mode state):
 *  bits 0-1  arithmetic of large contractions.  They run on the bf16 matrix cores by EXACT operand splitting
 *            (csrc/gemm_split.hip): every f32 operand is the exact sum of three bf16 numbers; the six partial
 *            products down to 2^-24 of the full product are accumulated in f32 (the three dropped ones are each
 *            below one f32 rounding of the product), so the result has f32-GEMM accuracy (checked against
 *            float64 in the tests) at 6/16 of the f32-MFMA cost.
 *              ASRK_GEMM_SPLIT_AUTO   when it pays (both output extents and K large) - the default (0)
 *              ASRK_GEMM_SPLIT_OFF    never: always v_mfma_f32_32x32x2_f32
 *              ASRK_GEMM_SPLIT_ALWAYS whenever the shape allows (K >= 8)
 *  bits 8-15 ASRK_GEMM_LDS_HINT(kib): request at least `kib` KiB of LDS per workgroup of the f32 tiled kernel
 *            (> 80 = one workgroup per CU instead of two), for GEMMs launched in the background of
 *            latency-critical kernels on another stream; 0 = no hint.
 * Workspace: a call that takes the split path writes the bf16 panels of both operands into `ws`
 * (caller-owned device memory, 16-byte aligned, at least asrk_gemm_ws_bytes(M, N, K, flags) bytes, private to
 * the call until the stream has passed it); other calls ignore ws (NULL / 0 allowed).  ASRK_EWORKSPACE if
 * it is missing or too small.  Replaces nothing in the reference by itself: it is the arithmetic behind the
 * same ATen GEMMs (src/module.py:131, src/asr.py:96,220 and their autograd contractions). */
#define ASRK_GEMM_SPLIT_AUTO 0
#define ASRK_GEMM_SPLIT_OFF 1
#define ASRK_GEMM_SPLIT_ALWAYS 2
#define ASRK_GEMM_LDS_HINT(kib) (((kib) & 0xff) << 8)
size_t asrk_gemm_ws_bytes(int M, int N, int K, int flags);
/* 1 if asrk_gemm_f32 runs an M x N x K contraction on the split path under `flags` */
int asrk_gemm_takes_split(int M, int N, int K, int flags);

/* Split panels as operands of their own: split an operand ONCE, multiply it several times (the weight
 * gradients dW_ih = dG^T X and dW_hh = dG^T H_prev of an LSTM layer share dG^T; autograd of nn.LSTM,
 * src/module.py:131).  A panel holds the three bf16 planes of a logical [rows][K] operand (K = contraction
 * index); `trans` != 0: src is stored [K][rows] (ld >= rows), else [rows][K] (ld >= K).  The buffer is the
 * caller's (asrk_split_panel_bytes bytes, 16-byte aligned).
 * asrk_gemm_panels_f32: C[M,N] = alpha * A[a_row0 .. a_row0+M, a_k0 .. a_k0+K] * B[b_row0 .. b_row0+N,
 * b_k0 .. b_k0+K]^T + beta * C + bias + bias2 with A, B given as panels of the stated full extents
 * (a_rows x a_K, b_rows x b_K).  Row offsets must be multiples of 128, k offsets multiples of 8; K must be a
 * multiple of 32 unless the range ends at the end of one of the panels (ASRK_ESHAPE otherwise).  `flags`: reserved,
 * must be 0. */
size_t asrk_split_panel_bytes(int rows, int K, int flags);
int asrk_split_panel_f32(const float *src, int ld, int rows, int K, int trans, void *panel, int flags, void *stream);
int asrk_gemm_panels_f32(int M, int N, int K, float alpha, const void *A_panel, int a_rows, int a_K,
                         int a_row0, int a_k0, const void *B_panel, int b_rows, int b_K, int b_row0, int b_k0,
                         float beta, float *C, int ldc, const float *bias, const float *bias2, int flags,
                         void *stream);

/* ---- strided 3-D copy: dst[i0][i1][0:n2] = src[i0][i1][0:n2] (strides in floats) ------
 * Used for [B,T,D]<->[T,B,D] and the pyramid 'concat'/'drop' time reduction
 * (src/module.py:141-153). accumulate!=0 -> dst += src. */
int asrk_copy3d_f32(const float *src, float *dst, int n0, int n1, int n2,
                    int64_t s_stride0, int64_t s_stride1, int64_t d_stride0, int64_t d_stride1,
                    int accumulate, void *stream);

/* out[n] (+)= sum_m X[m, n]   (X row-major [M,N], ldx floats); bias gradients. */
int asrk_colsum_f32(const float *X, int M, int N, int ldx, float *out, int accumulate,
                    void *stream);

/* elementwise helpers (activation epilogues that are not fused into a GEMM) */
int asrk_tanh_fwd_f32(const float *x, float *y, int64_t n, void *stream);
/* dx = dy * (1 - y^2) */
int asrk_tanh_bwd_f32(const float *y, const float *dy, float *dx, int64_t n, void *stream);

/* ---- row log-softmax (src/asr.py:96, src/decode.py:93,121) -----------------------------
 * y[r,:] = x[r,:] - logsumexp(x[r,:]); in-place allowed (y==x). */
int asrk_log_softmax_fwd_f32(const float *x, float *y, int rows, int cols, int ld,
                             void *stream);
/* dx[r,:] = dy[r,:] - exp(y[r,:]) * sum(dy[r,:]); in-place allowed (dx==dy). */
int asrk_log_softmax_bwd_f32(const float *y, const float *dy, float *dx, int rows, int cols,
                             int ld, void *stream);

/* ---- row top-k / arg-max (src/asr.py:136-142 greedy feedback, src/decode.py:124,150 beam
 * pruning, bin/test_asr.py:118-120): values [rows,k] descending, indices [rows,k] int64; ties
 * resolve to the smaller index (k = 1 is the first arg-max).  NaNs are never selected. */
int asrk_topk_f32(const float *x, int rows, int cols, int ld, int k, float *values,
                  int64_t *indices, void *stream);

/* ---- validation read-out (bin/train_asr.py:169-217 -> src/util.py:113-127 cal_er) -------------------------
 * token_crop: the rule every reference text encoder's decode() applies to a row of (arg-max) token ids
 * (src/text.py:61-71): stop at the first eos_idx, drop pad_idx, and with ignore_repeat != 0 drop an id equal
 * to the RAW previous element (CTC repeat merging).  ids [B, ld] int64 (T used per row) -> out [B, ld_out]
 * compacted in place order, out_len [B] int32.  One wave per row.
 * edit_distance: unit-cost Levenshtein distance of B independent pairs of int64 symbol sequences
 * (a [B, lda] with a_len, b [B, ldb] with b_len <= max_b_len <= 4096) -> dist [B] int32; replaces
 * editdistance.eval at src/util.py:126 (third-party, absent: the classic dynamic programme).  Symbols are
 * whatever the caller interned (token ids for token-level rates, word ids for WER). */
int asrk_token_crop_i64(const int64_t *ids, int64_t ld, int B, int T, int64_t pad_idx, int64_t eos_idx,
                        int ignore_repeat, int64_t *out, int64_t ld_out, int32_t *out_len, void *stream);
int asrk_edit_distance_i64(const int64_t *a, int64_t lda, const int32_t *a_len, const int64_t *b, int64_t ldb,
                           const int32_t *b_len, int B, int max_b_len, int32_t *dist, void *stream);

/* ---- fused softmax cross-entropy (bin/train_asr.py:47,130-131: CrossEntropyLoss(ignore_index=0))
 * fwd: row_lse[r] = logsumexp(logits[r,:]); sums[0] = sum over counted rows of (lse - logit[tgt]),
 *      sums[1] = number of counted rows (targets != ignore_index).  mean loss = sums[0]/sums[1].
 * bwd: dlogits[r,:] = (softmax(logits[r,:]) - onehot(tgt[r])) * gscale[0]  (0 for ignored rows);
 *      gscale is a DEVICE scalar (grad_out / count) so no host sync is needed. */
int asrk_cross_entropy_fwd_f32(const float *logits, int rows, int V, int ld,
                               const int64_t *targets, int ignore_index, float *row_lse,
                               float *sums, void *stream);
int asrk_cross_entropy_bwd_f32(const float *logits, int rows, int V, int ld,
                               const int64_t *targets, int ignore_index, const float *row_lse,
                               const float *gscale, float *dlogits, void *stream);

/* ---- bidirectional LSTM recurrence, persistent kernels (src/module.py:131 -> ATen lstm) ----
 * Time-major layout.  G: [T*B, ldg] with ldg = ndir*4H, column = dir*4H + gate*H + unit
 * (gate order i,f,g,o as torch).  On entry G holds X*W_ih^T + b_ih + b_hh; on exit the
 * ACTIVATED gates (saved for backward).  Y,C: [T*B, ndir*H] hidden / cell states.
 * whh_f/whh_r: [4H,H] torch layout (whh_r ignored when ndir==1).  Zero initial state
 * (module.py:131 passes none).  ws: asrk_lstm_ws_bytes() bytes of device scratch
 * (error word). */
size_t asrk_lstm_ws_bytes(void);
/* `flags` of every recurrence entry point and of the two plan queries below (the queries must be asked with
 * the flags of the launch they size): ASRK_REC_F32_MFMA = form the recurrent products with
 * v_mfma_f32_16x16x4_f32 instead of the exact bf16x6 operand split (wide layers, H = 512 / 1024, use the
 * split by default; results agree to f32 rounding).  0 = default. */
#define ASRK_REC_F32_MFMA 1
/* ASRK_REC_REARM: the launch hands `xchg` back ARMED - every byte 0xFF again once the stream gets past this call -
 * so the NEXT launch of the same shape and flags on this buffer may pass xchg_prefilled = 1 and no fill pass ever
 * runs in front of a recurrence (the workgroups refill the region of step s - 2 while they compute step s; a small
 * fill behind the kernel covers the last two steps).  A launch that ends in ASRK_ETIMEOUT (asrk_lstm_check_error)
 * leaves the buffer in an undefined state: fill or drop it. */
#define ASRK_REC_REARM 2
/* Bytes of the inter-workgroup EXCHANGE buffer a launch needs (fragment-ordered h_t / dG_t of
 * every step; the kernels pre-fill it with a NaN sentinel and poll the data itself). 0 = shape
 * unsupported. backward: 0 for rec_fwd, 1 for rec_bwd. */
size_t asrk_lstm_xchg_bytes(int T, int B, int H, int ndir, int backward, int flags);
/* Workgroups (= CUs, one each) a launch of the persistent kernel occupies for this shape; 0 = shape
 * unsupported.  Lets a caller decide what may usefully run beside it on another stream. */
int asrk_lstm_plan_workgroups(int T, int B, int H, int ndir, int backward, int flags);
/* xchg_prefilled != 0: the caller has already set every byte of `xchg` to 0xFF (e.g. on another
 * stream, off the critical path) since its last use; otherwise the launch fills it first. */
int asrk_lstm_rec_fwd_f32(float *G, const float *whh_f, const float *whh_r, float *Y, float *C,
                          int T, int B, int H, int ndir, void *xchg, int xchg_prefilled, void *ws,
                          int flags, void *stream);
/* The same with the time reduction that follows the layer (src/module.py:141-153) fused into the
 * output store: besides Y the kernel writes Y2, the tensor the NEXT layer reads.
 *   pyr_mode 1 ('concat'): Y2 [T/r, B, r*ndir*H], Y2[t/r][b][(t%r)*ndir*H + col] = Y[t][b][col] for
 *                          t < (T/r)*r (trailing frames dropped);
 *   pyr_mode 2 ('drop')  : Y2 [ceil(T/r), B, ndir*H] = Y[0::r];      pyr_mode 0: Y2 unused.
 * The backward variant reads dY in that layout (the gradient w.r.t. Y2; frames that were dropped
 * get zero), so neither direction needs a separate strided-copy pass.  When the reduced tensor is EMPTY
 * ('concat' with T < r) Y2 / dY may be NULL. */
int asrk_lstm_rec_fwd_pyr_f32(float *G, const float *whh_f, const float *whh_r, float *Y, float *C,
                              int T, int B, int H, int ndir, void *xchg, int xchg_prefilled, void *ws,
                              float *Y2, int pyr_mode, int pyr_rate, int flags, void *stream);
int asrk_lstm_rec_bwd_pyr_f32(float *gates, const float *whh_f, const float *whh_r, const float *C,
                              const float *dY, int T, int B, int H, int ndir, void *xchg,
                              int xchg_prefilled, void *ws, float *db, int pyr_mode, int pyr_rate,
                              int flags, void *stream);
/* Panels written by their producers (round 5).  The forward variant ALSO stores the layer's output as the row-major
 * split panel (asrk_split_panel_bytes(rows, K, 0) bytes, ZERO-initialised by the caller once: the kernel overwrites the
 * real extent, the padding stays zero) of the tensor the next layer multiplies - rows (t / r, b), K = r * ndir * H for
 * pyr_mode 1, rows (t, b), K = ndir * H for pyr_mode 0 - so that layer's input projection (asrk_gemm_panels_f32) needs no
 * split pass over it; the BPTT variant stores dG [T*B, ndir*4H] as the panel of dX = dG W_ih likewise (dg_panel,
 * may be NULL) and, in dgt_panel (may be NULL; needs B % 16 == 0, ASRK_ESHAPE otherwise), the TRANSPOSED panel dG^T
 * [ndir*4H rows, K = T*B tokens] (asrk_split_panel_bytes(ndir*4H, T*B, 0) bytes, zero-initialised once) that the
 * layer's weight gradients dW_ih = dG^T X, dW_hh = dG^T H_prev take row / k ranges of.  Only the
 * bf16x6 kernels emit panels (asrk_lstm_plan_is_bf; ASRK_ESHAPE otherwise; no 'drop' reduction, no GRU). */
int asrk_lstm_plan_is_bf(int T, int B, int H, int ndir, int backward, int flags);
int asrk_lstm_rec_fwd_pyr_panel_f32(float *G, const float *whh_f, const float *whh_r, float *Y, float *C,
                                    int T, int B, int H, int ndir, void *xchg, int xchg_prefilled, void *ws,
                                    float *Y2, int pyr_mode, int pyr_rate, void *x2_panel, int flags, void *stream);
int asrk_lstm_rec_bwd_pyr_panel_f32(float *gates, const float *whh_f, const float *whh_r, const float *C,
                                    const float *dY, int T, int B, int H, int ndir, void *xchg,
                                    int xchg_prefilled, void *ws, float *db, int pyr_mode, int pyr_rate,
                                    void *dg_panel, void *dgt_panel, int flags, void *stream);
/* Inference form with PER-ROW sequence lengths `lens` [B] (int64, device): row b runs steps s < lens[b] only and the
 * reverse direction starts at ITS last frame - what nn.LSTM computes when the reference encodes that utterance alone
 * and unpadded, which is how it decodes (bin/test_asr.py:163-167, src/decode.py:64,88: batch 1).  This lets U
 * utterances of different lengths share one encoder pass with results equal to U batch-1 passes.  Frames
 * t >= lens[b] of Y / Y2 / G / C are NOT written: zero-fill Y / Y2 first.  'concat' reduction trims lens[b] % r
 * frames of every row by itself.  No backward counterpart (decoding only). */
int asrk_lstm_rec_fwd_len_f32(float *G, const float *whh_f, const float *whh_r, float *Y, float *C,
                              const int64_t *lens, int T, int B, int H, int ndir, void *xchg,
                              int xchg_prefilled, void *ws, float *Y2, int pyr_mode, int pyr_rate, int flags,
                              void *stream);
/* torch.nn.GRU layers (module 'GRU' of src/module.py:112-113,131 and src/lm.py:20; gate order r, z, n)
 * in the same persistent kernels.  G [T*B, ndir*4H], per direction four H-wide blocks:
 *   in : x W_ir^T + b_ir + b_hr | x W_iz^T + b_iz + b_hz | x W_in^T + b_in | b_hn (every row)
 *   out: r | z | n | W_hn h_{t-1} + b_hn            whh_*: [3H, H] (nn.GRU weight_hh layout)
 * The BPTT entry point reads those blocks plus Y (the forward outputs, = h_{t-1} of the next step) and
 * leaves  dr | dz | dn | dn*r  in `gates`: blocks 0..2 are the input-side gate gradients (dX, dW_ih,
 * db_ih), blocks 0, 1, 3 the hidden-side ones (dW_hh, db_hh); db [ndir*4H] = their column sums.
 * xchg / ws / pyr_* exactly as for the LSTM entry points (same plans: asrk_lstm_xchg_bytes). */
int asrk_gru_rec_fwd_f32(float *G, const float *whh_f, const float *whh_r, float *Y, int T, int B, int H,
                         int ndir, void *xchg, int xchg_prefilled, void *ws, float *Y2, int pyr_mode,
                         int pyr_rate, int flags, void *stream);
int asrk_gru_rec_bwd_f32(float *gates, const float *whh_f, const float *whh_r, const float *Y,
                         const float *dY, int T, int B, int H, int ndir, void *xchg, int xchg_prefilled,
                         void *ws, float *db, int pyr_mode, int pyr_rate, int flags, void *stream);
/* Backward through time.  gates = activated gates from fwd (overwritten IN PLACE with the
 * pre-activation gradients dG, same layout); dY: [T*B, ndir*H] gradient w.r.t. Y (read only).
 * db (optional, [ndir*4H]): the bias gradient colsum(dG), accumulated by the kernel while it produces
 * dG (every workgroup sees all T steps of its units), so no extra pass over dG is needed.
 * Afterwards: dX = dG*W_ih, dW_ih = dG^T*X, dW_hh = dG^T*Y(t-1). */
int asrk_lstm_rec_bwd_f32(float *gates, const float *whh_f, const float *whh_r, const float *C,
                          const float *dY, int T, int B, int H, int ndir, void *xchg,
                          int xchg_prefilled, void *ws, float *db, int flags, void *stream);
/* Copies the in-kernel error word to host after synchronising `stream`; 0 or ASRK_ETIMEOUT. */
int asrk_lstm_check_error(void *ws, void *stream);

/* ---- pure-CTC prefix beam search on the device (CTCBeamDecoder.forward, src/ctc.py:241-352) ---------------
 * Graves-2014 prefix search with the reference's bookkeeping (expansion order, de-duplication by a stable sort
 * on the decimal string of the tokens, first-maximum survivors, stable score sort, length-normalised final
 * ranking) as ONE workgroup per utterance; hypotheses are identical to the reference's.
 *   ctc [T, V]: the log-probabilities the reference hands to its loop (log_softmax applied twice, src/ctc.py:250);
 *   allowed [V] bytes: 1 for members of vocab_range (ascending id order = the reference's tie order);
 *   beam <= 32, beam * (cand + 1) <= 1024, V <= 99999;
 *   lm: NULL, or [beam, V] RNN-LM log-probabilities of the CURRENT beam rows (row i = hypothesis i), weight
 *       lm_weight; with an LM exactly one frame per call (t1 == t0 + 1): after the call the caller steps the LM
 *       for the rows whose gather index is >= beam (new state row = index - beam; otherwise the state of old
 *       row `index` is inherited) - parent row / last token / gather index per new row are in the workspace;
 *   frames [t0, t1) of T are processed (the caller skips leading frames whose arg-max is blank, src/ctc.py:265);
 *   init != 0 resets the beam to the single empty hypothesis first; cur_buf says which of the two beam buffers
 *   holds the beam at t0 (it alternates every frame: after the call it is cur_buf ^ ((t1 - t0) & 1));
 *   lm_step_follows: t < T - 1 and an LM is fused (src/ctc.py:342).
 * ws: asrk_ctc_prefix_beam_ws_bytes(beam, T) bytes, 16-byte aligned, kept between calls of one utterance;
 * asrk_ctc_prefix_beam_ws_offsets gives the byte offsets of the results (all int32): live-row count, lengths
 * [beam], tokens [beam][T + 1] of buffer `buf`, and parent / last-token / gather-index [beam]. */
size_t asrk_ctc_prefix_beam_ws_bytes(int beam, int T);
int asrk_ctc_prefix_beam_ws_offsets(int beam, int T, int buf, int64_t *nb_off, int64_t *len_off, int64_t *tok_off,
                                    int64_t *parent_off, int64_t *last_off, int64_t *gidx_off);
int asrk_ctc_prefix_beam_f32(const float *ctc, int T, int V, const unsigned char *allowed, int beam, int cand,
                             const float *lm, float lm_weight, int t0, int t1, int cur_buf, int init,
                             int lm_step_follows, void *ws, size_t ws_bytes, void *stream);

/* ---- attention decoder step (src/module.py:179-258, src/asr.py:277-313) -------------------
 * BN = B*num_head rows ordered (b, head).  All tensors contiguous f32; lens int64 [B].
 * loc_conv:  c[b,t,k] = sum_{n,j} prev_att[b,n,t+j-ks] * Wc[k,n,j]   (Conv1d(N,K,2ks+1,pad ks))
 * energy:    loc=1: e = we . tanh(key + q + tanh(Wp c)) + be ; loc=0: e = key . q
 *            attn = softmax(e / temperature) over t < lens[b], 0 beyond (masked_fill(-inf)).
 *            e_scratch: [BN,T] floats of scratch for the scaled energies.
 * context:   ctx[bn,:] = sum_t attn[bn,t] * value[bn,t,:]  (row stride ctx_stride, so it can be
 *            written straight into the decoder's [emb | ctx] input buffer).
 * Backward kernels ACCUMULATE (+=) into *_acc buffers owned by the caller (zeroed once per
 * sequence): dkey_acc [BN,T,A], dWc_acc, dWp_acc, dwe_acc, dbe_acc.  dq/dattn/dprev_att/dc are
 * per-step outputs (dc is scratch [B,T,K], zeroed inside). */
int asrk_loc_conv_fwd_f32(const float *prev_att, const float *Wc, float *c, int B, int N, int T,
                          int K, int ks, void *stream);
int asrk_loc_conv_bwd_f32(const float *dc, const float *prev_att, const float *Wc,
                          float *dprev_att, float *dWc_acc, int B, int N, int T, int K, int ks,
                          void *stream);
int asrk_attn_energy_fwd_f32(int loc, const float *key, const float *q, const float *c,
                             const float *Wp, const float *we, const float *be,
                             const int64_t *lens, float *attn, float *e_scratch, int B, int N, int T,
                             int A, int K, float temperature, void *stream);
int asrk_attn_energy_bwd_f32(int loc, const float *key, const float *q, const float *c,
                             const float *Wp, const float *we, const int64_t *lens,
                             const float *attn, const float *dattn, float *dkey_acc, float *dq,
                             float *dc, float *dWp_acc, float *dwe_acc, float *dbe_acc, int B,
                             int N, int T, int A, int K, float temperature, void *stream);
int asrk_attn_context_fwd_f32(const float *attn, const float *value, float *ctx, int BN, int T,
                              int Dv, int64_t ctx_stride, void *stream);
int asrk_attn_context_bwd_f32(const float *dctx, const float *value, float *dattn, int BN, int T,
                              int Dv, int64_t dctx_stride, void *stream);

/* ---- one LSTM cell step (src/asr.py:218 decoder nn.LSTM on a length-1 sequence) ----------
 * gates [B,4H] = x W_ih^T + h W_hh^T + b (i,f,g,o) -> activated in place; c = f*c_prev + i*g;
 * h = o*tanh(c).  bwd: gates (activated) -> pre-activation grads in place; dh/dc_in may be NULL. */
int asrk_lstm_cell_fwd_f32(float *gates, const float *c_prev, float *c, float *h, int B, int H,
                           void *stream);
int asrk_lstm_cell_bwd_f32(float *gates, const float *c_prev, const float *c, const float *dh,
                           const float *dc_in, float *dc_prev, int B, int H, void *stream);

/* ---- GRU cell, one step (module: 'GRU' of src/module.py:112-113, src/asr.py:175-176, src/lm.py:20-21)
 * gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh: rows of 3H (gate order r, z, n; row strides ldi/ldh).
 * fwd: h_new = (1-z) n + z h_prev; r, z, n are written over gi (gh keeps gh_n).  h_prev NULL = zeros.
 * bwd: the gradient of h' is dh + dh2 (either may be NULL; dh2 is contiguous [B,H]: the carried
 *      recurrent gradient); gi/gh (as left by fwd) are overwritten with dL/dgi and dL/dgh; dh_prev = dh * z (the caller
 * adds dgh W_hh).  Then dx = dgi W_ih, dW_ih = dgi^T x, dW_hh = dgh^T h_prev, db = column sums. */
int asrk_gru_cell_fwd_f32(float *gi, const float *gh, int64_t ldi, int64_t ldh, const float *h_prev,
                          int64_t ldp, float *h_new, int64_t ldn, int B, int H, void *stream);
int asrk_gru_cell_bwd_f32(float *gi, float *gh, int64_t ldi, int64_t ldh, const float *h_prev,
                          int64_t ldp, const float *dh, int64_t ldd, const float *dh2, float *dh_prev,
                          int64_t ldo, int B, int H, void *stream);

/* ---- embedding gather / scatter-add (src/asr.py:103-109,134,142: nn.Embedding) ------------ */
int asrk_embedding_fwd_f32(const int64_t *idx, const float *W, float *out, int64_t n, int D, int V,
                           void *stream);
int asrk_embedding_bwd_f32(const int64_t *idx, const float *dout, float *dW_acc, int64_t n, int D,
                           int V, void *stream);

/* ---- the attention-decoder ("speller") loop as one call per direction ---------------------------
 * Replaces the teacher-forced decode loop of ASR.forward (src/asr.py:112-148 with tf_rate == 1) and
 * the autograd graph under it: per step Attention.forward (src/asr.py:277-313) ->
 * LocationAwareAttention.forward (src/module.py:234-258) -> Decoder.forward (src/asr.py:214-221).
 * Single head, location-aware attention, single-layer LSTM (or, `cell` = 1, GRU) decoder.  All buffers are caller-owned;
 * "tape" buffers written by the forward call are read by the backward call.
 *
 *   dims     B batch, Te encoder frames, A attention dim, Dv encoder feature dim, K/ks location
 *            kernels / half width (2ks+1 taps), H decoder dim, E embedding dim, L decode steps
 *   memory   key [B,Te,A] = tanh(proj_k(enc)), value [B,Te,Dv] = enc, lens [B]
 *   weights  Wq [A,H], bq [A]; Wc [K,1,2ks+1]; Wp [A,K]; we [A]; be [1]; W_ih [4H,E+Dv]; W_hh [4H,H];
 *            b_ih, b_hh [4H] (only read by asrk_speller_step_f32)
 *   eproj    [L,B,4H] = emb_t W_ih[:, :E]^T + b_ih + b_hh for the known (teacher) inputs of all steps
 *   tape     q [L,B,A]; conv [L,B,Te,K]; attn: row (b, step t) at attn + b*attn_ld + t*attn_step
 *            (att_seq [B,1,L,Te]: attn_ld = L*Te, attn_step = Te); ctx [L,B,Dv]; gates [L,B,4H]
 *            (activated i,f,g,o; the backward call turns them into pre-activation gradients in
 *            place); h, c [L+1,B,H] (slot 0 = initial state, written by the caller);
 *            states [B,L,H] = h_1..h_L batch-major (optional); e_scratch [B,Te];
 *            prev0 [B,Te] = the uniform attention that feeds step 0 (src/module.py:239-242). */
typedef struct asrk_speller {
    int B, Te, A, Dv, K, ks, H, E, L;
    float temperature;
    int shared_kv;   /* != 0 (asrk_speller_step_f32 only): key [1,Te,A], value [1,Te,Dv], lens [1] are one
                        utterance shared by all B rows (the live hypotheses of a beam search) */
    const float *key, *value;
    const int64_t *lens;
    const float *Wq, *bq, *Wc, *Wp, *we, *be, *W_ih, *W_hh, *b_ih, *b_hh;
    const float *eproj;
    float *q, *conv, *attn;
    int64_t attn_ld, attn_step;
    float *ctx, *gates, *h, *c, *states, *e_scratch;
    const float *prev0;
    const int *row_mem;   /* optional (asrk_speller_step_f32 only): batch row b attends over memory row row_mem[b] of
                             key [U,Te,A] / value [U,Te,Dv] / lens [U] - the beams of U utterances decoded together
                             (reference fan-out over utterances: bin/test_asr.py:163-167); NULL: row b / shared_kv */
    int cell;             /* 0: LSTM decoder cell (gates i,f,g,o).  1: GRU cell (src/asr.py:172 with module 'GRU')
                             in the SAME four-rows-per-unit layout:
                             rows [0,H) r, [H,2H) z, [2H,3H) n_x = W_in x + b_in, [3H,4H) n_h = W_hn h + b_hn, i.e.
                             W_ih = [W_ir; W_iz; W_in; 0], W_hh = [W_hr; W_hz; 0; W_hn] (the caller stacks them);
                             h' = (1 - z) tanh(n_x + r n_h) + z h.  The gates tape holds r, z, n, n_h; c is not used */
    /* stacked LSTM decoder (nn.LSTM(num_layers = nlayer), src/asr.py:175-176; round 6): nlayer 0 / 1 = one layer (the
       fields above), up to ASRK_SPELLER_MAX_LAYERS.  Upper layer l = 1.. uses slot l - 1 of: Wu_ih [4H,H], Wu_hh
       [4H,H], bu_ih / bu_hh [4H] (read by the forward call), tapes hu / cu [L+1,B,H] (slot 0 = initial state) and gu
       [L,B,4H] (activated gates; pre-activation gradients after the backward call).  Wq is then [A, nlayer*H] (the
       query reads the layer-concatenated state, src/asr.py:207-212), `states` receives the TOP layer's outputs and W_ih
       / W_hh / h / c / gates are layer 0's.  LSTM cells only (cell == 0); asrk_speller_step_f32 takes one layer. */
    int nlayer;
    const float *Wu_ih[2], *Wu_hh[2], *bu_ih[2], *bu_hh[2];
    float *hu[2], *cu[2], *gu[2];
    /* attention variants of the loop (round 6; asrk_speller_step_f32 takes neither).  att_mode 0: location-aware (all of
       the above); 1: ScaleDotAttention (src/module.py:198-212): e = q . key / temperature, no Wc / Wp / we / be / conv /
       prev0 (K = ks = 0 allowed).  nhead N > 1 (dot only; the location-aware form convolves ACROSS the heads' previous
       alignments, src/module.py:221,244): key [B*N,Te,A], value [B*N,Te,Dv] and lens [B] as src/asr.py:294-304 builds
       them (row b*N + n; heads of an utterance share lens[b]); Wq [N*A, nlayer*H]; the tapes q [L,B,N*A], attn (rows
       b*N + n, same attn_ld / attn_step addressing) and e_scratch [B*N,Te] hold B*N rows; ctxh [L,B,N*Dv] receives the
       heads' contexts and ctx = ctxh Wm^T + bm (merge_head, Wm [Dv, N*Dv], src/asr.py:308-311). */
    int att_mode, nhead;
    const float *Wm, *bm;
    float *ctxh;
    /* asrk_speller_step_f32 with row_mem only: RG > 1 = the batch rows [g*RG, (g+1)*RG) all attend over memory
       row_mem[g*RG] (the beam of one utterance in consecutive rows, B % RG == 0, RG <= 32): the context kernel then reads
       an utterance's value memory once per utterance instead of once per row.  0 / 1: no such promise. */
    int row_group;
} asrk_speller_t;
#define ASRK_SPELLER_MAX_LAYERS 3

/* backward-only buffers.  dstates [B,L,H] = dLoss/dh_t (batch-major, from the vocabulary projection);
 * dattn_seq = gradient of att_seq (same addressing as attn) or NULL; WT [(Dv+H),4H] =
 * [W_ih[:,E:] | W_hh]^T and WqT [H,A] = Wq^T (asrk_transpose_ld_f32).
 * Accumulated (+=, zero them first): dkey [B,Te,A]; per-workgroup partial sums dwe_part [B*tc,A],
 * dWp_part [B*tc,A*K], dbe_part [B*tc], dWc_part [B,K*(2ks+1)] (reduce over the leading axis
 * afterwards); tc from asrk_speller_plan.  Written: dxh [L,B,Dv+H] (dctx_t | dh via W_hh),
 * dq_pre [L,B,A] = dq_t (1 - q_t^2).  Scratch: dattn, dprev [B,Te]; dconv [B,Te,K];
 * dq_part [B*tc,A]; dc [B,H].  After the call `gates` holds dG [L,B,4H]; the weight gradients are
 * whole-sequence GEMMs over the tape (dW_ih = dG^T [emb|ctx], dW_hh = dG^T h_{0..L-1},
 * dWq = dq_pre^T h_{0..L-1}, dvalue[b] = attn[b]^T dctx[b], demb = dG W_ih[:, :E]). */
typedef struct asrk_speller_bwd {
    const float *dstates, *dattn_seq, *WT, *WqT;
    float *dkey, *dxh, *dq_pre, *dattn, *dprev, *dconv, *dq_part, *dwe_part, *dWp_part, *dbe_part,
        *dWc_part, *dc;
    int tc;
    /* stacked decoder, upper layer l = 1.. in slot l - 1: WuT [2H,4H] = [W_ih_l | W_hh_l]^T; dxu [L,B,2H] written
       (d h of the layer below at the same step | d h of layer l's previous step); dcu [B,H] scratch.  WqT is
       [nlayer*H, A]; the upper layers' weight gradients are dW_ih_l = dG_l^T h_{l-1}[1..L], dW_hh_l = dG_l^T h_l[0..L-1]. */
    const float *WuT[2];
    float *dxu[2], *dcu[2];
    /* nhead > 1: WmT [N*Dv, Dv] = Wm^T; dctxh [L,B,N*Dv] written (gradient of the heads' contexts: dvalue's operand, and
       dWm = dctx^T ctxh).  With several heads dkey is [B*N,Te,A], dq_pre [L,B,N*A], dattn [B*N,Te], dq_part [B*N*tc,A];
       the location-only buffers (dprev, dconv, dwe_part, dWp_part, dbe_part, dWc_part) may be NULL for att_mode 1. */
    const float *WmT;
    float *dctxh;
} asrk_speller_bwd_t;

/* number of frame chunks (workgroups per utterance) the energy kernels will use: sizes the
 * per-workgroup partial buffers above */
int asrk_speller_plan(const asrk_speller_t *dims, int *tc_fwd, int *tc_bwd);
int asrk_speller_fwd_f32(const asrk_speller_t *p, void *stream);
int asrk_speller_bwd_f32(const asrk_speller_t *p, const asrk_speller_bwd_t *g, void *stream);
/* one attention + decoder-cell step outside the training loop (greedy / beam decoding,
 * src/decode.py:110-121): tape slot `slot` (h, c slots slot -> slot+1), previous attention rows at
 * prev_att + b*prev_ld, embedded previous tokens emb [B,E]; eproj is not read (the embedding goes
 * through W_ih[:, :E] inside the step, plus b_ih + b_hh).  emb == NULL: the attention half only (query from h slot
 * `slot`, energies, alignment, context) - the caller runs the decoder cell itself (W_ih, W_hh, b_*, c not read; with many
 * rows the host layer uses bf16x6 panel GEMMs against weight panels split once per decode).  In that form Wq == NULL
 * means that q slot `slot` already holds the query tanh(h Wq^T + bq) (the same host route). */
int asrk_speller_step_f32(const asrk_speller_t *p, int slot, const float *prev_att, int64_t prev_ld,
                          const float *emb, void *stream);
/* dvalue[b,t',d] = sum_l attn(b, l)[t'] * dxh[l*step_ld + b*row_ld + d]  (overwritten): the gradient of
 * the encoder memory `value` over a whole decode loop from the tape (attn addressing as above, dctx_l
 * = the first Dv columns of dxh's step-l rows). */
int asrk_speller_dvalue_f32(const float *attn, int64_t attn_ld, int64_t attn_step, const float *dxh,
                            int64_t step_ld, int64_t row_ld, float *dvalue, int B, int L, int Te, int Dv,
                            void *stream);
/* joint token scores of one beam-search step (src/decode.py:123-148), rows = live hypotheses:
 *   out = (1 - ctc_weight) * att_logp + ctc_weight * hack,  hack[r, cand[r,c]] = psi[r,c] - prev_ctc[r],
 *   LOG_ZERO elsewhere;  out[:, 0] = LOG_ZERO;  out += lm_weight * lm_logp
 * cand NULL = no CTC term; lm_logp NULL = no LM term; out must not alias att_logp. */
int asrk_joint_score_f32(const float *att_logp, const int64_t *cand, const float *psi,
                         const float *prev_ctc, const float *lm_logp, float *out, int n, int V, int C,
                         double ctc_weight, double lm_weight, double logzero, void *stream);
/* one nn.LSTM step on a length-1 sequence, input projection, recurrent projection and cell update in
 * ONE kernel (decoder / RNN-LM steps of the beam search, src/decode.py:113,143-146, src/lm.py:38):
 * gates = x W_ih^T + h W_hh^T + b_ih + b_hh (x rows at x + m*ldx, In features), then c', h'.
 * Out-of-place: h_out / c_out must not alias h / c. */
int asrk_lstm_cell_fused_f32(const float *x, int64_t ldx, int In, const float *h, const float *c,
                             const float *W_ih, const float *W_hh, const float *b_ih,
                             const float *b_hh, float *h_out, float *c_out, int B, int H, void *stream);
/* out[c*ldo + r] = in[r*ldi + c] */
int asrk_transpose_ld_f32(const float *in, int64_t ldi, float *out, int64_t ldo, int rows, int cols,
                          void *stream);

/* ---- per-frame regularisers (src/module.py:116-119,135-138; src/asr.py:36,162) --------------
 * layer_norm: torch.nn.LayerNorm(cols) over contiguous rows [rows, cols]: biased variance, eps
 *   inside the sqrt; fwd also returns mean/rstd [rows] for the backward.  bwd: dx (may be NULL),
 *   dweight/dbias [cols] (both or neither; overwritten).
 * dropout: inverted dropout y = keep ? x/(1-p) : 0.  keep(i) is a pure function of
 *   (seed, offset, i): word i%4 of Philox4x32-10(counter = (i/4, offset), key = seed), kept when
 *   (word >> 8) >= round(p * 2^24).  No mask is stored: the backward is the same call on dy. */
int asrk_layer_norm_fwd_f32(const float *x, const float *weight, const float *bias, float *y,
                            float *mean, float *rstd, int rows, int cols, float eps, void *stream);
int asrk_layer_norm_bwd_f32(const float *x, const float *weight, const float *dy, const float *mean,
                            const float *rstd, float *dx, float *dweight, float *dbias, int rows,
                            int cols, void *stream);
int asrk_dropout_f32(const float *x, float *y, int64_t n, float p, uint64_t seed, uint64_t offset,
                     void *stream);

/* ---- device-side beam bookkeeping of the joint CTC-attention(-LM) search (src/decode.py:150-167, 209-239) ----------
 * One decode position t of U utterances at once, no read-back.  Utterance u owns the row slots [u*B, (u+1)*B); at
 * position t every live row (alive[row] != 0) is a hypothesis of t labels with score sum ssum[row] (float64: the
 * reference's Python floats).  Inputs: the position's top-B scores / labels of every row (topv / topi [U*B, B]) and, with
 * CTC (C > 0), the C candidates' labels and prefix scores (cand / psi [U*B, C]).  Per utterance the B*B continuations are
 * taken in the reference's record order (hypothesis-major, rank-minor), <eos> (label 1) and labels missing from `cand`
 * are dropped, and the B best by AVERAGE score (ssum + score) / (t + 1), ties in record order, go to the utterance's
 * slots in rank order:  prev_token / parent (row of position t) / col (candidate column) / pctc (psi of that column) /
 * ssum / alive, and row t of the back-pointer history hist_tok / hist_sc / hist_par [lmax][U*B].
 * Finished hypotheses are appended to the utterance's log (fin_* [U][fcap], fin_count [U]): kind 0 = row fin_row of
 * position fin_t followed by <eos> with score fin_term (a row whose top-B holds <eos>, once t >= min_len[u]; the LAST
 * <eos> among its top-B); kind 1 = the continuation in slot fin_row of position fin_t that was alive when the utterance
 * ended (no continuation left, t + 1 >= max_len[u], or B == 1 after its first finished hypothesis - then no kind-1
 * entries).  An ended utterance sets utt_done[u], clears its rows and decrements *live_utts.
 * B <= 32, C <= 48 (ASRK_ESHAPE).  All pointers are device memory. */
int asrk_beam_select_f32(const float *topv, const int64_t *topi, const float *psi, const int64_t *cand, int U, int B, int C,
                         int t, int lmax, int fcap, const int *min_len, const int *max_len, int *alive, double *ssum,
                         int *utt_done, int64_t *prev_token, int64_t *parent, int64_t *col, float *pctc, int *hist_tok,
                         float *hist_sc, int *hist_par, int *fin_count, int *fin_kind, int *fin_t, int *fin_row,
                         float *fin_term, double *fin_ssum, int *live_utts, void *stream);

/* The survivors' states for the next decode position in ONE launch: for segment s < nseg (<= 8) and row i < n
 *   dst[s][i, :] = src[s][parent[i] * mul[s] + (use_col[s] ? col[i] : 0), :]      rows of row_floats[s] floats
 * (decoder h / c and the previous alignment: mul 1; the CTC prefix state r [rows, C, Te, 2]: mul = C, use_col = 1, rows of
 * 2*Te floats; LM state layers: one segment each).  src / dst / row_floats / mul / use_col are HOST arrays of nseg entries;
 * parent / col device int64 [n].  Sources and destinations must not overlap. */
int asrk_gather_rows_multi_f32(int nseg, const float *const *src, float *const *dst, const int *row_floats, const int *mul,
                               const int *use_col, const int64_t *parent, const int64_t *col, int n, void *stream);

/* ---- convolutional prenets (src/module.py:7-90: VGGExtractor / CNNExtractor) -----------------
 * Activations are channels-last [B, H(time), W(freq), C].  A convolution is im2col -> asrk_gemm_f32
 * against weight.view(Cout, Cin*KH*KW) (+bias) -> [B*Ho*Wo, Cout] = the next channels-last tensor.
 * im2col: col[(b,ho,wo), (cin*KH+kh)*KW+kw] = x[b*sb + (ho*SH+kh-PH)*sh + (wo*SW+kw-PW)*sw + cin*sc]
 *         (0 outside the input); Ho/Wo = asrk_conv_out_size(extent, k, stride, pad) (0 = invalid).
 * col2im: the adjoint (gather form, overwrites dx at the same strides).
 * relu_fwd is in place; relu_bwd: dx = y > 0 ? dy : 0.
 * maxpool2x2: stride 2, floor mode, x contiguous [B,H,W,C]; y / dy addressed with explicit output
 *         strides (osb, osh, osw, osc) so the last pool can emit [B, T/4, C*F/4] directly
 *         (src/module.py:62-65); idx [B,H/2,W/2,C] keeps the arg-max (0..3) for the backward, which
 *         writes every element of the contiguous dx. */
int asrk_conv_out_size(int in, int k, int stride, int pad);
int asrk_im2col_f32(const float *x, float *col, int B, int H, int W, int C, int KH, int KW, int SH, int SW,
                    int PH, int PW, int64_t sb, int64_t sh, int64_t sw, int64_t sc, void *stream);
int asrk_col2im_f32(const float *dcol, float *dx, int B, int H, int W, int C, int KH, int KW, int SH,
                    int SW, int PH, int PW, int64_t sb, int64_t sh, int64_t sw, int64_t sc, void *stream);
/* im2col with patch rows ldcol >= C*KH*KW floats apart, columns C*KH*KW .. ldcol-1 written as zeros: a K that is not a
 * multiple of 4 (the first VGG layer: 1-3 channels x 9 taps) padded up so that the GEMMs over the patches take their
 * 16-byte paths (against a weight padded with zero columns; the extra dW columns are dropped). */
int asrk_im2col_ld_f32(const float *x, float *col, int ldcol, int B, int H, int W, int C, int KH, int KW, int SH, int SW,
                       int PH, int PW, int64_t sb, int64_t sh, int64_t sw, int64_t sc, void *stream);
/* The same pair with the K axis ordered (kh, kw, cin) - col[(b,ho,wo), (kh*KW+kw)*C + cin] - for inputs whose channels
 * are contiguous (sc == 1, C % 4 == 0, sb / sh / sw multiples of 4, 16-byte aligned pointers; ASRK_ESHAPE otherwise):
 * both become strided copies in whole 16-byte pieces.  The GEMM then multiplies against the weight re-ordered by
 * asrk_conv_weight_reorder_f32 (weight.view(Cout, Cin, KH*KW) -> [Cout][KH*KW][Cin]; inverse != 0: back, for dW). */
int asrk_im2col_cl_f32(const float *x, float *col, int B, int H, int W, int C, int KH, int KW, int SH, int SW,
                       int PH, int PW, int64_t sb, int64_t sh, int64_t sw, int64_t sc, void *stream);
int asrk_col2im_cl_f32(const float *dcol, float *dx, int B, int H, int W, int C, int KH, int KW, int SH,
                       int SW, int PH, int PW, int64_t sb, int64_t sh, int64_t sw, int64_t sc, void *stream);
int asrk_conv_weight_reorder_f32(const float *src, float *dst, int Cout, int Cin, int KK, int inverse, void *stream);
/* 3x3 / stride 1 / pad 1 convolutions WITHOUT a patch matrix (the VGG prenet's layers with 64 or 128 input channels,
 * src/module.py:21-33: Conv2d(64,64,3,padding=1), Conv2d(64,128,...), Conv2d(128,128,...)): implicit GEMMs on the f32-input
 * matrix cores over contiguous channels-last activations x [B,H,W,C] -> y [B,H,W,Cout].  Same arithmetic as im2col +
 * asrk_gemm_f32 (exact f32 products, f32 accumulation; the summation order differs).
 *   asrk_conv3x3_supported: 1 when the shape has a kernel (C, Cout in {64, 128}, W <= 128), else 0 - callers keep the
 *       im2col path for everything else (the first layer, the CNN prenet).
 *   asrk_conv3x3_weight_f32: the parameter w[Cout][Cin][3][3] in the kernels' fragment order (9*Cin*Cout floats).
 *       transpose == 0: the forward weight; != 0: the data gradient's (cin <-> cout, taps flipped).
 *   asrk_conv3x3_f32: y = conv(x where xmask > 0, wf) + bias, optionally max(., 0).  xmask (shape of x) and bias may be
 *       NULL.  The data gradient is this entry on dy: x = dy, xmask = the layer's ReLU output (or NULL), wf = the
 *       transposed weight, C = the layer's Cout, Cout = the layer's Cin, no bias, no ReLU.
 *   asrk_conv3x3_wgrad_f32: dw[Cout][Cin][3][3] (parameter layout) = sum over positions of (dy where ymask > 0) x patches
 *       of x; db[Cout] = column sums of the same masked dy.  Either output may be NULL; ymask may be NULL.  Deterministic
 *       (per-workgroup partial sums in `ws`, added in a fixed order).  ws: asrk_conv3x3_wgrad_ws_bytes bytes, 16-byte
 *       aligned (ASRK_EWORKSPACE if smaller).
 * ASRK_ESHAPE: unsupported shape or a pointer that is not 16-byte aligned. */
int asrk_conv3x3_supported(int H, int W, int C, int Cout);
int asrk_conv3x3_weight_f32(const float *w, float *wf, int Cout, int Cin, int transpose, void *stream);
int asrk_conv3x3_f32(const float *x, const float *xmask, const float *wf, const float *bias, float *y, int B, int H, int W,
                     int C, int Cout, int relu, void *stream);
size_t asrk_conv3x3_wgrad_ws_bytes(int B, int H, int W, int C, int Cout);
int asrk_conv3x3_wgrad_f32(const float *x, const float *dy, const float *ymask, float *dw, float *db, int B, int H, int W,
                           int C, int Cout, void *ws, size_t ws_bytes, void *stream);
/* The FIRST VGG layer (Conv2d(in_channel, 64, 3, padding=1), in_channel = 1..3 delta-feature planes, src/module.py:21;
 * view_input, 44-57, is folded into the element strides: x[b*sb + h*sh + w*sw + c*sc]): 9*C <= 27 fits one MFMA tile, so
 * forward = one kernel (bias and ReLU fused) writing y [B,H,W,Cout], and the weight + bias gradients = one kernel over dy
 * (gated by ymask > 0 when given) plus a fixed-order reduction of per-workgroup partial sums in `ws`
 * (asrk_conv3x3_first_wgrad_ws_bytes bytes, 16-byte aligned).  w / dw: the parameter's [Cout][C][3][3] layout.
 * C <= 3, Cout a multiple of 64, W <= 128 (asrk_conv3x3_first_supported; ASRK_ESHAPE otherwise).  The gradient with
 * respect to the features is not part of this pair (callers that need it keep the im2col path for that one product). */
int asrk_conv3x3_first_supported(int H, int W, int C, int Cout);
int asrk_conv3x3_first_f32(const float *x, const float *w, const float *bias, float *y, int B, int H, int W, int C, int Cout,
                           int64_t sb, int64_t sh, int64_t sw, int64_t sc, int relu, void *stream);
size_t asrk_conv3x3_first_wgrad_ws_bytes(int B, int H, int W, int C, int Cout);
int asrk_conv3x3_first_wgrad_f32(const float *x, const float *dy, const float *ymask, float *dw, float *db, int B, int H,
                                 int W, int C, int Cout, int64_t sb, int64_t sh, int64_t sw, int64_t sc, void *ws,
                                 size_t ws_bytes, void *stream);
int asrk_relu_fwd_f32(float *x, int64_t n, void *stream);
int asrk_relu_bwd_f32(const float *y, const float *dy, float *dx, int64_t n, void *stream);
int asrk_maxpool2x2_fwd_f32(const float *x, float *y, uint8_t *idx, int B, int H, int W, int C,
                            int64_t osb, int64_t osh, int64_t osw, int64_t osc, void *stream);
int asrk_maxpool2x2_bwd_f32(const float *dy, const uint8_t *idx, float *dx, int B, int H, int W, int C,
                            int64_t osb, int64_t osh, int64_t osw, int64_t osc, void *stream);

/* ---- fused optimiser updates (src/solver.py:76-91: clip_grad_norm_ + torch.optim step) --------
 * One streaming pass per parameter tensor (n elements).  clip_coef: DEVICE scalar holding
 * max_norm / (total_norm + 1e-6) (clamped to <= 1 inside), or NULL for no clipping; the gradient is
 * read-only (the clipped gradient is never written back).
 * adadelta: torch.optim.Adadelta(lr, rho, eps, weight_decay=0) state (square_avg, acc_delta).
 * adam:     torch.optim.Adam(lr, betas, eps, weight_decay=0, amsgrad=False) state (exp_avg,
 *           exp_avg_sq); `step` is the 1-based step count used for the bias corrections.
 * Hyper-parameters are doubles (as Python floats): 1-rho, 1-beta and the bias corrections are formed
 * in double before rounding to f32, as torch does. */
int asrk_adadelta_step_f32(float *param, const float *grad, float *square_avg, float *acc_delta,
                           int64_t n, double lr, double rho, double eps, const float *clip_coef,
                           void *stream);
int asrk_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                       double lr, double beta1, double beta2, double eps, int64_t step,
                       const float *clip_coef, void *stream);
/* Multi-tensor forms: `count` parameter tensors in one call (HOST arrays of device pointers and
 * element counts; one kernel launch per 24 tensors, the table travels in the kernel arguments).
 * Element-wise identical to the single-tensor calls above; all tensors share the hyper-parameters
 * (one torch param_group) and, for adam, the step count. */
int asrk_adadelta_multi_f32(int count, float *const *params, const float *const *grads,
                            float *const *square_avg, float *const *acc_delta, const int64_t *numel,
                            double lr, double rho, double eps, const float *clip_coef, void *stream);
int asrk_adam_multi_f32(int count, float *const *params, const float *const *grads,
                        float *const *exp_avg, float *const *exp_avg_sq, const int64_t *numel,
                        double lr, double beta1, double beta2, double eps, int64_t step,
                        const float *clip_coef, void *stream);

/* Global 2-norm of a list of gradient tensors and the clipping coefficient of clip_grad_norm_(params, max_norm)
 * (src/solver.py:84): norm_out[0] = sqrt(sum_t sum_i g_t[i]^2), coef_out[0] = max_norm / (norm + 1e-6) (either may be
 * NULL), both on the device so that nothing is read back before the optimiser step.  Two deterministic stages
 * (per-block partial sums, one block adds them in a fixed order): bit-reproducible.  ws: asrk_grad_norm_ws_bytes
 * bytes of device scratch, 8-byte aligned.  asrk_fill_f32: x[0:n] = value. */
size_t asrk_grad_norm_ws_bytes(int count, const int64_t *numel);
int asrk_grad_norm_multi_f32(int count, const float *const *grads, const int64_t *numel, float max_norm,
                             float *norm_out, float *coef_out, void *ws, size_t ws_bytes, void *stream);
int asrk_fill_f32(float *x, int64_t n, float value, void *stream);

/* ---- audio front end (src/audio.py:7-133; fbank = torchaudio.compliance.kaldi.fbank) ------
 * frames:  wave [n_samples] f32 -> frames [m, ldf]: snip_edges framing (frame i = samples
 *          [i*shift, i*shift+win)), optional per-frame DC removal, pre-emphasis with replicate
 *          padding, multiplication by `window` [win] (povey); columns >= win are zero.
 * The spectrum is frames x (cos|sin basis) through asrk_gemm_f32; power: spec [m, 2*nb] = [re|im]
 * -> power [m, nb]; mel energies = power x melT through asrk_gemm_f32; log_floor: x =
 * log(max(x, eps)) in place.
 * delta:   x [D,T] -> y [C,D,T], y[c,d,t] = sum_j filters[c,j] * x[d, t+j-(L-1)/2], zero padded
 *          (src/audio.py:48-54; L odd).  cmvn: rows [R,T] -> (x-mean)/(eps+std_unbiased) over T
 *          (src/audio.py:24-27).  transpose: [rows, cols] -> [cols, rows] (Postprocess 85-89). */
int asrk_fbank_frames_f32(const float *wave, int64_t n_samples, const float *window, float *frames,
                          int m, int win, int shift, int ldf, float preemph, int remove_dc,
                          void *stream);
int asrk_power_spectrum_f32(const float *spec, float *power, int64_t m, int nb, void *stream);
int asrk_log_floor_f32(float *x, int64_t n, float eps, void *stream);
int asrk_delta_f32(const float *x, const float *filters, float *y, int C, int D, int T, int L,
                   void *stream);
int asrk_cmvn_f32(const float *x, float *y, int rows, int T, float eps, void *stream);
int asrk_transpose_f32(const float *x, float *y, int rows, int cols, void *stream);
/* The same front end for a whole padded batch (collate: src/data.py:14-43 extracts one file at a time in
 * DataLoader workers and pads afterwards; here the padded PCM of the batch is the input):
 * frames_batch: wave [B, ld_wave] of int16 PCM (sample_bytes 2; scale = 1/32768 reproduces torchaudio.load's
 *          float conversion exactly) or f32 (sample_bytes 4); utterance b owns rows frame_off[b] ..
 *          frame_off[b+1] of frames [total_m, ldf] (frame_off [B+1] int64 on the device, max_m = the longest
 *          utterance's frame count; n_samples_host: optional HOST array [B] checked against ld_wave).
 *          The spectrum / mel / log steps then run once over all total_m rows.
 * delta_cmvn_batch: mel [total_m, D] -> out [B, Tmax, C*D], zero beyond each utterance's frames:
 *          Delta (C = order + 1 filters of L <= 16 odd taps, zero padded at the utterance edges), CMVN over the
 *          utterance's own frames (apply_cmvn != 0; unbiased std, eps added to std), Postprocess's
 *          [T, c*D + d] layout and pad_sequence's zero padding in one kernel. */
int asrk_fbank_frames_batch_f32(const void *wave, int sample_bytes, int64_t ld_wave, const int64_t *n_samples_host,
                                const int64_t *frame_off, int B, int max_m, const float *window, float *frames,
                                int win, int shift, int ldf, float scale, float preemph, int remove_dc,
                                void *stream);
int asrk_delta_cmvn_batch_f32(const float *mel, const int64_t *frame_off, int B, int D, const float *filters, int C,
                              int L, int apply_cmvn, float eps, float *out, int Tmax, void *stream);
/* asrk_fbank_logmel_batch_f32 (round 6): torchaudio.compliance.kaldi.fbank for a whole padded PCM batch in ONE kernel
 * (reference call site src/audio.py:104-108): frame f of utterance b -> row frame_off[b] + f of mel [sum m, nmel] =
 * log(max(mel energies, eps)).  One wave per frame: framing exactly as asrk_fbank_frames_batch_f32 (same arguments),
 * the zero-padded 2^log2n-point real DFT as a 2^(log2n-1)-point complex FFT in LDS, power spectrum, triangular mel
 * weights.  Tables are the caller's: window [win]; tw_fft [2^(log2n-1)] (re, im) pairs of e^{-2 pi i k / 2^(log2n-1)};
 * tw_unpack [2^(log2n-1) + 1] pairs of e^{-2 pi i k / 2^log2n}; melT [>= 2^(log2n-1) + 1 rows][ld_mel] the dense mel
 * weights (row = FFT bin); mel_range [nmel][2] (int32) = first / one-past-last FFT bin with a non-zero weight.
 * log2n in 8..11 and win <= 2^log2n, else ASRK_ESHAPE. */
int asrk_fbank_logmel_batch_f32(const void *wave, int sample_bytes, int64_t ld_wave, const int64_t *n_samples_host,
                                const int64_t *frame_off, int B, int max_m, const float *window, const float *tw_fft,
                                const float *tw_unpack, const float *melT, const int *mel_range, int nmel, int ld_mel,
                                float *mel, int win, int shift, int log2n, float scale, float preemph, int remove_dc,
                                float eps, void *stream);

/* ---- CTC loss (bin/train_asr.py:49,123-124 -> torch.nn.CTCLoss(blank=0)) ---------------
 * log_probs element (t,b,c) at lp[t*stride_t + b*stride_b + c]; targets [B,L] int64 (row stride
 * tgt_stride) zero-padded; input_lengths/target_lengths int64 [B].
 * alpha/beta: [B, T, S] scratch/saved, S = 2*Lmax+1; nll: [B] per-utterance -log p.
 * The scalar reduction (mean over b of nll/len) is done by the caller (host glue). */
int asrk_ctc_loss_fwd_f32(const float *lp, int64_t stride_t, int64_t stride_b, int T, int B,
                          int V, const int64_t *targets, int64_t tgt_stride, int Lmax,
                          const int64_t *input_lengths, const int64_t *target_lengths,
                          int blank, float *alpha, float *beta, float *lpg, float *nll,
                          void *stream);
/* fwd also needs lpg [B, T, S] (scratch: the gathered log-probs lp[t,b,ext_b[s]]) and, when `beta`
 * is non-NULL, computes the beta lattice concurrently with alpha (training: pass it, then bwd is
 * only the gradient assembly; inference: NULL).
 * grad[t,b,c] = (exp(lp) - exp(logsum_{s:ext[s]=c}(alpha+beta) + nll_b - lp)) * gscale[b]
 * for t < input_length[b], else 0  (== ATen ctc_loss backward; gscale folds grad_out and the
 * 'mean' normaliser).  grad uses the same (stride_t, stride_b) addressing as given. */
int asrk_ctc_loss_bwd_f32(const float *lp, int64_t stride_t, int64_t stride_b, int T, int B,
                          int V, const int64_t *targets, int64_t tgt_stride, int Lmax,
                          const int64_t *input_lengths, const int64_t *target_lengths,
                          int blank, const float *alpha, const float *beta, const float *lpg,
                          const float *nll, const float *gscale, float *grad, int64_t g_stride_t,
                          int64_t g_stride_b, void *stream);

/* ---- CTC prefix scores for joint CTC-attention beam search (src/ctc.py:76-116) -----------
 * All (hypothesis h, candidate c) pairs of one beam step in one launch.  x [T,V] log-probs;
 * r_prev [n,T,2] (0 = non-blank, 1 = blank path) of each hypothesis' prefix g_h; prefix_len[h] =
 * |g_h|, last_char[h] = g_h[-1]; candidates [n,C] (int32).  Outputs psi [n,C] = log p_ctc(g_h+c,...)
 * and r_out [n,C,T,2], exactly CTCPrefixScore.cheap_compute (incl. phi = blank-only path when
 * c == g_h[-1], psi[<eos>] = logaddexp(r_prev[-1]), logzero = -1e8 initialisation). */
int asrk_ctc_prefix_score_f32(const float *x, const float *r_prev, const int *prefix_len,
                              const int *last_char, const int *candidates, float *psi,
                              float *r_out, int n, int C, int T, int V, int blank, int eos,
                              float logzero, void *stream);
/* the same for the beams of SEVERAL utterances in one launch: x [U,T,V] (zero-padded to the longest utterance),
 * hypothesis h belongs to utterance row_mem[h] whose log-probabilities have mem_len[row_mem[h]] frames; the
 * recursion of h runs over ITS frames only (r_out / r_prev keep the common stride T; entries beyond the
 * utterance's length stay `logzero` and are never read), psi[<eos>] looks at its last frame. */
int asrk_ctc_prefix_score_multi_f32(const float *x, const int *row_mem, const int *mem_len, const float *r_prev,
                                    const int *prefix_len, const int *last_char, const int *candidates,
                                    float *psi, float *r_out, int n, int C, int T, int V, int U, int blank,
                                    int eos, float logzero, void *stream);

/* ---- FLAC reader (host code; replaces torchaudio.load on LibriSpeech's .flac files, src/audio.py:102) --
 * info: STREAMINFO of the file (total_samples per channel, 0 = unknown; md5_16 = MD5 of the unencoded
 * audio, all zero = not set).  decode: interleaved samples [n, channels] as int32 (value range of the
 * file's bits_per_sample); frame header CRC-8 and frame CRC-16 are verified.  When the stream holds more
 * than capacity_samples the decode still runs to the end, *decoded_samples is the count the stream holds and
 * the call returns ASRK_EWORKSPACE (the caller retries with that capacity) - never a silent truncation. */
int asrk_flac_info(const char *path, int *sample_rate, int *channels, int *bits_per_sample,
                   int64_t *total_samples, uint8_t *md5_16);
int asrk_flac_decode_i32(const char *path, int32_t *out, int64_t capacity_samples,
                         int64_t *decoded_samples);

#ifdef __cplusplus
}
#endif
#endif /* ASRK_H */
