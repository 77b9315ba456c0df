/*
 * goalnet_hip.h — C ABI of libgoalnet_hip.so: the MI355X (gfx950) kernels behind the reference's
 * `AVM` hot path (forward, MSE, backward, Adam).
 *
 * The reference (Vasilispapg/CVML-GoalNet) has NO FFI, plugin or operator interface: its hot path
 * is a Python class that bottoms out in PyTorch's ATen CPU kernels. Each entry
 * point below therefore names the reference LINES whose arithmetic it replaces; the Python mirror of
 * the reference interface (cvml_goalnet_amd/avm.py: `AVM(audio_included)`, `model(audio, visual)`,
 * `state_dict`) is what a reference script binds, and it reaches these functions through ctypes
 * (see INTEGRATION.md).
 *
 * Conventions (fixed by SURVEY.md §8(b), last row):
 *   - every function returns int: 0 = OK, negative = argument/shape error (GOALNET_E_*),
 *     positive = hipError_t of the failed launch; goalnet_last_error() gives the text;
 *   - never throws, never allocates or frees device memory, never synchronises the stream, holds no
 *     global mutable state except the thread-local error string;
 *   - all tensors are caller-allocated fp32 device buffers passed as raw pointers + explicit dims;
 *     `stream` is a hipStream_t passed as void*;
 *   - activations are NHWC ([N][H][W][C], C contiguous); conv weights are OHWI ([Cout][3][3][Cin]);
 *     linear weights are [out][in] row-major; linear5's input dimension is in NHWC-flatten order
 *     (h*W+w)*C + c (the Python layer permutes to/from the reference's NCHW-flatten order);
 *   - 64-bit element counts wherever N*H*W*C can exceed 2^31.
 */
#ifndef GOALNET_HIP_H
#define GOALNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GOALNET_ABI_VERSION 4

#define GOALNET_OK 0
#define GOALNET_E_NULL (-1)      /* required pointer is NULL */
#define GOALNET_E_SHAPE (-2)     /* dimension out of the supported set */
#define GOALNET_E_ALIGN (-3)     /* pointer or leading dimension not 16-byte aligned */
#define GOALNET_E_WORKSPACE (-4) /* workspace too small */

/* upper bound of the per-block partial rows produced by the statistics kernels. The caller chooses nparts in
 * [1, GOALNET_STAT_PARTS] (goalnet_stat_parts: one per frame, clamped) and passes the same value to the producer and to
 * the kernel that sums the rows; for a given nparts the summation order is fixed => deterministic. */
#define GOALNET_STAT_PARTS 1024
int goalnet_stat_parts(int64_t units);

int goalnet_abi_version(void);
const char* goalnet_last_error(void);

/* ---- synthetic data (cvml_goalnet_amd/synth.py formula; bench/test inputs only) --------------- */
int goalnet_fill_uniform(float* dst, int64_t n, uint64_t seed, uint32_t tensor_id, float lo, float hi, void* stream);
/* dropout multipliers: 0 or 1/(1-p), keep iff u >= p.  Replaces nn.Dropout's Bernoulli draw,
 * /root/reference/utils.py:170, 245-254 (torch's CPU RNG stream cannot be reproduced on device). */
int goalnet_dropout_mask(float* dst, int64_t n, uint64_t seed, uint32_t tensor_id, float p, void* stream);

/* ---- layout converters (state_dict interchange) --------------------------- */
/* [B][R][C] -> [B][C][R]: OIHW<->OHWI (B=O,R=I,C=9 and back), NCHW<->NHWC, linear5 column order. */
int goalnet_transpose_inner(const float* src, float* dst, int64_t B, int64_t R, int64_t C, void* stream);
/* dgrad weights: wt[ci][2-kh][2-kw][co] = w[co][kh][kw][ci] */
int goalnet_conv3x3_weight_flip(const float* w_ohwi, float* wt, int Cout, int Cin, void* stream);
/* two tensors in one launch (conv3's and conv2's weights, flipped in every backward of the 10-frame step) */
int goalnet_conv3x3_weight_flip2(const float* wa_ohwi, float* wta, int CoutA, int CinA, const float* wb_ohwi, float* wtb, int CoutB, int CinB,
                                 void* stream);

/* ---- VisBl block 1: conv1 (3->64, k3 s3 p3) + bias + ReLU.  utils.py:151-152, 174-175 ---------- */
int goalnet_conv1_fwd(const float* x_nchw, const float* w_ohwi, const float* bias, float* y_nhwc,
                      int N, int H, int W, void* stream);
/* dW (OHWI) and dbias of conv1 from dy = grad wrt its pre-ReLU output (NHWC). autograd of utils.py:174 */
size_t goalnet_conv1_wgrad_ws_bytes(int N, int H, int W);
int goalnet_conv1_wgrad(const float* x_nchw, const float* dy_nhwc, float* dw_ohwi, float* dbias,
                        void* ws, size_t ws_bytes, int N, int H, int W, void* stream);

/* ---- MaxPool2d(3,1,0) + train-mode BatchNorm statistics.  utils.py:153-154 (and 158-159, 163-164) */
/* p = maxpool3x3s1(y); idx = argmax position 0..8 (first max in kh,kw scan order, as ATen), stored slice-major:
 * [N][C / S][Hc-2][Wc-2][S] bytes with S = 32 (S = C when C % 32 != 0) so that stores are whole cache lines;
 * partials[nparts][2][C] (double) = per-block sum and sum of squares of p. */
int goalnet_pool_bnstats_fwd(const float* y, float* p, uint8_t* idx, double* partials, int nparts,
                             int N, int Hc, int Wc, int C, void* stream);
/* The same with the pooled activation stored as bf16 (C % 32 == 0): precision = "bf16" keeps p of blocks 2 and 3 in bf16 — it is
 * read three more times per step and every GEMM behind it consumes bf16. The statistics are those of the values as stored.
 * y: fp32 (y_bf16 = 0) or bf16 (1, from goalnet_conv3x3_fwd_bf16p_o16): rounding is monotonic, so p is the same either way. */
int goalnet_pool_bnstats_fwd_p16(const void* y, int y_bf16, void* p_bf16, uint8_t* idx, double* partials, int nparts,
                                 int N, int Hc, int Wc, int C, int f16, void* stream);
/* mean/biased var -> invstd, scale = gamma*invstd, shift = beta - mean*scale; running stats updated
 * with `momentum` and the unbiased variance, as nn.BatchNorm2d does in train mode. */
int goalnet_bn_finalize(const double* partials, int nparts, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float eps, int64_t count,
                        int C, float* mean, float* invstd, float* scale, float* shift, void* stream);
/* BatchNorm backward, phase 1: per-channel sum(dz) and sum(dz * xhat) -> partials (double). */
int goalnet_bn_bwd_reduce(const float* dz, const float* p, const float* mean, const float* invstd,
                          double* partials, int nparts, int64_t npix, int C, void* stream);
/* same with dz and p each fp32 (flag 0) or bf16 (flag 1): dz as goalnet_linear_bwd_dx_bf16_o16 / goalnet_conv3x3_fwd_bf16p_o16
 * write it, p as goalnet_pool_bnstats_fwd_p16 stores it */
int goalnet_bn_bwd_reduce_t(const void* dz, int dz_bf16, const void* p, int p_bf16, const float* mean, const float* invstd,
                            double* partials, int nparts, int64_t npix, int C, int f16, void* stream);
/* phase 2: dgamma, dbeta and the three per-channel coefficients of dp = a*dz + b*p + c. */
int goalnet_bn_bwd_finalize(const double* partials, int nparts, const float* gamma, const float* mean, const float* invstd,
                            int64_t count, int C, float* dgamma, float* dbeta, float* coef3, void* stream);
/* phase 3, fused: BN backward apply -> max-pool backward (gather by argmax) -> ReLU backward. The conv output y is not
 * an operand: a window whose argmax is a given pixel has p equal to that pixel's y, so the ReLU mask (y > 0) is read off p.
 * dy[N][Hc][Wc][C] = grad wrt the conv's pre-ReLU output; dbias_partials (double
 * [nparts][C]) = per-block column sums of dy. */
int goalnet_bnpool_bwd(const float* dz, const float* p, const uint8_t* idx, const float* coef3,
                       float* dy, double* dbias_partials, int nparts, int N, int Hc, int Wc, int C, void* stream);
/* same, writing dy as bf16 into the zero-padded layout of goalnet_to_bf16_padded (dy_pad_bf16 = padded pixel 0);
 * the fp32 dy is optional (NULL when only the bf16 GEMMs consume it). */
int goalnet_bnpool_bwd_bf16p(const float* dz, const float* p, const uint8_t* idx, const float* coef3,
                             float* dy, void* dy_pad_bf16, double* dbias_partials, int nparts, int N, int Hc, int Wc, int C,
                             int f16, void* stream);
/* same with dz and p each fp32 (flag 0) or bf16 (flag 1); either output may be NULL (not both) */
int goalnet_bnpool_bwd_bf16p_t(const void* dz, int dz_bf16, const void* p, int p_bf16, const uint8_t* idx, const float* coef3,
                               float* dy, void* dy_pad_bf16, double* dbias_partials, int nparts, int N, int Hc, int Wc, int C,
                               int f16, void* stream);
/* out[c] = sum over parts of partials[part][c] (row stride `stride` doubles), cast to float */
int goalnet_partials_sum(const double* partials, int nparts, int64_t stride, int C, float* out, void* stream);
/* Small shapes (the reference's 10-frame sub-batches, main.py:177-196; <= 2^20 elements, C <= 512, fp32): the passes above
 * with the finalise step folded in. <= 32 blocks of 1024 threads, one partial row each; the last block to arrive sums the rows
 * of every column (one batch of loads, row order) and writes what goalnet_bn_finalize / goalnet_bn_bwd_finalize /
 * goalnet_partials_sum would. forward = pool + statistics + finalise in ONE launch; backward = reduce + finalise in one launch
 * (goalnet_bn_bwd_reduce_small), then either goalnet_bnpool_bwd_small (max-pool / ReLU backward + bias gradient in one launch, a
 * direct 9-window gather) or the rolling-row goalnet_bnpool_bwd + goalnet_partials_sum (faster at 40 x 40: DESIGN.md §4.3).
 * ws: goalnet_bn_small_ws_bytes(C); ctr: one int32, zero on entry, zero again on exit. */
size_t goalnet_bn_small_ws_bytes(int C);
int goalnet_pool_bn_fwd_small(const float* y, float* p, uint8_t* idx, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps,
                              float* mean, float* invstd, float* scale, float* shift,
                              void* ws, size_t ws_bytes, int* ctr, int N, int Hc, int Wc, int C, void* stream);
int goalnet_bn_bwd_reduce_small(const float* dz, const float* p, const float* mean, const float* invstd, const float* gamma,
                                float* dgamma, float* dbeta, float* coef3, void* ws, size_t ws_bytes, int* ctr,
                                int N, int Hc, int Wc, int C, void* stream);
int goalnet_bnpool_bwd_small(const float* dz, const float* p, const uint8_t* idx, const float* coef3, float* dy, float* dbias,
                             void* ws, size_t ws_bytes, int* ctr, int N, int Hc, int Wc, int C, void* stream);
/* same in double: the one-row form of a partials array that a rank contributes to the cross-rank BatchNorm sums
 * (ddp.SyncStats; SURVEY.md §8(e) "SyncBN": all-reduce of per-channel sum(x), sum(x^2)) */
int goalnet_partials_sum_f64(const double* partials, int nparts, int64_t stride, int C, double* out, void* stream);
/* goalnet_partials_sum of two arrays in one launch */
int goalnet_partials_sum2(const double* pa, int na, int Ca, float* oa, const double* pb, int nb, int Cb, float* ob, void* stream);

/* ---- VisBl blocks 2,3: conv 3x3 s1 p1 as implicit GEMM on fp32 MFMA.  utils.py:156-157, 161-162 --- */
/* y = [relu](conv(bnapply(x), w) + bias).  scale/shift (per input channel) may be NULL (no BN on load);
 * zero padding is applied AFTER the affine, as the reference pads the BatchNorm output. bias may be
 * NULL. Also computes the data gradient when called with goalnet_conv3x3_weight_flip'ed weights. */
/* Few pixels (the reference's 10-frame sub-batches, main.py:177-184) give too few output tiles for 256 CUs: with a
 * workspace of goalnet_conv3x3_fwd_ws_bytes (0 = not needed) K is split into slabs that are summed in a fixed order.
 * ws may be NULL (no split; same result up to fp32 summation order). */
size_t goalnet_conv3x3_fwd_ws_bytes(int N, int H, int W, int Cin, int Cout);
/* tile_ctr (nullable) / n_ctr: int32 ticket counters in device memory, ZERO on entry, one per 128 x 128 output tile; when
 * given (and K is split) the last block of each tile sums the tile's slabs in split order and applies the epilogue, so the
 * reduction needs no second launch (same bits as the two-launch form). The library leaves the counters zero again; a
 * caller may lend the same counters to later calls on the same stream, not to calls that may run concurrently. */
int goalnet_conv3x3_fwd(const float* x, const float* scale, const float* shift, const float* w_ohwi,
                        const float* bias, int relu, float* y, int N, int H, int W, int Cin, int Cout,
                        void* ws, size_t ws_bytes, int* tile_ctr, int n_ctr, void* stream);
/* dw[Cout][3][3][Cin] = sum_m dy[m][co] * bnapply(x)[m + tap][ci]; split over m, deterministic.
 * codes (nullable): the per-pixel border-code table of goalnet_conv3x3_wgrad_codes for the same N, H, W (it depends on
 * nothing else, so a caller that keeps it saves a launch per call; NULL: built inside ws on every call).
 * tile_ctr / n_ctr: as for goalnet_conv3x3_fwd (used where no border correction is pending: few frames, or no affine). */
size_t goalnet_conv3x3_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout);
size_t goalnet_conv3x3_wgrad_codes_bytes(int N, int H, int W);
int goalnet_conv3x3_wgrad_codes(uint32_t* codes, int N, int H, int W, void* stream);
int goalnet_conv3x3_wgrad(const float* x, const float* scale, const float* shift, const float* dy, float* dw,
                          void* ws, size_t ws_bytes, const uint32_t* codes, int* tile_ctr, int n_ctr,
                          int N, int H, int W, int Cin, int Cout, void* stream);

/* ---- precision = "bf16" / "fp16" modes: the same contractions on the 16-bit matrix cores (v_mfma_f32_32x32x16_bf16 /
 * v_mfma_f32_32x32x16_f16, fp32 accumulate). Operands are 16-bit copies produced by the cast passes below; everything
 * else stays fp32. The reference is fp32 (utils.py:37-47): these modes are extensions with their own tolerance
 * (logits <= 1e-3). Every entry point that reads or writes 16-bit data takes `f16`: 0 = bfloat16, 1 = IEEE binary16 —
 * both are 16 bits per element, so layouts, alignment rules and workspace sizes are the same ("bf16" in names and
 * parameter names below stands for "the 16-bit format selected by f16"). ---- */
/* y_bf16[i] = bf16(x[i]); n % 8 == 0 */
int goalnet_cast_bf16(const float* x, void* y_bf16, int64_t n, int f16, void* stream);
/* y[i] = float(x_bf16[i]) (exact); n % 8 == 0 */
int goalnet_cast_f32(const void* x_bf16, float* y, int64_t n, int f16, void* stream);
/* y_bf16 = bf16(x * scale[c] + shift[c]), c = i mod C: the BatchNorm output utils.py:177/182/187, materialised in bf16 */
int goalnet_bn_apply_bf16(const float* x, const float* scale, const float* shift, void* y_bf16, int64_t n, int C, int f16, void* stream);
int goalnet_bn_apply_bf16_p16(const void* x_bf16, const float* scale, const float* shift, void* y_bf16, int64_t n, int C, int f16, void* stream);   /* bf16 input */
/* y = [relu](conv3x3(x_bf16 NHWC, w_bf16 OHWI) + bias), fp32 out; Cin % 64 == 0. Data gradient with flipped weights. */
int goalnet_conv3x3_fwd_bf16(const void* x_bf16, const void* w_bf16, const float* bias, int relu, float* y,
                             int N, int H, int W, int Cin, int Cout, int f16, void* stream);
/* goalnet_linear_fwd on bf16 operands (x_bf16 [M][K] with leading dim ldx elements, w_bf16 [J][K]); K % 64 == 0 */
size_t goalnet_linear_fwd_bf16_ws_bytes(int M, int64_t K, int J);
int goalnet_linear_fwd_bf16(const void* x_bf16, int64_t ldx, const void* w_bf16, const float* bias, int relu,
                            const float* dropmask, int64_t ldmask, float* y, int64_t ldy, float* mult_out, int64_t ldmult,
                            int M, int64_t K, int J, void* ws, size_t ws_bytes, int f16, void* stream);

/* zero-padded bf16 activations [N][H+2][W+2][C]: with a zero border the 3x3 taps are constant pixel shifts (no masks),
 * which is what lets the weight gradient run as a plain GEMM over the padded pixel grid. The buffer has W+3 zero
 * guard pixels in front of padded pixel 0 and W+3+64 behind; `*_pad` arguments are the address of padded pixel 0 and
 * the caller zeroes the whole buffer once (only interior pixels are ever written). */
int goalnet_bf16_padded_layout(int N, int H, int W, int C, int64_t* total_elems, int64_t* offset_elems);
int goalnet_to_bf16_padded(const float* x, const float* scale, const float* shift, void* y_pad, int N, int H, int W, int C, int f16, void* stream);
int goalnet_to_bf16_padded_p16(const void* x_bf16, const float* scale, const float* shift, void* y_pad, int N, int H, int W, int C, int f16, void* stream);   /* bf16 input */
size_t goalnet_conv3x3_fwd_bf16p_ws_bytes(int N, int H, int W, int Cin, int Cout);   /* split-K slabs, as goalnet_conv3x3_fwd */
int goalnet_conv3x3_fwd_bf16p(const void* x_pad, const void* w_bf16, const float* bias, int relu, float* y,
                              int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, int f16, void* stream);
/* The same convolution with its result stored as bf16 [N][H][W][Cout] (bias, relu as above; both off = the data-gradient use
 * with w = flipped weights). Its consumers are HBM-bound passes only: the max-pool of the forward
 * (goalnet_pool_bnstats_fwd_p16) and the BatchNorm backward (goalnet_bn_bwd_reduce_t, goalnet_bnpool_bwd_bf16p_t).
 * fp32 accumulation, one rounding at the store. Served by the
 * 256 x 256 tile only: goalnet_conv3x3_fwd_bf16p_o16_ok() says whether the dims are; otherwise use the fp32-output form. */
int goalnet_conv3x3_fwd_bf16p_o16_ok(int N, int H, int W, int Cin, int Cout);
int goalnet_conv3x3_fwd_bf16p_o16(const void* x_pad, const void* w_bf16, const float* bias, int relu, void* y_bf16,
                                  int N, int H, int W, int Cin, int Cout, int f16, void* stream);
size_t goalnet_conv3x3_wgrad_bf16_ws_bytes(int N, int H, int W, int Cin, int Cout);
int goalnet_conv3x3_wgrad_bf16(const void* x_pad, const void* dy_pad, float* dw, void* ws, size_t ws_bytes,
                               int N, int H, int W, int Cin, int Cout, int f16, void* stream);
/* ---- precision = "bf16x6" / "fp16x3" (csrc/split3.hip): the large GEMMs of VisBl (/root/reference/utils.py:156-170, 179-193; their
 * backward, /root/reference/main.py:192) with fp32-grade products on the 16-bit MFMA. `parts` = 3: an fp32 operand value is stored as
 * three bf16 values hi + mid + lo (exact), six partial products per product; `parts` = 2: as two fp16 values hi + mid of the value
 * scaled by a power of two (22 significand bits; the scale puts the tensor's largest magnitude, goalnet_absmax, into [2^14, 2^15)),
 * three partial products per product, and the epilogues undo the scales (`oscale` = one exponent from goalnet_split_scales; NULL for parts = 3).
 * The parts lie side by side along the channel / row axis; a GEMM is ONE launch of the 16-bit kernel with the partial products as
 * K-segments, fp32 accumulation, fp32 result.
 *   absmax        : atomic max of the bit pattern of |x[r][c] * scale[c % bnC] + shift[c % bnC]| into *amax_bits (zeroed by the caller)
 *   split_scales  : *oscale = -(k_a + k_b), the exponent the GEMM epilogue adds (ldexpf) to undo the operands' scales 2^k_a, 2^k_b
 *   split_padded  : x fp32 [N][H][W][C] (optional per-channel affine = the BatchNorm, applied in fp32) -> zero-padded
 *                   [N][H+2][W+2][parts C] 16-bit, interior only (borders / guards zeroed once by the caller, layout of to_bf16_padded)
 *   split_rows    : x fp32 [rows][C] (row stride ldx; optional affine per column c with channel c % bnC) -> [rows][parts C] 16-bit
 *                   (weights [Cout][9][Cin] -> [Cout][9][parts Cin]; linear5's operands [M][K] -> [M][parts K])
 *   conv3x3_fwd_split: y fp32 = act(conv(x, w) + bias); the data gradient = the same call on the split gradient and flipped weights
 *   conv3x3_wgrad_split: dw fp32 [Cout][3][3][Cin] from split x and split dy (padded layouts)
 *   linear_*_split: linear5's forward (epilogue fields as goalnet_linear_fwd_bf16), dX and dW on xs [M][parts K], ws [J][parts K],
 *                   dys [M][parts J]; goalnet_linear_split_ok says whether the dims are served (256 x 256 tile only) */
int goalnet_absmax(const float* x, int64_t ldx, const float* scale, const float* shift, int bnC, int64_t rows, int64_t C,
                   unsigned* amax_bits, void* stream);
int goalnet_split_scales(const unsigned* amax_a, const unsigned* amax_b, int* oscale, void* stream);
int goalnet_split_padded(int parts, const float* x, const float* scale, const float* shift, const unsigned* amax_bits, void* y_pads,
                         int N, int H, int W, int C, void* stream);
int goalnet_split_rows(int parts, const float* x, int64_t ldx, const float* scale, const float* shift, int bnC, const unsigned* amax_bits,
                       void* ys, int64_t rows, int64_t C, void* stream);
int goalnet_conv3x3_fwd_split(int parts, const void* x_pads, const void* ws, const float* bias, int relu, float* y,
                              int N, int H, int W, int Cin, int Cout, const int* oscale, void* stream);
size_t goalnet_conv3x3_wgrad_split_ws_bytes(int parts, int N, int H, int W, int Cin, int Cout);
int goalnet_conv3x3_wgrad_split(int parts, const void* x_pads, const void* dy_pads, float* dw, void* ws, size_t ws_bytes,
                                int N, int H, int W, int Cin, int Cout, const int* oscale, void* stream);
int goalnet_linear_split_ok(int parts, int M, int64_t K, int J);
size_t goalnet_linear_fwd_split_ws_bytes(int parts, int M, int64_t K, int J);
int goalnet_linear_fwd_split(int parts, const void* xs, const void* ws_parts, const float* bias, int relu, const float* dropmask, int64_t ldmask,
                             float* y, int64_t ldy, float* mult_out, int64_t ldmult, int M, int64_t K, int J, void* ws, size_t ws_bytes,
                             const int* oscale, void* stream);
int goalnet_linear_bwd_dx_split(int parts, const void* dys, const void* ws_parts, float* dx, int64_t lddx, int M, int64_t K, int J,
                                const int* oscale, void* stream);
int goalnet_linear_bwd_dw_split(int parts, const void* dys, const void* xs, float* dw, int M, int64_t K, int J, const int* oscale, void* stream);
int goalnet_linear_bwd_dx_bf16(const void* dy_bf16, int64_t lddy, const void* w_bf16, const float* mult, int64_t ldmult,
                               float* dx, int64_t lddx, int M, int64_t K, int J, int f16, void* stream);
/* dx as bf16 (no mult), same contract as goalnet_conv3x3_fwd_bf16p_o16 */
int goalnet_linear_bwd_dx_bf16_o16_ok(int M, int64_t K, int J);
int goalnet_linear_bwd_dx_bf16_o16(const void* dy_bf16, int64_t lddy, const void* w_bf16, void* dx_bf16, int64_t lddx,
                                   int M, int64_t K, int J, int f16, void* stream);
int goalnet_linear_bwd_dw_bf16(const void* dy_bf16, int64_t lddy, const void* x_bf16, int64_t ldx, float* dw,
                               int M, int64_t K, int J, int f16, void* stream);

/* ---- Linear layers (linear5, audbl.linear3, fusion.0/3/6/9).  utils.py:168-170, 211, 243-253 ---- */
/* y[m][j] = act(sum_k xa[m][k] * w[j][k] + bias[j]) * dropmask[m][j]
 *   xa = x*scale[k % bnC] + shift[k % bnC] when scale != NULL (BatchNorm folded into the load);
 *   relu != 0 applies ReLU; dropmask may be NULL; mult_out (nullable) receives
 *   (pre-activation > 0 ? 1 : 0) * dropmask, the multiplier the backward pass needs.
 * Split-K with a deterministic slab reduction is used when the grid would not fill the chip. */
size_t goalnet_linear_fwd_ws_bytes(int M, int64_t K, int J);
int goalnet_linear_fwd(const float* x, int64_t ldx, const float* scale, const float* shift, int bnC,
                       const float* w, const float* bias, int relu, const float* dropmask, int64_t ldmask,
                       float* y, int64_t ldy, float* mult_out, int64_t ldmult,
                       int M, int64_t K, int J, void* ws, size_t ws_bytes, void* stream);
/* dx[m][k] = (sum_j dy[m][j] * w[j][k]) * mult[m][k]   (mult nullable) */
int goalnet_linear_bwd_dx(const float* dy, int64_t lddy, const float* w, const float* mult, int64_t ldmult,
                          float* dx, int64_t lddx, int M, int64_t K, int J, void* stream);
/* dw[j][k] = sum_m dy[m][j] * xa[m][k]   (xa as in goalnet_linear_fwd); db[j] = sum_m dy[m][j] when db != NULL */
int goalnet_linear_bwd_dw(const float* dy, int64_t lddy, const float* x, int64_t ldx,
                          const float* scale, const float* shift, int bnC,
                          float* dw, float* db, int M, int64_t K, int J, void* stream);
/* out[j] = sum_m x[m][j] (deterministic) — bias gradients */
int goalnet_colsum(const float* x, int64_t ldx, int M, int J, float* out, void* stream);
/* y = x * mult (elementwise over [M][J] with leading dims) */
int goalnet_mul(const float* x, int64_t ldx, const float* mult, int64_t ldmult, float* y, int64_t ldy,
                int M, int J, void* stream);

/* ---- AudBl convolutions.  utils.py:203-207, 216-220 -------------------------------------------- */
/* y[n][co][lo] = [relu](b[co] + sum_{ci,k} x[n][ci][stride*lo - pad + k] * w[co][ci][k]), k = 3 */
int goalnet_conv1d_fwd(const float* x, const float* w, const float* b, int relu, float* y,
                       int N, int Cin, int L, int Cout, int stride, int pad, void* stream);
/* dz = grad wrt pre-activation. dx nullable (first layer needs none). ws (nullable, goalnet_conv1d_bwd_ws_bytes; 0 for few
 * frames): fp64 partial sums of the weight gradient over frame slices, added in slice order. */
size_t goalnet_conv1d_bwd_ws_bytes(int N, int Cin, int Cout);
int goalnet_conv1d_bwd(const float* x, const float* dz, const float* w, float* dx, float* dw, float* db,
                       int N, int Cin, int L, int Cout, int stride, int pad, void* ws, size_t ws_bytes, void* stream);
/* the same for few frames (N < 64: the reference's sub-batches) in ONE launch, with the layer's own ReLU backward folded in:
 * y (nullable) = the layer's ReLU output, the effective output gradient is dz * (y > 0); dx nullable; Cin <= 256 */
int goalnet_conv1d_bwd_small(const float* x, const float* dz, const float* y, const float* w, float* dx, float* dw, float* db,
                             int N, int Cin, int L, int Cout, int stride, int pad, void* stream);
/* dz = dy * (y > 0) */
int goalnet_relu_bwd(const float* dy, const float* y, float* dz, int64_t n, void* stream);

/* ---- the whole fusion MLP + head, and its backward, at <= 16 rows (the reference's sub-batches, main.py:177-184): ONE launch
 * per direction (csrc/mlp.hip: 64 resident blocks walk the layers and meet at a bounded grid barrier between them).
 * Pointer arrays are HOST arrays of device pointers, read at call time: w / b / dw / db [5] = fusion.0, .3, .6, .9, .12;
 * mask / h / mult [4] = the dropout multipliers (nullable), outputs (n, 512 | 512 | 256 | 128, contiguous) and saved
 * (pre-activation > 0) * mask (nullable) of fusion.0, .3, .6, .9. cat (n, K0) has row stride ldcat; K0 = 640 (audio) or 512.
 * labels (nullable): the same launch then also evaluates goalnet_mse_bcast(out, labels) -> loss[0], dout[n] (same code, same bits).
 * sync: int32[3] in device memory, zero on entry, zero again on exit ([2] is set, and stays set, if the barrier timed out).
 * Backward: x [5] = cat, h1..h4; m [5] = the saved multipliers of those (mcat with row stride ldmcat, then mult[0..3]);
 * writes dw / db of the five layers, dcat (n, K0) = the gradient wrt the pre-activations behind `cat`, and (db5 nullable)
 * the column sums of dcat[:, voff:] = visbl.linear5.bias's gradient. ws: goalnet_mlp_bwd_ws_bytes(n). */
int goalnet_mlp_blocks(void);
int goalnet_mlp_fwd(const float* cat, int64_t ldcat, int K0, const float* const* w, const float* const* b,
                    const float* const* mask, const int64_t* ldmask, float* const* h, float* const* mult,
                    float* logit, float* out, const float* labels, float* loss, float* dout, int n, int* sync, void* stream);
size_t goalnet_mlp_bwd_ws_bytes(int n);
int goalnet_mlp_bwd(const float* dout, const float* out, const float* const* x, int64_t ldcat, const float* const* m, int64_t ldmcat,
                    const float* const* w, float* const* dw, float* const* db, float* dcat, int64_t lddcat, float* db5, int voff,
                    int n, int K0, void* ws, size_t ws_bytes, int* sync, void* stream);

/* ---- head: fusion.12 (128 -> 1), Sigmoid, 4*y + 1.  utils.py:255-256, 270 ----------------------- */
int goalnet_head_fwd(const float* h, int64_t ldh, const float* w, const float* b, float* logit, float* out,
                     int N, int K, void* stream);
/* dout[N] -> dh = dlogit * w * mult (nullable), dw, db.  dlogit = dout * (out-1) * (5-out) / 4 */
int goalnet_head_bwd(const float* dout, const float* out, const float* h, int64_t ldh, const float* w,
                     const float* mult, int64_t ldmult, float* dh, int64_t lddh, float* dw, float* db,
                     int N, int K, void* stream);

/* ---- loss: nn.MSELoss()(pred(n,1), labels(n,)) with its (n,n) broadcast.  main.py:68, 191 -------- */
/* loss = 1/n^2 sum_i sum_j (p_i - y_j)^2 ; dpred_i = 2/n * (p_i - mean(y)) (nullable) */
int goalnet_mse_bcast(const float* pred, const float* labels, int N, float* loss, float* dpred, void* stream);

/* ---- optimizer: torch.optim.Adam defaults over a flat arena.  main.py:70, 193 ------------------- */
/* g is scaled by grad_scale first (1/world_size after a SUM all-reduce). `step` is 1-based. */
/* ---- audio pre-processing (/root/reference/utils.py:313-349 from the decoded waveform on) -------------------
 * y[rows][B] = (float)(R[B][T] . x[rows][T]) in double: the per-row cubic resample of utils.py:337-343 as a matrix (R = the
 * not-a-knot spline of scipy's interp1d(kind='cubic') evaluated at linspace(0, T-1, B); built on the host). T >= 4. */
int goalnet_cubic_resample(const float* x, const double* R, float* y, int64_t rows, int T, int B, void* stream);
/* log-mel power spectrogram in dB (librosa.feature.melspectrogram + power_to_db before its top_db clip, utils.py:333) of
 * n_slots segments of one waveform: slot s = samples [start[s], start[s] + len[s]); STFT frame t < 1 + len[s] / 512 is the
 * Hann-windowed 2048 samples centred at t * 512 (zeros outside the segment). logmel: [n_slots][Tmax][128] doubles; window
 * [2048], twiddle [1024][2] = (cos, -sin)(2 pi k / 2048); mel band m = sum of mel_count[m] bins from mel_start[m] with
 * weights mel_w[mel_off[m] ...]. */
int goalnet_logmel_slots(const float* y, const int64_t* start, const int* len, int n_slots, int Tmax, const double* window,
                         const double* twiddle, const int* mel_start, const int* mel_count, const int* mel_off, const double* mel_w,
                         double* logmel, void* stream);
/* per slot: clip at (slot max - top_db), DCT with dct[n_mfcc][128], resample the T = 1 + len/512 columns to B with the
 * (B, T) matrix at r_all + r_off[slot]: out [n_slots][n_mfcc][B] floats (the (N, 30, B) tensor AudBl takes) */
int goalnet_mfcc_from_logmel(const double* logmel, const int* len, int n_slots, int Tmax, const double* dct, const double* r_all,
                             const int64_t* r_off, float* out, int n_mfcc, int B, double top_db, void* stream);

/* ---- classifier-head variant (EXTENSION; comment-only in the reference: /root/reference/utils.py:257 nn.Softmax(dim = 1),
 * main.py:69 nn.CrossEntropyLoss(), main.py:96, 189 `(labels - 1).long()`, main.py:97, 190 `argmax + 1`) -----------------------
 * scores (N, C) = 4 softmax(h w^T + b) + 1 (utils.py:270 applies to either head); C <= 8, K < 1024. */
int goalnet_cls_head_fwd(const float* h, int64_t ldh, const float* w, const float* b, float* logits /* (N,C) or NULL */,
                         float* scores, int N, int K, int C, void* stream);
/* loss[0] = mean_n (logsumexp(scores_n) - scores_n[labels_n - 1]) (labels: float classes 1..C); dscores (N, C) or NULL */
int goalnet_cross_entropy(const float* scores, const float* labels, float* loss /* or NULL */, float* dscores, int N, int C, void* stream);
/* backward of goalnet_cls_head_fwd: dh = (dz w) * mult, dw[C][K], db[C] with dz = p (.) (4 dscores - <4 dscores, p>), p = (scores-1)/4 */
int goalnet_cls_head_bwd(const float* dscores, const float* scores, const float* h, int64_t ldh, const float* w, const float* mult,
                         int64_t ldmult, float* dh, int64_t lddh, float* dw, float* db, int N, int K, int C, void* stream);
/* classes[n] = (float)(argmax_c scores[n][c] + 1), first maximum on ties (torch.argmax) */
int goalnet_argmax_plus1(const float* scores, float* classes, int N, int C, void* stream);

/* ---- dispatch introspection (no launch): the kernel template goalnet_conv3x3_fwd / goalnet_conv3x3_fwd_bf16p(_o16) select for
 * these dims, as a compiler-spelled string ("... [AL = ..., BL = ...]"). bench.py labels its roofline object with it. */
const char* goalnet_conv3x3_fwd_kernel_name(int N, int H, int W, int Cin, int Cout, int affine);
const char* goalnet_conv3x3_fwd_bf16p_kernel_name(int N, int H, int W, int Cin, int Cout, int forward);

int goalnet_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                      double beta2, double eps, int step, float grad_scale, void* stream);

/* ---- device-resident step state: what a captured HIP graph cannot take as host scalars -------------
 * The reference's per-video loop (/root/reference/main.py:169-198) runs one optimizer step per <=10 frames; captured
 * as one graph per sub-batch, the Adam step count, the dropout draw index and the frame cursor must be read from
 * device memory. All counters are int64 in device memory. */
int goalnet_counter_add(int64_t* counter, int64_t delta, void* stream);
/* counters[i] += d_i for four consecutive counters (adam step, dropout draw, frame cursor, sub-batch index): one launch */
int goalnet_counters_add4(int64_t* counters, int64_t d0, int64_t d1, int64_t d2, int64_t d3, void* stream);
/* `layers` masks back to back in dst: mask l is (n, widths[l]) row-major, drawn from stream
 * tid_base + tid_stride * (*step) + l — the same bits as goalnet_dropout_mask with that tensor_id. widths: host array.
 * row_offset >= 0: mask l holds rows [row_offset, row_offset + n) of the (row_offset + n, widths[l]) mask of that stream, so
 * that data-parallel ranks draw disjoint rows of the mask one process would draw for the concatenated batch. */
int goalnet_dropout_masks_dev(float* dst, int n, const int* widths, int layers, uint64_t seed, uint32_t tid_base,
                              uint32_t tid_stride, const int64_t* step, float p, int64_t row_offset, void* stream);
/* goalnet_adam_step with the 1-based step count = *step + step_bias (bias 1: the counter holds the completed steps) */
int goalnet_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                          double beta2, double eps, const int64_t* step, int64_t step_bias, float grad_scale, void* stream);
/* the same on at most max_blocks blocks of 256 threads: a background pass that leaves CUs and HBM bandwidth to the kernels it
 * runs beside (the 10-frame step updates linear5.weight on its own stream under the rest of backward) */
int goalnet_adam_step_dev_blocks(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                                 double eps, const int64_t* step, int64_t step_bias, float grad_scale, int max_blocks, void* stream);
/* the same, additionally writing bf16(p_new) for the slice [shadow_begin, shadow_begin + shadow_count) of the arena (both
 * multiples of 4): the next step's bf16 GEMM operand (visbl.linear5.weight) without a separate 7.7 GB cast pass */
int goalnet_adam_step_dev_shadow(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                                 double beta2, double eps, const int64_t* step, int64_t step_bias, float grad_scale,
                                 void* shadow_bf16, int64_t shadow_begin, int64_t shadow_count, int f16, void* stream);
/* precision = "fp16": gradients are computed from a loss-scaled dL/dpred (goalnet_scale) so that the 16-bit activation
 * gradients stay inside binary16's range; grad_scale of the Adam entry points carries 1 / loss_scale. Overflow guard:
 * goalnet_grad_finite_check stamps *bad_step = *step + step_bias when g[0..n) holds an inf / nan (any non-finite value of an
 * earlier layer's gradient reaches the last-computed conv1 gradients: checking the tail bucket is enough) and counts it in
 * *skipped; goalnet_adam_step_dev_guarded (goalnet_adam_step_dev[_shadow] + that stamp) then leaves p, m, v untouched. */
int goalnet_scale(float* x, int64_t n, float s, void* stream);
int goalnet_grad_finite_check(const float* g, int64_t n, const int64_t* step, int64_t step_bias, int64_t* bad_step, int64_t* skipped,
                              void* stream);
int goalnet_adam_step_dev_guarded(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                                  double eps, const int64_t* step, int64_t step_bias, float grad_scale, void* shadow_16 /* nullable */,
                                  int64_t shadow_begin, int64_t shadow_count, int f16, const int64_t* bad_step, void* stream);
/* goalnet_counters_add4 for a guarded step: when *bad_step == counters[0] + d0 (this step was stamped and its Adam skipped) the
 * step count does not advance — a skipped step is not counted, as with torch's GradScaler — and the stamp is cleared so that the
 * retry under the same count is judged on its own gradients; counters 1..3 advance as usual. */
int goalnet_counters_add4_guarded(int64_t* counters, int64_t d0, int64_t d1, int64_t d2, int64_t d3, int64_t* bad_step, void* stream);
/* block[0:nrows] = table[*cursor : *cursor + nrows]  (batch_frames[a:b], main.py:181-184); row_bytes % 4 == 0 */
int goalnet_rows_gather(const void* table, void* block, int64_t row_bytes, int nrows, const int64_t* cursor, void* stream);
/* table[*cursor : *cursor + nrows] = block[0:nrows]  (predictions.extend(...), losses.append(...), main.py:195-196) */
int goalnet_rows_scatter(const void* block, void* table, int64_t row_bytes, int nrows, const int64_t* cursor, void* stream);
/* ---- after the hot path, once per video: summary selection and F-score.  utils.py:586-643 -------------------
 * Integer work, bit-exact with the reference's Python (DESIGN.md "post-processing"). All pointers are device memory. */
/* 0/1 knapsack of utils.py:465-510 on already scaled integer weights (int(w * scale_factor)) and capacity
 * (int(capacity * scale_factor)); selected[i] = 1 for the items the reference's back-tracking returns. */
size_t goalnet_knapsack_ws_bytes(int n_items, int capacity_scaled);
int goalnet_knapsack(const int64_t* values, const int32_t* weights_scaled, int n_items, int capacity_scaled,
                     int32_t* selected, void* ws, size_t ws_bytes, void* stream);
/* get_fscore, utils.py:552-580: gd [n_users][full_n_frames] and mask [full_n_frames] hold 0/1; fscore[0] = mean,
 * fscore[1] = max over the users; counts: scratch of 2 * (n_users + 1) int64 (exact integer sums) */
int goalnet_fscore(const uint8_t* gd, const uint8_t* mask, int n_users, int full_n_frames, double* fscore, int64_t* counts,
                   void* stream);
/* postprocess (utils.py:606-643) [+ get_fscore when gd != NULL]: pred = the model's N_sampled outputs; importances =
 * int8(round half even), expanded x skip_frames to full_n_frames (expand_array, utils.py:396-410); per clip
 * value = sum over [a:b), length = len([a:b)) (utils.py:445-463), weight = length * weight_scale; knapsack with
 * capacity_scaled = int(int(0.15 * full_n_frames) * 5) computed by the caller (utils.py:633, 478); mask[a..b] = 1 for
 * the selected clips (end inclusive, utils.py:637-641). status[0] != 0: a selected interval leaves the video (the
 * reference raises IndexError there). Outputs: mask [full_n_frames], selected [n_clips] flags, clip_values [n_clips],
 * clip_lengths [n_clips], fscore [2], status [1]. */
size_t goalnet_postprocess_ws_bytes(int n_clips, int capacity_scaled, int n_users);
int goalnet_postprocess(const float* pred, int n_sampled, int skip_frames, int full_n_frames, const int32_t* change_points,
                        int n_clips, int weight_scale, int capacity_scaled, const uint8_t* gd, int n_users, uint8_t* mask,
                        int32_t* selected, int64_t* clip_values, int32_t* clip_lengths, double* fscore, int32_t* status,
                        void* ws, size_t ws_bytes, void* stream);

/* ---- before the hot path: frame pre-processing of the loader.  utils.py:274-292 (decode excluded) -----------------
 * frames_hwc: device uint8 [N][H0][W0][3] (BGR as cv2 decodes); out_nchw: float32 [N][3][H][W] = per-frame min-max
 * normalisation (float64, + 1e-7) -> float32 -> bilinear resize (cv2.resize INTER_LINEAR semantics) -> channel-first.
 * minmax: scratch int32 [N][2]. PARITY UNPINNED (OpenCV absent from the build image). */
int goalnet_frames_preprocess(const uint8_t* frames_hwc, int N, int H0, int W0, float* out_nchw, int H, int W,
                              int32_t* minmax, void* stream);

/* up to GOALNET_ROWCOPY_MAX gathers (gather != 0: dst[0:nrows] = src[c : c + nrows]) and scatters
 * (gather == 0: dst[c : c + nrows] = src[0:nrows]) in one launch; c = *cursor + cursor_bias */
#define GOALNET_ROWCOPY_MAX 4
typedef struct {
    const void* src; void* dst; int64_t row_bytes; int nrows; int gather; const int64_t* cursor; int64_t cursor_bias;
} goalnet_rowcopy;
int goalnet_rows_copy_batch(const goalnet_rowcopy* segs, int count, void* stream);
/* the last launch of a sub-batch step: goalnet_rows_copy_batch (scatter / gather segments, small: one block copies them) and THEN
 * goalnet_counters_add4[_guarded] (bad_step nullable) — the cursors are read before any counter moves. main.py:195-198 */
int goalnet_rows_scatter_tick(const goalnet_rowcopy* segs, int count, int64_t* counters, int64_t d0, int64_t d1, int64_t d2, int64_t d3,
                              int64_t* bad_step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GOALNET_HIP_H */
