/*
 * libmi355conv — C ABI of the MI355X (gfx950) conv-net hot path.
 *
 * The reference (bababyVN/medical-image-segmentation-and-classification) is pure
 * Python/PyTorch and has no FFI of its own: its boundary for this path is the
 * torch.nn leaf-op API called from models/ and utils/helpers.py::train.  Each
 * entry point below therefore cites the reference call site whose ATen/cuDNN
 * kernel it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - all tensors are device pointers owned by the caller (PyTorch's caching
 *     allocator); the library never allocates, frees or synchronises;
 *   - activations are NHWC ("pixel rows"): element (n,h,w,c) of a tensor with
 *     channel stride `ld` lives at ((n*H+h)*W+w)*ld + c, so a channel slice of a
 *     wider (concatenated) tensor is addressed with the same pointer arithmetic;
 *   - `dtype` selects the storage/MFMA input type of activations and packed
 *     weights (MI355_F32: v_mfma_f32_32x32x2_f32, MI355_BF16 / MI355_F16:
 *     v_mfma_f32_{32x32x16,16x16x32}_{bf16,f16}); accumulation, statistics,
 *     parameters and parameter gradients are always fp32;
 *   - every launcher takes the hipStream_t to enqueue on and returns 0 or a
 *     negative MI355_ERR_* / positive hipError_t; mi355_last_error() gives text.
 */
#ifndef MI355CONV_H_
#define MI355CONV_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* mi355_stream_t; /* == hipStream_t */

enum { MI355_F32 = 0, MI355_BF16 = 1, MI355_F16 = 2 };
enum { MI355_OK = 0, MI355_ERR_ARG = -1, MI355_ERR_UNSUPPORTED = -2, MI355_ERR_RUNTIME = -3 };

int mi355_version(void);
const char* mi355_last_error(void);

/* ---- layout staging ------------------------------------------------------------------ */

/* NCHW fp32 network input -> NHWC `dtype` with channels zero-padded to Cpad (multiple of
 * 32).  Replaces the implicit layout/precision change of `x.to(device)` + autocast,
 * utils/helpers.py:318-322. */
int mi355_pack_input_nchw(const float* x, void* y, int N, int C, int H, int W, int Cpad, int dtype,
                          mi355_stream_t s);
/* The same input as the 3 x 3 patches of a stride-1, pad-1 convolution: y[n][h][w][c * 9 + kh * 3 + kw] = x[n][c][h + kh - 1][w + kw - 1]
 * (zero outside the image), 32 channels per pixel, C <= 3.  The stem Conv2d(3, Co, 3, 1, 1) (AttentionUNet.py:6,60) is then a
 * pointwise convolution over K = 27 -> 32 on the SAME parameter memory ([Co][3][3][3] read as [Co][27][1][1]). */
int mi355_pack_input_im2col3(const float* x, void* y, int N, int C, int H, int W, int dtype, mi355_stream_t s);
/* NHWC `dtype` (channel stride ld) -> NCHW fp32 (model outputs / logits). */
int mi355_unpack_output_nchw(const void* x, float* y, int N, int C, int H, int W, int ld, int dtype,
                             mi355_stream_t s);
/* NCHW fp32 -> NHWC `dtype` without padding (incoming output gradients). */
int mi355_pack_nchw(const float* x, void* y, int N, int C, int H, int W, int ld, int dtype, mi355_stream_t s);

/* fp32 parameter [Co][Ci][KH][KW] -> packed forward weights Wf[Co][KH*KW][Cip] and (if wb != NULL)
 * packed data-gradient weights Wb[Cip][KH*KW][Co] (Cip = Ci padded up to a multiple of 32, zero
 * filled).  `transposed` != 0 reads a ConvTranspose2d parameter [Ci][Co][KH][KW] instead
 * (models/segmentation_models/ResnetUnet.py:21). */
int mi355_pack_conv_weight(const float* w, void* wf, void* wb, int Co, int Ci, int Cip, int KH, int KW,
                           int transposed, int dtype, mi355_stream_t s);

/* Every weight pack of a launch plan in ONE launch: `table` = n descriptors of 9 int64 in device memory
 * {w, wf, wb (0 = none), Co, Ci, Cip, KH*KW, transposed, scale (0 = none)}, same semantics as mi355_pack_conv_weight;
 * scale[Co] multiplies the rows of the packs (eval-mode BatchNorm folded into the weights, !transposed only). */
int mi355_pack_conv_weights_batched(const int64_t* table, int n, int fields /* = 9: checked, the table is device memory */,
                                    int dtype, mi355_stream_t s);

/* ---- implicit-GEMM convolution on MFMA -------------------------------------------------
 * out[m][j] (+)= bias[j] + sum_{kh,kw,c} in[src(m,kh,kw)][c] * wk[j][kh*KW+kw][c]
 *   m = (n,ho,wo);  t = o*mul + k*kmul + off (per axis);  valid iff t % div == 0 and
 *   0 <= t/div < (up ? 2*Hi : Hi);  src row = t/div (>>1 when `up`: fused nearest x2).
 * forward conv stride s pad p : mul=s, kmul=+1, off=-p, div=1, wk = Wf      (nn.Conv2d forward,
 *   models/segmentation_models/AttentionUNet.py:6,9,20,32,36,40,84; ResNet.py:17-20,102)
 * data gradient stride s pad p: mul=1, kmul=-1, off=+p, div=s, wk = Wb     (convolution_backward
 *   input grad, reached from loss.backward(), utils/helpers.py:329)
 * ConvTranspose2d(k,s)        : data-gradient form with wk = Wf of the transposed parameter
 *   (ResnetUnet.py:21,51).
 * Ci % 32 == 0 (16 for fp32), Co % 32 == 0.  `accumulate` is a bit set: bit 0 adds the result into `out`,
 * bit 1 applies max(0, .) to conv + bias first (nn.ReLU fused into the epilogue: VGG.py:9-41, and eval-mode
 * Conv -> BN -> ReLU with the BN folded into weights and bias), bit 2 sums every 2x2 group of output pixels into a
 * half-resolution `out` [N][Ho/2][Wo/2] (the gradient of the nearest x2 up-sampling that the forward conv folds into
 * its gather, AttentionUNet.py:19; halo kernels only: mi355_conv2d_igemm_variant(...) in {2, 3, 5, 6, 7, 8}).
 * bf16 / fp16 dispatch: 3x3/s1/p1 with Co % 64 == 0 on images divisible by an 8x32 or 16x16 tile ->
 * conv3x3_halo_rw_kernel (halo patch in LDS by LDS-DMA, patch-row register window); 1x1/s1 with (Ci, Co) among
 * 32/64/128-channel pairs (the attention-gate projections, AttentionUNet.py:33-45) -> conv1x1_stream_kernel (weights in
 * registers, pixels streamed straight into MFMA fragments); everything else -> conv_igemm_dma_kernel (LDS-DMA ring);
 * fp32 -> register-staged conv_igemm_kernel. */
int mi355_conv2d_igemm(const void* in, const void* wk, const float* bias, void* out,
                       int N, int Hi, int Wi, int Ci, int ldi,
                       int Ho, int Wo, int Co, int ldo,
                       int KH, int KW, int mul, int kmul, int off, int div, int up,
                       int accumulate, float* stats, int dtype, mi355_stream_t s);
/* Which kernel serves a shape: 0 generic register-staged, 1 LDS-DMA ring, 2 / 3 halo kernel with 8x32 / 16x16 tiles,
 * 4 streaming pointwise kernel, 5 / 6 the 512-thread ping-pong halo kernels with 64 / 128 output channels per workgroup
 * (5: MI355_HALO_PP=1 only; 6: 3x3/s1/p1, Co % 128 == 0, Ci % 64 == 0 and Ci >= 256, image divisible by 16 x 32 — one workgroup
 * per CU, so the launcher falls back to variant 2 for a batch whose grid would leave more than a fifth of the last round of
 * workgroups empty; mi355_conv2d_igemm_stat_rows, which knows N, follows the launcher); 7 the weight-stationary persistent
 * kernel (3x3/s1/p1, Ci == 64, Co % 64 == 0, image divisible by 8 x 32: the 64-channel layers of every U-Net level,
 * AttentionUNet.py:62-63,82, R2AttU_Net.py:36-39; the launcher falls back to variant 2 below two tiles per workgroup); 8 its
 * Ci == 128 instantiation (4 x 32-pixel tiles: AttentionUNet.py:65-66,78-79,81); 9 the padding-free GEMM kernel (conv_gemm256_kernel:
 * 1x1 at stride 1 / 2, 2x2 / stride 2, ConvTranspose2d(2, 2) as four pointwise phases; Ci % 64 == 0, Co % 128 == 0, whole 256-row
 * tiles, at least 128 of them — the ResNet-50 encoder's 1x1 convolutions and ResnetUnet.py:21,51's transposed convolutions). */
int mi355_conv2d_igemm_variant(int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int mul, int kmul, int off,
                               int div, int up, int dtype);
/* ... and the variant the launcher actually runs for a batch of N images (the batch-dependent fall-backs applied). */
int mi355_conv2d_igemm_variant_n(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int mul, int kmul,
                                 int off, int div, int up, int dtype);
/* Fused BatchNorm statistics: when `stats` != NULL the epilogue also writes, per M tile b of the kernel
 * (per WORKGROUP for the streaming pointwise kernel) it dispatches to, stats[(b*2+0)*Co + c] = sum and stats[(b*2+1)*Co + c] = sum of squares of the
 * (rounded) outputs — the same partial layout mi355_bn_finalize consumes.  The number of tile rows is
 * mi355_conv2d_igemm_stat_rows(...) (0 = not available for this shape/dtype: use mi355_bn_stats). */
/* Output-channel tile (128 / 64 / 32) of the LDS-DMA ring kernel (variant 1) for N x Ho x Wo output rows: the widest that divides
 * Co unless its grid would leave most of the chip idle (bench.py's kernel names). */
int mi355_conv2d_igemm_dma_tile(int N, int Ho, int Wo, int Ci, int Co);
int mi355_conv2d_igemm_generic_tile(int N, int Ho, int Wo, int Co);      /* ... and of the generic kernel (variant 0; fp32) */
int mi355_conv2d_igemm_stat_rows(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW,
                                 int mul, int kmul, int off, int div, int up, int dtype);

/* Weight gradient: ws[split][Co][KH*KW][Ci] = sum over the split's pixel range of
 * dy[m][co] * x[src(m,kh,kw)][ci] (forward addressing as above), then
 * mi355_conv2d_wgrad_reduce sums the splits into the fp32 parameter-gradient layout
 * [Co][Ci_real][KH][KW] (or [Ci_real][Co][KH][KW] when transposed), beta in {0,1}.
 * (convolution_backward weight grad, utils/helpers.py:329.) */
int mi355_conv2d_wgrad_splits(int N, int Ho, int Wo, int Ci, int Co, int KH, int KW);
/* The kernel mi355_conv2d_wgrad runs for a 2-byte layer of this geometry: 0 = the generic per-tap split-K kernel; nine-tap kernels
 * (3x3 / stride 1 / pad 1, Ho % 8 == 0): 1 = four waves, rows of 32-pixel segments; 2 = four waves, 16-pixel-wide images two at a
 * time; 3 = eight waves, rows of 64-pixel segments (wgrad3x3_halo8_kernel); 4 = eight waves, 32-pixel-wide images two at a time. */
int mi355_conv2d_wgrad_variant(int N, int Ho, int Wo, int KH, int KW, int stride, int pad, int dtype);
int mi355_conv2d_wgrad(const void* x, const void* dy, float* ws, int splits,
                       int N, int Hi, int Wi, int Ci, int ldx,
                       int Ho, int Wo, int Co, int ldy,
                       int KH, int KW, int stride, int pad, int up, int dtype, mi355_stream_t s);
/* Weight gradient of ONE 3x3 / stride 1 / pad 1 convolution that was applied `napp` (<= 6) times to different inputs — the
 * shared convolution of a recurrent block (R2AttU_Net.py:29-45: conv(x), then five times conv(x + x1)): dW = sum over the
 * pairs (x_i, dy_i), computed by ONE launch of the nine-tap kernel (the pairs are more work items) into ONE set of partial
 * slabs, followed by ONE mi355_conv2d_wgrad_reduce.  All pairs share geometry and channel strides; unused pairs are NULL.
 * Served when mi355_conv2d_wgrad_multi_ok(N, Ho, Wo, dtype) != 0 (bf16 / fp16, images divisible into the kernel's row
 * segments); splits = mi355_conv2d_wgrad_splits(N * napp, ...), ws = splits * Co * 9 * Ci floats. */
int mi355_conv2d_wgrad_multi_ok(int N, int Ho, int Wo, int dtype);
int mi355_conv2d_wgrad_multi(const void* x0, const void* dy0, const void* x1, const void* dy1, const void* x2, const void* dy2,
                             const void* x3, const void* dy3, const void* x4, const void* dy4, const void* x5, const void* dy5,
                             int napp, float* ws, int splits, int N, int Hi, int Wi, int Ci, int ldx, int Ho, int Wo, int Co,
                             int ldy, int up, int dtype, mi355_stream_t s);
int mi355_conv2d_wgrad_reduce(const float* ws, int splits, float* dw, int Co, int Ci, int Ci_real,
                              int KH, int KW, int transposed, float beta, mi355_stream_t s);

/* ---- per-channel reductions / BatchNorm (nn.BatchNorm2d, AttentionUNet.py:7,10,21,34,38,42) -- */

/* partial[(b*2+0)*C + c] = sum over block b's rows of x[m][c]; [(b*2+1)*C+c] = sum of squares.
 * nblocks = mi355_rowreduce_blocks(M). */
int mi355_rowreduce_blocks(long long M);
int mi355_bn_stats(const void* x, float* partial, long long M, int C, int ld, int dtype, mi355_stream_t s);
/* Train-mode finalize: batch mean / biased var -> scale = gamma*invstd, shift = beta - mean*scale,
 * saves mean/invstd, updates running stats (momentum, unbiased var) and num_batches_tracked. */
int mi355_bn_finalize(const float* partial, int nblocks, long long M, int C, const float* gamma,
                      const float* beta, float* running_mean, float* running_var, int64_t* nbt,
                      float momentum, float eps, float* scale, float* shift, float* mean, float* invstd,
                      mi355_stream_t s);
/* out[j][c] = sum of partial rows [j*per, (j+1)*per) (per = ceil(rows / nsplit)), c < rowlen: pre-folds the one-row-per-tile
 * statistics a convolution epilogue leaves (thousands of rows) so that mi355_bn_finalize reads nsplit rows. */
int mi355_fold_rows(const float* partial, int rows, int rowlen, float* out, int nsplit, mi355_stream_t s);
/* Eval-mode scale/shift from running stats. */
/* out[c] = scale[c] * (bias ? bias[c] : 0) + shift[c]: the bias of a convolution with eval-mode BN folded in. */
int mi355_bn_fold_bias(const float* bias, const float* scale, const float* shift, float* out, int C, mi355_stream_t s);
int mi355_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, int C, float* scale, float* shift,
                         mi355_stream_t s);
/* y = act(x*scale[c]+shift[c] (+ x2*scale2[c]+shift2[c] | + r)) ; act bit0: ReLU, bit1: add `res` AFTER the
 * activation (y = relu(bn(x)) + r, the recurrent-block input of R2AttU_Net.py:44) instead of before it.
 * x2/scale2/shift2 optional second normalised operand (attention gate g1+x1,
 * AttentionUNet.py:51); `res` optional already-activated residual (ResNet.py:43). */
int mi355_bn_act(const void* x, int ldx, const float* scale, const float* shift,
                 const void* x2, int ldx2, const float* scale2, const float* shift2,
                 const void* res, int ldr, void* y, int ldy, long long M, int C, int act, int dtype,
                 mi355_stream_t s);
/* ... and with the MaxPool2d(2, 2) that follows it in every encoder level (AttentionUNet.py:61,89-95; R2AttU_Net.py:92,122-134)
 * in the same pass: y = act(x*scale+shift) as above (no second operand, no residual) AND p[n][h/2][w/2][c] = max of the 2 x 2
 * group of y (of the values as stored).  H and W even.  p == NULL: the activation only — the same kernel serves plain BatchNorm
 * apply passes on even images (a thread owns a 2 x 2 window: measured ≈8 % faster than the row-ordered mi355_bn_act). */
int mi355_bn_act_pool2(const void* x, int ldx, const float* scale, const float* shift, void* y, int ldy, void* p, int ldp,
                       int N, int H, int W, int C, int act, int dtype, mi355_stream_t s);
/* mi355_bn_act's forms WITHOUT a second normalised operand (plain, or with the residual `res` added before the activation — act
 * bit 1: after it, the recurrent block's x + relu(bn(.)), R2AttU_Net.py:44) in the window order of mi355_bn_act_pool2: the same
 * values, ≈8 % faster on even images. */
int mi355_bn_act_windows(const void* x, int ldx, const float* scale, const float* shift, const void* res, int ldr, void* y, int ldy,
                         int N, int H, int W, int C, int act, int dtype, mi355_stream_t s);
/* Backward reductions for y = act(bn(x) [+ other]) given dL/dy:
 * partial sums of g and g*xhat per channel, g = dy * (y > 0 if act).  When `y` is NULL the ReLU mask
 * is recomputed as x*mscale[c]+mshift[c] > 0 (the forward's own coefficients), which saves reading
 * the activated tensor; `y` is required when a residual / second operand was added before the ReLU. */
/* Partial rows mi355_bn_bwd_reduce can leave non-zero (= the workgroups it runs on, at most one per CU); rows from there up to
 * mi355_rowreduce_blocks(M) are written as zeros, so mi355_bn_bwd_finalize may fold this many rows only. */
int mi355_bn_bwd_reduce_rows(long long M);
int mi355_bn_bwd_reduce(const void* dy, int lddy, const void* y, int ldy, const void* x, int ldx,
                        const float* mean, const float* invstd, const float* mscale, const float* mshift,
                        float* partial, long long M, int C, int act, int dtype, mi355_stream_t s);
/* sums[0..C) = sum g (dbeta), sums[C..2C) = sum g*xhat (dgamma); beta-accumulate into dgamma/dbeta. */
int mi355_bn_bwd_finalize(const float* partial, int nblocks, int C, float* sums, float* dgamma, float* dbeta,
                          float acc, mi355_stream_t s);
/* The same fold over partial rows of nq quantities per channel: (sum g, sum g*xhat) = quantities (q0, q1) of each row. */
int mi355_bn_bwd_finalize_at(const float* partial, int nblocks, int nq, int q0, int q1, int C, float* sums, float* dgamma,
                             float* dbeta, float acc, mi355_stream_t s);
/* dx = gamma*invstd*(g - sum_g/M - xhat*sum_gx/M); optional dres = g (residual / second operand
 * gradient, pre-normalisation: the ReLU mask applied); optional dpost (+)= dy, the UNMASKED incoming gradient, for an
 * operand that was added AFTER the activation (x + relu(bn(.)), R2AttU_Net.py:44; post_acc != 0 accumulates) — the
 * pass reads dy anyway; optional bias-gradient partials (column sums of dx). */
int mi355_bn_bwd_apply(const void* dy, int lddy, const void* y, int ldy, const void* x, int ldx,
                       const float* gamma, const float* mean, const float* invstd, const float* mscale,
                       const float* mshift, const float* sums, void* dx, int lddx, void* dres, int lddres,
                       void* dpost, int lddpost, int post_acc, float* dbias_partial,
                       long long M, int C, int act, int dtype, mi355_stream_t s);
/* The same pass for the LAST of several applications whose incoming gradients all flow to the same post-activation operand (the
 * recurrent block's x in x + relu(bn(conv(.))), applied t times with shared weights: R2AttU_Net.py:41-44): dpost (+)= dy + ex0 +
 * ex1 + ex2 + ex3, the earlier applications' incoming gradients (ex1..ex3 may be NULL; row pitch ldex), summed in fp32 and rounded
 * once — the earlier applications' passes then carry no dpost at all (one pass over d x instead of t read-modify-writes). */
int mi355_bn_bwd_apply_post4(const void* dy, int lddy, const void* y, int ldy, const void* x, int ldx,
                             const float* gamma, const float* mean, const float* invstd, const float* mscale,
                             const float* mshift, const float* sums, void* dx, int lddx, void* dpost, int lddpost,
                             int post_acc, const void* ex0, const void* ex1, const void* ex2, const void* ex3, int ldex,
                             long long M, int C, int act, int dtype, mi355_stream_t s);
/* The same two passes for a BatchNorm + ReLU layer whose activation also feeds a MaxPool2d(2, 2) (mi355_bn_act_pool2;
 * AttentionUNet.py:61,89-95): dp, the gradient of the POOLED tensor [N][H/2][W/2][C], is added on the fly to the pixels that are
 * the first maximum of their window (torch's tie rule, the activation recomputed from x with mscale / mshift exactly as the
 * forward rounded it), so mi355_maxpool_bwd's pass over dy is not run: g = relu'(.) * (dy + [first max] * dp).  partial / sums /
 * dx as in mi355_bn_bwd_reduce / _apply with act = 1 and no y.  dy == NULL: the pooling is the activation's only consumer (VGG.py),
 * g = relu'(.) * [first max] * dp.  mi355_bn_bwd_pool2_ok: 1 when the window-ordered pass covers the
 * geometry (C / (16 / element size) a power of two <= 32, W a power-of-two multiple of 512 / that), else use the separate passes. */
int mi355_bn_bwd_pool2_ok(int H, int W, int C, int dtype);
int mi355_bn_bwd_reduce_pool2_rows(long long M);      /* partial rows mi355_bn_bwd_reduce_pool2 can leave non-zero (cf. mi355_bn_bwd_reduce_rows) */
int mi355_bn_bwd_reduce_pool2(const void* dy, int lddy, const void* dp, int lddp, const void* x, int ldx, const float* mean,
                              const float* invstd, const float* mscale, const float* mshift, float* partial,
                              int N, int H, int W, int C, int dtype, mi355_stream_t s);
int mi355_bn_bwd_apply_pool2(const void* dy, int lddy, const void* dp, int lddp, const void* x, int ldx, const float* gamma,
                             const float* mean, const float* invstd, const float* mscale, const float* mshift,
                             const float* sums, void* dx, int lddx, int N, int H, int W, int C, int dtype, mi355_stream_t s);
/* out[c] (+)= sum_b partial[b*stride*C + c]  (used for conv bias gradients). */
int mi355_colsum_finalize(const float* partial, int nblocks, int stride, int C, float* out, float acc,
                          mi355_stream_t s);
/* partial column sums of a plain tensor (bias gradient of convs without BN). */
int mi355_colsum(const void* x, int ld, float* partial, long long M, int C, int dtype, mi355_stream_t s);

/* ---- pooling / resampling / elementwise ------------------------------------------------- */
/* nn.MaxPool2d(k, s, p) forward and backward (AttentionUNet.py:61; ResNet.py:105); backward routes
 * to the first maximum in window scan order, accumulating into dx when accumulate != 0. */
int mi355_maxpool_fwd(const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C,
                      int k, int stride, int pad, int dtype, mi355_stream_t s);
int mi355_maxpool_bwd(const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx,
                      int N, int H, int W, int C, int k, int stride, int pad, int accumulate, int dtype,
                      mi355_stream_t s);
/* gradient of nn.Upsample(scale_factor=2) (nearest): dx[h][w] = sum of the 2x2 block of dy. */
int mi355_upsample2_bwd(const void* dy, int lddy, void* dx, int lddx, int N, int H, int W, int C,
                        int accumulate, int dtype, mi355_stream_t s);
/* y = a + b (b optional => copy), strided. */
int mi355_add(const void* a, int lda, const void* b, int ldb, void* y, int ldy, long long M, int C, int dtype,
              mi355_stream_t s);
/* y = relu(x) forward; dx = dy * (y > 0) backward (VGG.py:10). */
int mi355_relu_fwd(const void* x, int ldx, void* y, int ldy, long long M, int C, int dtype, mi355_stream_t s);
int mi355_relu_bwd(const void* dy, int lddy, const void* y, int ldy, void* dx, int lddx, long long M, int C,
                   int dtype, mi355_stream_t s);

/* ---- attention gate / single-output 1x1 conv (AttentionUNet.py:29-54, 84) ------------------ */
/* z[m] = b + sum_c x[m][c]*w[c]  (ONE output channel of a Conv2d(C,K,1): the attention gate's psi, K = 1, and the logit
 * heads `out` / `conv_1x1` with out_channel = K, AttentionUNet.py:84, R2AttU_Net.py:117); optional per-block partial
 * (sum z, sum z^2).  K > 1: the launch produces channel plane k of an NCHW fp32 map [N][K][HW] — pass z + k*HW (and the
 * k-th weight row / bias); pixel m = n*HW + p lands at z[n*K*HW + p].  K == 1: HW is ignored. */
int mi355_rowdot_fwd(const void* x, int ldx, const float* w, const float* b, float* z, float* partial,
                     long long M, int C, int HW, int K, int dtype, mi355_stream_t s);
/* dx[m][c] = dz[m]*w[c] (masked by x>0 if relu_mask), added to dx when accumulate (the further planes of a K-channel
 * head); partial dw[c] = sum_m dz[m]*x[m][c], db.  dz is indexed like z above. */
int mi355_rowdot_bwd(const float* dz, const void* x, int ldx, const float* w, void* dx, int lddx,
                     float* partial, long long M, int C, int relu_mask, int HW, int K, int accumulate, int dtype,
                     mi355_stream_t s);
/* y[m][c] = x[m][c] * sigmoid(z[m]*scale[0]+shift[0]) */
int mi355_gate_mul_fwd(const void* x, int ldx, const float* z, const float* scale, const float* shift,
                       void* y, int ldy, long long M, int C, int dtype, mi355_stream_t s);
/* dx (+)= dy*psi ; dzn[m] = (sum_c dy*x) * psi*(1-psi); partial (sum dzn, sum dzn*zhat). */
int mi355_gate_mul_bwd(const void* dy, int lddy, const void* x, int ldx, const float* z, const float* scale,
                       const float* shift, const float* mean, const float* invstd, void* dx, int lddx,
                       int accumulate, float* dzn, float* partial, long long M, int C, int dtype,
                       mi355_stream_t s);
/* scalar-field BN backward: dz = gamma*invstd*(dzn - S0/M - zhat*S1/M). */
int mi355_bn1_bwd_apply(const float* dzn, const float* z, const float* gamma, const float* mean,
                        const float* invstd, const float* sums, float* dz, long long M, mi355_stream_t s);
/* psi_in = relu(BN_g(g1) + BN_x(x1)) and z = psi conv of it in ONE pass over the raw branch outputs (instead of mi355_bn_act with a
 * second operand + mi355_rowdot_fwd): psi_in is computed as mi355_bn_act would have stored it and is not stored — the backward
 * below recomputes it as well.  partial: per-block (sum z, sum z^2) or null.  _ok: C / (16 / element size) <= 64.
 * x1 == NULL (with scale_x / shift_x NULL): ONE normalised operand, z = w . relu(BN_g(g1)) + b — the logit head behind the last
 * decoder layer (AttentionUNet.py:84,119); the two backward entry points accept the same (the x-branch arguments NULL, quantity 2
 * of the partial rows zero, dx1 not written). */
int mi355_gate_psi_fwd_ok(int C, int dtype);
int mi355_gate_psi_fwd(const void* g1, int ldg, const void* x1, int ldx, const float* scale_g, const float* shift_g,
                       const float* scale_x, const float* shift_x, const float* w, const float* b, float* z,
                       float* partial, long long M, int C, int dtype, mi355_stream_t s);
/* Backward of the gate's two normalised branches in two passes instead of mi355_rowdot_bwd + 2 x (mi355_bn_bwd_reduce,
 * mi355_bn_bwd_apply) (AttentionUNet.py:32-38,48-52: psi_in = relu(BN_g(W_g g) + BN_x(W_x x)), z = psi conv).  g1 / x1 are the RAW
 * branch convolution outputs, scale / shift / mean / invstd the forward's coefficients of the two BatchNorms, w the psi
 * convolution's weight [C], dz the gradient of its output: the gradient of psi_in, dz[m] * w[c] where psi_in > 0, is recomputed
 * on the fly (psi_in exactly as mi355_bn_act rounded it) and never stored.  reduce leaves FIVE quantities per channel and partial
 * row — sum dp, sum dp * xhat_g, sum dp * xhat_x, sum dz * psi_in (the psi weight's gradient), sum dz (its bias's) — which
 * mi355_bn_bwd_finalize_at (quantities (0, 1) and (0, 2) of 5) and mi355_colsum_finalize (stride 5) fold; apply writes the input
 * gradients of both BatchNorms.  mi355_gate_bn_bwd_reduce_rows: the partial rows that can be non-zero. */
int mi355_gate_bn_bwd_reduce_rows(long long M);
int mi355_gate_bn_bwd_reduce(const float* dz, const void* g1, int ldg, const void* x1, int ldx, const float* scale_g,
                             const float* shift_g, const float* mean_g, const float* invstd_g, const float* scale_x,
                             const float* shift_x, const float* mean_x, const float* invstd_x, const float* w,
                             float* partial, long long M, int C, int dtype, mi355_stream_t s);
int mi355_gate_bn_bwd_apply(const float* dz, const void* g1, int ldg, const void* x1, int ldx, const float* scale_g,
                            const float* shift_g, const float* mean_g, const float* invstd_g, const float* scale_x,
                            const float* shift_x, const float* mean_x, const float* invstd_x, const float* w,
                            const float* gamma_g, const float* gamma_x, const float* sums_g, const float* sums_x,
                            void* dg1, int lddg, void* dx1, int lddx, long long M, int C, int dtype, mi355_stream_t s);

/* ---- heads (ResNet.py:112-115, VGG.py:109-119) --------------------------------------------- */
int mi355_global_pool_fwd(const void* x, int ldx, float* y, int32_t* argmax, int N, int HW, int C, int is_max,
                          int dtype, mi355_stream_t s);
int mi355_global_pool_bwd(const float* dy, const int32_t* argmax, void* dx, int lddx, int N, int HW, int C,
                          int is_max, int dtype, mi355_stream_t s);
/* y[b][o] = act(bias[o] + sum_i x[b][i]*w[o][i]) (fp32, tiny M); backward pieces. */
int mi355_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int I, int O, int relu,
                     mi355_stream_t s);
int mi355_linear_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw,
                     float* db, int B, int I, int O, int relu, float beta, float* scratch, mi355_stream_t s);
/* fp32 elements of `scratch` mi355_linear_bwd needs to produce dx (0 for small layers; the wide layers of the torchvision
 * VGG head — 25088 x 4096 — stream W once per 16 batch rows and fold 16 deterministic output-range partials). */
int mi355_linear_bwd_scratch(int B, int I, int O);
/* inverted dropout with a counter hash (seed, counter[0], element index): mask byte kept for backward;
 * `counter` is a device word the host bumps per forward so a static launch plan draws fresh masks. */
int mi355_dropout_fwd(const float* x, float* y, uint8_t* mask, long long n, float p, uint64_t seed,
                      const int32_t* counter, mi355_stream_t s);
int mi355_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, long long n, float p, mi355_stream_t s);

/* ---- losses (utils/helpers.py:244-246) ---------------------------------------------------- */
/* BCEWithLogitsLoss (mean): loss[0] and dz = (sigmoid(z)-t)/n * gscale[0] (NCHW fp32 with C==1 is
 * the same memory as NHWC, so logits/targets are taken as flat arrays). */
int mi355_bce_logits(const float* z, const float* t, float* loss, float* dz, const float* gscale, long long n,
                     mi355_stream_t s);
/* CrossEntropyLoss(label_smoothing): loss[0], dz[B][C]. */
int mi355_ce_smooth(const float* z, const int64_t* y, float* loss, float* dz, const float* gscale, int B,
                    int C, float smoothing, mi355_stream_t s);

/* ---- optimiser on flat fp32 buffers (utils/helpers.py:251,304,332-336) ---------------------- */
/* sumsq partials of a flat gradient buffer; nblocks = mi355_rowreduce_blocks(n). */
int mi355_sumsq_partial(const float* g, float* partial, long long n, mi355_stream_t s);
/* norm[0] = sqrt(sum partial) * inv_scale * (dev_scale ? dev_scale[0] : 1); coef[0] = min(1, max_norm/(norm+1e-6))
 * (clip_grad_norm_); found_inf[0] = 1 if the norm is not finite; otherwise step[0] += 1 (the optimiser step counter).
 * dev_scale: optional device-side factor — the loss scaler's 1/scale (GradScaler.unscale_, helpers.py:329). */
int mi355_clip_coef(const float* partial, int nblocks, float max_norm, float inv_scale, const float* dev_scale,
                    float* norm, float* coef, float* found_inf, int32_t* step, mi355_stream_t s);
/* AdamW, decoupled decay; g is multiplied by coef[0]*inv_scale*(dev_scale ? dev_scale[0] : 1) first; skipped when
 * found_inf[0] != 0 (GradScaler.step).  step_count is read from device memory (graph-capturable). */
int mi355_adamw(float* p, const float* g, float* m, float* v, long long n, const float* lr, float beta1,
                float beta2, float eps, float wd, const float* coef, float inv_scale, const float* dev_scale,
                const float* found_inf, const int32_t* step, mi355_stream_t s);
/* step[0] += 1 unless found_inf[0] != 0 (found_inf may be NULL). */
int mi355_step_tick(int32_t* step, const float* found_inf, mi355_stream_t s);
/* torch.amp.GradScaler.update() (helpers.py:285,336) on device state: found_inf -> scale *= backoff, tracker = 0;
 * otherwise ++tracker == interval -> scale *= growth, tracker = 0;  inv_scale = 1/scale. */
int mi355_amp_update(float* scale, float* inv_scale, int32_t* growth_tracker, const float* found_inf, float growth,
                     float backoff, int interval, mi355_stream_t s);
int mi355_fill_f32(float* p, float v, long long n, mi355_stream_t s);

/* AdaptiveAvgPool2d((OH,OW)) + Flatten of the torchvision VGG head (helpers.py:158-166 loads `vgg16_bn`):
 * NHWC `dtype` activations -> fp32 [N][C*OH*OW] in NCHW flatten order; backward writes (does not accumulate) dx. */
int mi355_adaptive_avgpool_fwd(const void* x, int ldx, float* y, int N, int H, int W, int C, int OH, int OW, int dtype,
                               mi355_stream_t s);
int mi355_adaptive_avgpool_bwd(const float* dy, void* dx, int lddx, int N, int H, int W, int C, int OH, int OW, int dtype,
                               mi355_stream_t s);

/* ---- input pipeline on the GPU (utils/trainer.py:52-115 Albumentations transforms; utils/dataset.py:100-134) ----------- */
/* dst[n][y][x][c] (uint8) = sample of src[n] ([Hs][Ws][C] uint8, C <= 4) at (sx, sy) = m[n] (2x3, row-major) applied to the dst
 * pixel (x, y): bilinear with cv2's rounding, or nearest; border = replicate (A.Resize) or reflect-101 (A.ShiftScaleRotate). */
int mi355_warp_u8(const uint8_t* src, int N, int Hs, int Ws, int C, const float* m, uint8_t* dst, int H, int W, int nearest,
                  int reflect, mi355_stream_t s);
/* out[n][c][y][x] (fp32 NCHW, what ToTensorV2 yields) = (clip(round(alpha_n*v + beta_n*255)) / 255 - mean[c]) / std[c];
 * bc = [N][2] (alpha, beta) or NULL (A.RandomBrightnessContrast off); mean == NULL: v / 255 (masks, dataset.py:126). */
int mi355_normalize_u8(const uint8_t* src, int N, int H, int W, int C, const float* bc, const float* mean, const float* stdv,
                       float* out, mi355_stream_t s);

/* ---- PNG files -> uint8 batch on the host (utils/dataset.py:55,101-102: PIL Image.open(path).convert("RGB" | "L")) ----------- */
/* Header fields of a PNG held in memory.  MI355_ERR_UNSUPPORTED (fields still filled) for an unknown interlace method or more
 * than 32768 pixels per side. */
int mi355_png_info(const uint8_t* file, long long nbytes, int* W, int* H, int* color_type, int* bit_depth);
/* out[H][W][channels] uint8 = what PIL yields for .convert("RGB") (channels = 3) or .convert("L") (channels = 1): gray replicated,
 * alpha dropped, palette looked up, 1/2/4-bit gray scaled to 0..255, RGB -> L by (19595 R + 38470 G + 7471 B + 0x8000) >> 16;
 * 16-bit RGB / RGBA / gray+alpha keep the high byte, 16-bit gray (PIL mode I;16) SATURATES at 255; Adam7-interlaced files too. */
int mi355_png_decode(const uint8_t* file, long long nbytes, int channels, uint8_t* out, long long out_bytes);
/* n files of one size W x H into out + i * stride (a pinned host batch), decoded by `threads` native threads. */
int mi355_png_decode_batch(const uint8_t* const* files, const long long* nbytes, int n, int channels, uint8_t* out,
                           long long stride, int W, int H, int threads);

/* ---- joint inference pipeline glue (utils/pipeline.py:324-357 classify, 359-418 process_image) ---------- */
/* pred[b] = argmax_c logits[b][c] (first maximum), conf[b] = 100 * max softmax; kept[0..n_kept) = the batch indices
 * with pred == keep_class, in order.  B <= 1024. */
int mi355_cls_decide(const float* logits, int B, int C, int keep_class, int32_t* pred, float* conf, int32_t* kept,
                     int32_t* n_kept, mi355_stream_t s);
/* y[i] = x[idx[i]], rows of `row` fp32 elements (row % 4 == 0): compacts the kept images. */
int mi355_gather_rows(const float* x, const int32_t* idx, int n, long long row, float* y, mi355_stream_t s);
/* out[idx[i]][p] = sigmoid(logit[i][p]) > thr ? 255 : 0 (pipeline.py:350-352); out is [B][per] uint8, zeroed by the caller. */
int mi355_mask_scatter(const float* logit, const int32_t* idx, int n, long long per, float thr, uint8_t* out,
                       mi355_stream_t s);

/* ---- segmentation metrics counters (utils/tester.py:92-193; helpers.py:223-227) ------------- */
/* per sample b: counts[b*4+{0,1,2,3}] = tp, pred-positive, target-positive, equal   (p = prob > thr) */
int mi355_seg_counts(const float* prob_or_logit, const float* target, float* counts, int B, long long per,
                     int is_logit, float thr, mi355_stream_t s);

/* ---- launch-plan replay: the per-batch host loop of utils/helpers.py:317-342 (model(x) ... loss.backward()) as ONE call ---- */
/* A plan is a table of pre-resolved launches of the entry points above: `name`, its arguments as 64-bit slots in prototype
 * order INCLUDING the trailing stream (pointers and integers by value, floats as their IEEE-754 bit pattern in the low 32
 * bits), flags bit 0 = "runs on the side stream" (weight-gradient launches).  mi355_plan_run issues launches [first, last) in
 * order; whenever a run of side-stream launches starts, the side stream first waits for everything issued on the main stream
 * so far (event fork).  mi355_plan_join makes the main stream wait for the side stream.  Nothing synchronises with the host.
 * mi355_plan_arity(name) = number of slots the entry point takes, or -1 if `name` is not a launcher of this header. */
int mi355_plan_arity(const char* name);
void* mi355_plan_create(int n);
int mi355_plan_set(void* plan, int i, const char* name, const uint64_t* args, int nargs, int flags);
int mi355_plan_patch(void* plan, int i, int arg, uint64_t value);
int mi355_plan_run(void* plan, int first, int last, mi355_stream_t main_stream, mi355_stream_t side_stream);
int mi355_plan_join(void* plan, mi355_stream_t main_stream, mi355_stream_t side_stream);
int mi355_plan_last_index(void* plan);
int mi355_plan_destroy(void* plan);

/* ---- data-parallel gradient exchange ----------------------------------------------------------------
 * The reference is single-device (no DDP / NCCL anywhere: utils/trainer.py:119-213); the data-parallel step of this path sums
 * the flat fp32 gradient buffer over the GPUs of a node in a few contiguous buckets.  One process per GPU, one communicator per
 * process, RCCL resolved at run time (the librccl.so already loaded by PyTorch when there is one).  mi355_comm_unique_id: rank 0
 * fills a 128-byte id (ncclUniqueId) that the host distributes; mi355_comm_init: collective over all ranks with that id;
 * mi355_allreduce_bucket: in-place sum of `count` elements at `ptr` over the ranks, enqueued on `s` (the caller orders `s` behind
 * the bucket's last writer); mi355_comm_world: ranks of the live communicator (0 = none); mi355_comm_destroy. */
int mi355_comm_unique_id(void* id128);
int mi355_comm_init(int rank, int world, const void* id128);
int mi355_comm_world(void);
int mi355_allreduce_bucket(void* ptr, long long count, int dtype, mi355_stream_t s);
int mi355_comm_destroy(void);
/* bf16 / fp16 buckets (SURVEY.md 8e "bf16 (perf) buckets"; the reference has no gradient exchange at all, utils/helpers.py:329-335):
 * wire[i] = round(g[i]) for a bucket of n gradients, and back (g[i] = wire[i]) behind the all-reduce of the staging buffer;
 * g 16-byte aligned, wire 8-byte aligned (a bucket starts on a multiple of four gradients), dtype = MI355_BF16 / MI355_F16. */
int mi355_grads_to_wire(const float* g, void* wire, long long n, int dtype, mi355_stream_t s);
int mi355_grads_from_wire(const void* wire, float* g, long long n, int dtype, mi355_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* MI355CONV_H_ */
