/*
 * unet_hip.h -- C ABI of libunet_hip.so: the MI355X (gfx950) kernels behind the UNet
 * segmentation train-step path.
 *
 * The reference (Florescence/UNet-Medical-Image-Contour-Segmentation) has no FFI of its own:
 * its hot path is the torch.nn surface of unet/unet_parts.py, utils/dice_score.py,
 * utils/boundary_loss.py and train.py:113-159.  Each entry point below
 * names the reference statement(s) it replaces.  Conventions for every function:
 *
 *   - plain pointers and sizes only; device pointers unless a parameter says "host";
 *   - activations are NHWC ("channels_last", train.py:113,262): element (b,h,w,c) of a tensor
 *     with pixel stride `ld` (elements) lives at ((b*H + h)*W + w)*ld + c.  `ld` >= C lets a
 *     tensor be a channel slice of a wider buffer (zero-copy torch.cat of unet_parts.py:95);
 *   - `dt` is the activation dtype: UH_F32 or UH_BF16; statistics, reductions, weights' master
 *     copies and all gradients of parameters are fp32;
 *   - no allocation, no host sync, no global mutable state inside: workspaces are caller owned,
 *     kernels are enqueued on `stream` (a hipStream_t) and the call returns immediately;
 *   - returns 0 on success, a negative UH_E* code otherwise; uh_last_error() gives the text
 *     (thread-local).  The Python shim raises RuntimeError from it.
 */
#ifndef UNET_HIP_H
#define UNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* uh_stream;            /* hipStream_t */

enum { UH_F32 = 0, UH_BF16 = 1,
       /* fp32 tensors, 3x3 conv products on the bf16 matrix pipe ("bf16x3": w*x ~= wh*xh + wh*xl + wl*xh with bf16
        * halves, fp32 accumulate, ~1e-5 relative).  Accepted by uh_pack_w3x3 (writes [hi | lo] bf16 arrays of
        * Cout*9*Cin elements each into the same number of bytes as the fp32 pack), uh_conv3x3_fwd,
        * uh_conv3x3_fwd_affine_relu (MFMA-aligned shapes only: Cin % 16 == 0, Cout % 64 == 0) and uh_conv3x3_wgrad
        * (channel counts multiples of 64). */
       UH_F32X3 = 2,
       /* Flags OR-ed into the dtype argument of uh_pack_w3x3 and uh_conv3x3_fwd / uh_conv3x3_fwd_affine_relu: the packed
        * filter is FRAGMENT-MAJOR instead of KRSC -- 1 KiB blocks, one per (16 filter rows, tap, 64-byte K chunk), laid out
        * as the 64 lanes of an MFMA A-operand read them, so that the conv kernel's fragment load is 8 whole cache lines
        * instead of 16 half lines (the forward kernel was bound by its vector-memory path, not by the matrix pipe).  Only
        * for calls that uh_conv3x3_wfrag_ok() accepts (the LDS-DMA MFMA kernel); UH_WFRAG = forward copy / the `w` of a
        * conv call, UH_WFRAG_D = backward-data copy of uh_pack_w3x3 (it is the `w` of the backward-data conv call, which
        * then passes UH_WFRAG).  In uh_pack_w3x3_batched the flags are per layer (table column 9: bit 0 forward copy,
        * bit 1 backward-data copy). */
       UH_WFRAG = 0x100, UH_WFRAG_D = 0x200 };
enum { UH_OK = 0, UH_EINVAL = -1, UH_ELAUNCH = -2, UH_EWORKSPACE = -3 };

const char* uh_last_error(void);
int uh_version(void);

/* ---- parameter layout --------------------------------------------------------------------
 * nn.Conv2d weight [O,I,3,3] fp32 with arbitrary strides (contiguous or channels_last,
 * train.py:262) -> KRSC [O][3][3][I] in `dt` (w_fwd) and the flipped/transposed copy
 * [I][3][3][O] (w_dgrad, may be NULL) that turns conv3x3_fwd into the data-gradient conv. */
int uh_pack_w3x3(const float* w, int64_t sO, int64_t sI, int64_t sH, int64_t sW, int Cout, int Cin,
                 void* w_fwd, void* w_dgrad, int dt, uh_stream stream);
/* All layers at once: table = nlayers x 10 int64 on the DEVICE {w pointer, sO, sI, sH, sW, Cout, Cin, first element
 * of the layer in the flat outputs, first tile of the layer, flags}; a layer has ceil(Cout/32)*ceil(Cin/32)*9 tiles and
 * ntiles is their sum.  flags: bit 0 / bit 1 = forward / backward-data copy fragment-major (UH_WFRAG).  w_dgrad_flat may
 * be NULL. */
int uh_pack_w3x3_batched(const int64_t* table, int nlayers, int64_t ntiles, void* w_fwd_flat,
                         void* w_dgrad_flat, int dt, uh_stream stream);
/* KRSC fp32 weight gradient -> gradient tensor with the parameter's own strides. */
int uh_unpack_dw3x3(const float* dw_krsc, float* dw, int64_t sO, int64_t sI, int64_t sH, int64_t sW,
                    int Cout, int Cin, uh_stream stream);

/* ---- nn.Conv2d(k=3, padding=1, bias=False)  (unet_parts.py:15,18) --------------------------
 * y[b,h,w,o] = sum_{r,s,i} x[b,h+r-1,w+s-1,i] * w[o][r][s][i]; the input is the virtual channel
 * concat of (x0:C0) and (x1:C1) (x1 may be NULL with C1 = 0).
 * stat_partials (may be NULL): nslab = uh_conv3x3_stat_slabs(); fp32 [nslab][2][Cout] per-slab
 * (mean, M2 = sum (y - mean)^2) of the STORED y over the slab's pixels, then [nslab] pixel counts
 * (a kernel that needs fewer slabs writes 0 counts for the rest), then [nslab] scratch:
 * nslab*(2*Cout + 2) floats.  These are BatchNorm2d's batch
 * statistics (unet_parts.py:16,19) without a second pass over y.  With w = w_dgrad this is conv
 * backward-data. */
int uh_conv3x3_stat_slabs(int B, int H, int W, int Cin, int Cout, int dt);
/* 1 if uh_conv3x3_fwd will run this shape on the LDS-DMA MFMA kernel (16-byte aligned pointers assumed), i.e. if the
 * filter may be packed fragment-major (UH_WFRAG); dt = UH_F32 or UH_BF16. */
int uh_conv3x3_wfrag_ok(int B, int H, int W, int C0, int C1, int Cout, int ld0, int ld1, int ldy, int dt);
int uh_conv3x3_fwd(const void* x0, int C0, int ld0, const void* x1, int C1, int ld1,
                   const void* w, void* y, int ldy, int Cout, float* stat_partials,
                   int B, int H, int W, int dt, uh_stream stream);
/* Backward-data of the SECOND conv of a DoubleConv fused with the first pass of the BatchNorm backward of the layer in front of it
 * (unet_parts.py:15-20 differentiated).  The launch computes dx = conv(dy, w_dgrad) as uh_conv3x3_fwd
 * does -- dx is the gradient of z = ReLU(BatchNorm(q)), q [B,H,W,Cdx] the raw output of the first conv -- and, from its
 * accumulators (rounded to bf16 as stored) and one read of q, the per-workgroup partial sums
 *     partials[row][0][c] = sum_p g,   partials[row][1][c] = sum_p g * (q - mean[c]) * rstd[c],    g = dx where q*scale+shift > 0 else 0
 * in the [nblk][2][C] layout uh_bn_relu_bwd_apply / uh_bn_bwd_finalize take: uh_bn_relu_bwd_reduce (a read of dx and of q) is not
 * launched for that layer.  coef = [scale | shift | mean | rstd], Cdx floats each (what uh_bn_finalize produced in the forward).
 * bf16, MFMA-aligned single-source shapes below 2 GiB, ldq == lddx: uh_conv3x3_dgrad_bnsum_rows() returns the number of partial rows the
 * call writes (nblk), or 0 when the shape must take the two separate kernels.  dt may carry UH_WFRAG (pack of w_dgrad). */
int uh_conv3x3_dgrad_bnsum_rows(int B, int H, int W, int Cdy, int Cdx, int lddy, int lddx, int ldq, int dt);
int uh_conv3x3_dgrad_bnsum(const void* dy, int Cdy, int lddy, const void* w_dgrad, void* dx, int lddx, int Cdx,
                           const void* q, int ldq, const float* coef, float* partials,
                           int B, int H, int W, int dt, uh_stream stream);
/* The stem of the network with a RECOMPUTED output (inc.double_conv.0-2: Conv2d(1 -> 64) -> BatchNorm2d -> ReLU on a
 * single-channel image, unet_parts.py:15-17; unet_model.py:15): the conv output y costs 9 multiply-adds per element and is the
 * largest tensor of the model, so it is never stored -- every consumer rebuilds it from the image on the matrix pipe (a GEMM
 * with K = 9: csrc/stem_mfma.hip) with the roundings of the stored path (y to bf16 before BatchNorm, dy to bf16 before the
 * contraction).  The MFMA adds the nine products in its own order: against the serial-FMA stem kernel of uh_conv3x3_fwd, y
 * differs by one bf16 ulp on about one element in 10^4; all four entry points use the same arithmetic, so the backward pass
 * sees exactly the forward pass's ReLU mask and xhat.
 *   uh_stem_stats               per-workgroup BatchNorm statistics rows of y (layout / row count of uh_conv3x3_fwd: feed uh_bn_finalize)
 *   uh_stem_bn_relu_fwd         z = max(round_bf16(conv(x, w)) * scale + shift, 0)
 *   uh_stem_bn_relu_bwd_reduce  partials[uh_stem_nblk()][2][64] = {sum dz [z>0], sum dz [z>0] xhat}  (finish with uh_bn_bwd_finalize)
 *   uh_stem_bn_relu_bwd_wgrad   dw[64][3][3][1] = sum dy (x) x with dy = scale * (dz [z>0] - dbeta/n - xhat * dgamma/n) rounded
 *                               to bf16 as uh_bn_relu_bwd_apply would store it (n_total: pixel count of the statistics, 0 = B*H*W)
 * bf16, Cin = 1, w = KRSC pack [64][9][1] (uh_pack_w3x3); dz / z 16-byte aligned.  uh_stem_ok() says whether a layer qualifies. */
int uh_stem_ok(int Cin, int Cout, int dt);
int uh_stem_nblk(int B, int H, int W);
int uh_stem_stats(const void* x, int Cin, int ldx, const void* w, float* stat_partials, int B, int H, int W, int dt,
                  uh_stream stream);
int uh_stem_bn_relu_fwd(const void* x, int Cin, int ldx, const void* w, const float* scale, const float* shift, void* z, int ldz,
                        int B, int H, int W, int dt, uh_stream stream);
int uh_stem_bn_relu_bwd_reduce(const void* dz, int lddz, const void* x, int Cin, int ldx, const void* w, const float* scale,
                               const float* shift, const float* mean, const float* rstd, float* partials, int B, int H, int W,
                               int dt, uh_stream stream);
size_t uh_stem_bwd_wgrad_ws_bytes(int B, int H, int W, int Cin);
int uh_stem_bn_relu_bwd_wgrad(const void* dz, int lddz, const void* x, int Cin, int ldx, const void* w, const float* scale,
                              const float* shift, const float* mean, const float* rstd, const float* dgamma, const float* dbeta,
                              int64_t n_total, float* dw_krsc, void* ws, size_t ws_bytes, int B, int H, int W, int dt,
                              uh_stream stream);
/* Inference form of (Conv2d -> BatchNorm2d(eval) -> ReLU)  (unet_parts.py:15-20 under model.eval(),
 * evaluate.py:30 / predict.py:17): z = max(conv(x, w)*scale + shift, 0) with scale/shift from
 * uh_bn_eval_coeffs, applied to the accumulators -- the pre-BatchNorm tensor is never written. */
int uh_conv3x3_fwd_affine_relu(const void* x0, int C0, int ld0, const void* x1, int C1, int ld1,
                               const void* w, void* z, int ldz, int Cout, const float* scale,
                               const float* shift, int B, int H, int W, int dt, uh_stream stream);
/* conv backward-weights: dw[o][r][s][i] = sum_{b,h,w} dy[b,h,w,o] * x[b,h+r-1,w+s-1,i] (fp32 KRSC).
 * bf16 MFMA path: the per-split partial sums (fp32 accumulators) travel through `ws` as block-scaled fp16 -- 11 significant
 * bits, one power-of-two scale per workgroup block -- and are added in fp32 in a fixed order (deterministic); UH_WGRAD_SLAB_F32=1
 * in the environment keeps them fp32. */
size_t uh_conv3x3_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout, int dt);
int uh_conv3x3_wgrad(const void* dy, int lddy, const void* x0, int C0, int ld0,
                     const void* x1, int C1, int ld1, float* dw_krsc, int Cout,
                     void* ws, size_t ws_bytes, int B, int H, int W, int dt, uh_stream stream);
/* Backward-weights in two stages, the second deferred and batched.  uh_conv3x3_wgrad_partials = uh_conv3x3_wgrad without its
 * closing reduction over the pixel splits: the partial results stay in `ws` (which must stay alive and untouched until the
 * reduction has run) and desc[0..7] -- HOST memory -- receives { slabs, dw_krsc, n, nsplit, format, row, blocks, rows per scale block }; desc[3] == 0
 * means the shape took a path without slabs and dw_krsc is already final.  uh_slab_reduce_batched finishes any number of such
 * layers in ONE launch: `table` is DEVICE memory holding the rows (8 int64 each) with [6] replaced by the row's first block
 * (running sum of the block counts), total_blocks their sum.  Filter gradients only feed the optimizer (train.py:157-158), so
 * the eighteen small reduce launches of a backward pass can wait until then.  Results are bit-identical to uh_conv3x3_wgrad. */
int uh_conv3x3_wgrad_partials(const void* dy, int lddy, const void* x0, int C0, int ld0, const void* x1, int C1, int ld1,
                              float* dw_krsc, int Cout, void* ws, size_t ws_bytes, int B, int H, int W, int dt,
                              int64_t* desc, uh_stream stream);
int uh_slab_reduce_batched(const int64_t* table, int nrows, int64_t total_blocks, uh_stream stream);
/* The BatchNorm + ReLU BETWEEN the two convs of a DoubleConv (unet_parts.py:16-17) applied by the CONSUMER conv's loader
 * (SURVEY.md section 7 step 6): x0 is the RAW output y_prev of the previous conv and the layer's input
 *     x = max(y_prev * pre_scale + pre_shift, 0)            (per channel of source 0, rounded to bf16 as uh_bn_relu_apply stores it)
 * is rebuilt in LDS on its way to the MFMAs -- the activation never exists in HBM, uh_bn_relu_apply is not launched.
 * Results are bit-identical to the separate-kernel path.  bf16, one source of <= 512 channels (multiple of 64), Cout % 64 == 0,
 * tensors below 2 GiB: uh_conv3x3_pre_ok() says whether a call qualifies.  dt may carry UH_WFRAG in the forward call.
 * uh_conv3x3_wgrad_pre: backward-weights of such a layer (workspace: uh_conv3x3_wgrad_ws_bytes); backward-data is the
 * ordinary uh_conv3x3_fwd on dy with the transposed filter and yields the gradient of x.
 * BUILD FLAG: the kernel instantiations behind these two calls are compiled only with -DUH_BUILD_PRE=1 (build.py:
 * UH_BUILD_PRE=1 -> libunet_hip_pre.so); measured a net loss on the train step, so the default library
 * answers uh_conv3x3_pre_ok() with 0 and the two calls fail with UH_EINVAL and a message that says so. */
int uh_conv3x3_pre_ok(int B, int H, int W, int C0, int Cout, int ld0, int ldy, int dt);
int uh_conv3x3_fwd_pre(const void* x0, int C0, int ld0, const float* pre_scale, const float* pre_shift, const void* w,
                       void* y, int ldy, int Cout, float* stat_partials, int B, int H, int W, int dt, uh_stream stream);
int uh_conv3x3_wgrad_pre(const void* dy, int lddy, const void* x0, int C0, int ld0, const float* pre_scale,
                         const float* pre_shift, float* dw_krsc, int Cout, void* ws, size_t ws_bytes, int B, int H, int W,
                         int dt, uh_stream stream);
/* Narrow-tensor forms for the small-width models (UNet_S / UNet_T, unet_model.py:52-126: 8..64-channel layers).  The
 * layer is COMPUTED as the next 64-aligned layer (filters zero-padded to C0 + C1 -> Cout, all multiples of 64, so the MFMA
 * kernels apply), but the tensors in HBM hold only their first C0v / C1v / Coutv channels per pixel (multiples of one
 * 16-byte piece: 8 bf16 / 4 fp32; ld* >= the valid count): channels beyond the valid count are read as zeros and never
 * written.  stat_partials / scale / shift are sized for the padded Cout.  fwd: scale == shift == NULL stores the raw conv
 * output (training, + optional statistics); otherwise the inference form z = max(conv*scale + shift, 0).
 * wgrad: dw_krsc is the padded [Cout][3][3][C0 + C1] gradient (rows / columns of padding come out 0); workspace from
 * uh_conv3x3_wgrad_ws_bytes of the padded shape.  Shapes outside the MFMA path return UH_EINVAL.
 * uh_pack_w3x3_padded: uh_pack_w3x3 of the reference-layout filter [Cout][C0 + C1][3][3] into the padded layer
 * [Coutp][Cp0 + Cp1] (source block 0 -> padded channels 0.., block 1 -> Cp0.., zeros elsewhere); the backward-data copy
 * [Cp0 + Cp1][3][3][Coutp] holds the per-source filters as its row blocks (UH_F32 / UH_BF16). */
int uh_pack_w3x3_padded(const float* w, int64_t sO, int64_t sI, int64_t sH, int64_t sW, int Cout, int C0, int C1,
                        int Coutp, int Cp0, int Cp1, void* w_fwd, void* w_dgrad, int dt, uh_stream stream);
int uh_conv3x3_fwd_narrow(const void* x0, int C0, int C0v, int ld0, const void* x1, int C1, int C1v, int ld1,
                          const void* w, void* y, int ldy, int Cout, int Coutv, float* stat_partials,
                          const float* scale, const float* shift, int B, int H, int W, int dt, uh_stream stream);
int uh_conv3x3_wgrad_narrow(const void* dy, int lddy, int Cout, int Coutv, const void* x0, int C0, int C0v, int ld0,
                            const void* x1, int C1, int C1v, int ld1, float* dw_krsc, void* ws, size_t ws_bytes,
                            int B, int H, int W, int dt, uh_stream stream);

/* ---- nn.BatchNorm2d + nn.ReLU(inplace)  (unet_parts.py:16-17,19-20) ------------------------
 * finalize: merge the conv's stat slabs (Chan's formula, double) -> mean, rstd = 1/sqrt(var_biased + eps),
 * scale = gamma*rstd, shift = beta - mean*scale; running stats (may be NULL) updated in place with
 * `momentum` and the UNBIASED variance (n = pixels per channel); *num_batches_tracked (int64 on the
 * device, may be NULL) += 1.  Slab rows whose pixel count is 0 are ignored; n == 0: use the sum of the rows' counts.
 * m2_out (may be NULL): the merged
 * M2 = sum (y - mean)^2 per channel -- with (mean, M2, n) per rank as rows, a second call merges ranks (SyncBN). */
int uh_bn_finalize(const float* stat_partials, int nslab, int C, int64_t n,
                   const float* gamma, const float* beta, float* running_mean, float* running_var,
                   int64_t* num_batches_tracked, float momentum, float eps,
                   float* scale, float* shift, float* mean, float* rstd, float* m2_out, uh_stream stream);
/* uh_bn_finalize over the first C channels of stat rows that are ldc >= C channels wide (small-width layers: the conv, and
 * so its statistics, are laid out for the 64-aligned channel count, uh_conv3x3_fwd_narrow); per-channel arrays: C entries. */
int uh_bn_finalize_ld(const float* stat_partials, int nslab, int ldc, int C, int64_t n,
                      const float* gamma, const float* beta, float* running_mean, float* running_var,
                      int64_t* num_batches_tracked, float momentum, float eps,
                      float* scale, float* shift, float* mean, float* rstd, float* m2_out, uh_stream stream);
/* eval mode: scale/shift from the running statistics. */
int uh_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, int C, float* scale, float* shift,
                      uh_stream stream);
/* z = max(y*scale + shift, 0) */
int uh_bn_relu_apply(const void* y, int ldy, const float* scale, const float* shift,
                     void* z, int ldz, int64_t npix, int C, int dt, uh_stream stream);
/* backward, pass 1: per-channel sums of dz*[z>0] and dz*[z>0]*xhat -> partials [nblk][2][C];
 * returns the number of partial rows it will write through uh_bn_bwd_nblk(). */
int uh_bn_bwd_nblk(int64_t npix, int C);
int uh_bn_relu_bwd_reduce(const void* dz, int lddz, const void* y, int ldy,
                          const float* scale, const float* shift, const float* mean, const float* rstd,
                          float* partials, int64_t npix, int C, int dt, uh_stream stream);
/* backward, pass 2: dgamma = sum2, dbeta = sum1 (written fp32), and
 * dy = scale*(dz*[z>0] - sum1/n - xhat*sum2/n) with n = n_total (0: npix).  nblk == 0: dgamma / dbeta already hold
 * the sums (e.g. all-reduced over the ranks of a data-parallel job: SyncBN, with n_total the global pixel count)
 * and are only read.  uh_bn_bwd_finalize is the first half alone (partials -> dgamma, dbeta). */
int uh_bn_bwd_finalize(const float* partials, int nblk, int C, float* dgamma, float* dbeta, uh_stream stream);
int uh_bn_relu_bwd_apply(const void* dz, int lddz, const void* y, int ldy,
                         const float* scale, const float* shift, const float* mean, const float* rstd,
                         const float* partials, int nblk, float* dgamma, float* dbeta,
                         void* dy, int lddy, int64_t npix, int64_t n_total, int C, int dt, uh_stream stream);

/* ---- BatchNorm + ReLU fused with its consumer (csrc/bn_fused.hip) ---------------------------
 * "pool tail": the second conv of an encoder DoubleConv (unet_parts.py:18-20), whose activation is the skip connection
 * AND the input of nn.MaxPool2d(2) (unet_parts.py:32; unet_model.py:28-32).  uh_bn_relu_pool_ok: H, W even, C a multiple
 * of 16 bytes.  apply: z = relu(bn(y)) [B,H,W,C] and pooled [B,H/2,W/2,C] in one pass over y.  The backward passes
 * replace uh_maxpool2_bwd + uh_bn_relu_bwd_reduce / _apply: dz = dskip (may be NULL) + route(dpool) is rebuilt on the fly
 * (never stored), bit-identical to what the unfused kernels compute; partials / nblk / dgamma / dbeta / n_total as in
 * uh_bn_relu_bwd_reduce / uh_bn_relu_bwd_apply with npix = B*H*W. */
int uh_bn_relu_pool_ok(int B, int H, int W, int C, int dt);
int uh_bn_relu_pool_apply(const void* y, int ldy, const float* scale, const float* shift, void* z, int ldz,
                          void* pooled, int ldp, int B, int H, int W, int C, int dt, uh_stream stream);
int uh_bn_relu_pool_bwd_reduce(const void* dskip, int ldskip, const void* dpool, int lddp, const void* y, int ldy,
                               const float* scale, const float* shift, const float* mean, const float* rstd,
                               float* partials, int B, int H, int W, int C, int dt, uh_stream stream);
int uh_bn_relu_pool_bwd_apply(const void* dskip, int ldskip, const void* dpool, int lddp, const void* y, int ldy,
                              const float* scale, const float* shift, const float* mean, const float* rstd,
                              const float* partials, int nblk, float* dgamma, float* dbeta, void* dy, int lddy,
                              int B, int H, int W, int64_t n_total, int C, int dt, uh_stream stream);
/* "head tail": the last DoubleConv's second conv, whose activation only feeds the 1x1 OutConv (unet_parts.py:103;
 * unet_model.py:37).  uh_bn_relu_head_ok: C = 8 or 16 sixteen-byte channel groups (64 / 128 channels in bf16), n_classes
 * <= 4.  fwd: logits[p][k] = head_b[k] + sum_c relu(bn(y))[p][c] * head_w[k][c] (fp32; the activation is rounded to the
 * tensor dtype first, as if it had been stored) -- z is never written.  bwd_reduce: the BatchNorm partials of
 * dz = dlogits . head_w AND the OutConv gradients dhead_w [ncls][C], dhead_b [ncls] (fp32, written); ws: scratch of
 * uh_bn_relu_head_bwd_ws_bytes.  bwd_apply: as uh_bn_relu_bwd_apply with dz rebuilt from dlogits. */
int uh_bn_relu_head_ok(int C, int ncls, int dt);
int uh_bn_relu_head_fwd(const void* y, int ldy, const float* scale, const float* shift, const float* head_w,
                        const float* head_b, float* logits, int64_t npix, int C, int ncls, int dt, uh_stream stream);
size_t uh_bn_relu_head_bwd_ws_bytes(int64_t npix, int C, int ncls);
int uh_bn_relu_head_bwd_reduce(const float* dlogits, const float* head_w, const void* y, int ldy, const float* scale,
                               const float* shift, const float* mean, const float* rstd, float* partials,
                               float* dhead_w, float* dhead_b, void* ws, size_t ws_bytes, int64_t npix, int C,
                               int ncls, int dt, uh_stream stream);
int uh_bn_relu_head_bwd_apply(const float* dlogits, const float* head_w, const void* y, int ldy, const float* scale,
                              const float* shift, const float* mean, const float* rstd, const float* partials,
                              int nblk, float* dgamma, float* dbeta, void* dy, int lddy, int64_t npix,
                              int64_t n_total, int C, int ncls, int dt, uh_stream stream);

/* ---- nn.MaxPool2d(2)  (unet_parts.py:32) -------------------------------------------------- */
int uh_maxpool2_fwd(const void* x, int ldx, void* y, int ldy, int B, int H, int W, int C, int dt,
                    uh_stream stream);
/* dx = (dskip ? dskip : 0) + route(dy) to the FIRST maximum of each 2x2 window in row-major
 * order; rows/cols beyond 2*floor(H/2) get only dskip. */
int uh_maxpool2_bwd(const void* x, int ldx, const void* dy, int lddy, const void* dskip, int ldskip,
                    void* dx, int lddx, int B, int H, int W, int C, int dt, uh_stream stream);

/* ---- nn.Upsample(2, 'bilinear', align_corners=True) + F.pad  (unet_parts.py:70,85-88) ------
 * x [B,h,w,C] -> y [B,Ho,Wo,C]: the 2h x 2w upsampled image sits at (pad_top, pad_left), the rest
 * of y is zero filled. */
int uh_upsample2x_fwd(const void* x, int ldx, void* y, int ldy, int B, int h, int w, int C,
                      int Ho, int Wo, int pad_top, int pad_left, int dt, uh_stream stream);
/* The same with the BatchNorm + ReLU in front of it applied on the way in: x is the RAW output of the last conv below an Up block
 * (unet_model.py:34-37), the activation max(x*scale + shift, 0) -- read by nothing but nn.Upsample (unet_parts.py:70,80) -- is
 * rounded to the tensor dtype as uh_bn_relu_apply would store it and interpolated, never written.  Bit-identical to
 * uh_bn_relu_apply + uh_upsample2x_fwd.  uh_bn_relu_upsample2x_ok: 1 when the shape takes the fused kernel (C a multiple of a
 * 16-byte piece; pointers / strides 16-byte aligned are checked by the call). */
int uh_bn_relu_upsample2x_ok(int B, int h, int w, int C, int Ho, int Wo, int dt);
int uh_bn_relu_upsample2x_fwd(const void* x, int ldx, const float* scale, const float* shift, void* y, int ldy,
                              int B, int h, int w, int C, int Ho, int Wo, int pad_top, int pad_left, int dt, uh_stream stream);
int uh_upsample2x_bwd(const void* dy, int lddy, void* dx, int lddx, int B, int h, int w, int C,
                      int Ho, int Wo, int pad_top, int pad_left, int dt, uh_stream stream);

/* ---- nn.ConvTranspose2d(Cin, Cout, 2, 2) + F.pad  (unet_parts.py:73,85-88) -----------------
 * w is the parameter itself: [Cin][Cout][2][2] fp32 contiguous; bias [Cout] fp32. */
int uh_convt2x2_fwd(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy,
                    int B, int h, int w_, int Cin, int Cout, int Ho, int Wo, int pad_top, int pad_left,
                    int dt, uh_stream stream);
int uh_convt2x2_dgrad(const void* dy, int lddy, const float* w, void* dx, int lddx,
                      int B, int h, int w_, int Cin, int Cout, int Ho, int Wo, int pad_top, int pad_left,
                      int dt, uh_stream stream);
size_t uh_convt2x2_wgrad_ws_bytes(int B, int h, int w_, int Cin, int Cout);
int uh_convt2x2_wgrad(const void* dy, int lddy, const void* x, int ldx, float* dw, float* dbias,
                      void* ws, size_t ws_bytes, int B, int h, int w_, int Cin, int Cout,
                      int Ho, int Wo, int pad_top, int pad_left, int dt, uh_stream stream);

/* MFMA path of the same layer (csrc/convt_mfma.hip): all three directions as K-contiguous GEMMs over pixels.
 * uh_convt2x2_mfma_ok() tells whether a problem qualifies (channel counts multiples of a 64-byte chunk, tensors
 * below 2 GiB, pixel count a multiple of the chunk); otherwise use the functions above.
 * uh_convt2x2_pack: reference weight [Cin][Cout][2][2] fp32 -> w_fwd [(q,co)][ci] and w_dgrad [ci][(q,co)]
 * (Cin*Cout*4 elements each, activation dtype), q = 2*r + s. */
int uh_convt2x2_mfma_ok(int B, int h, int w_, int Cin, int Cout, int Ho, int Wo, int dt);
int uh_convt2x2_pack(const float* w, int Cin, int Cout, void* w_fwd, void* w_dgrad, int dt, uh_stream stream);
int uh_convt2x2_fwd_mfma(const void* x, int ldx, const void* w_fwd, const float* bias, void* y, int ldy,
                         int B, int h, int w_, int Cin, int Cout, int Ho, int Wo, int pad_top, int pad_left,
                         int dt, uh_stream stream);
int uh_convt2x2_dgrad_mfma(const void* dy, int lddy, const void* w_dgrad, void* dx, int lddx,
                           int B, int h, int w_, int Cin, int Cout, int Ho, int Wo, int pad_top, int pad_left,
                           int dt, uh_stream stream);
size_t uh_convt2x2_wgrad_mfma_ws_bytes(int B, int h, int w_, int Cin, int Cout, int dt);
int uh_convt2x2_wgrad_mfma(const void* dy, int lddy, const void* x, int ldx, float* dw, float* dbias,
                           void* ws, size_t ws_bytes, int B, int h, int w_, int Cin, int Cout,
                           int Ho, int Wo, int pad_top, int pad_left, int dt, uh_stream stream);

/* ---- OutConv: nn.Conv2d(Cin, ncls, 1) with bias  (unet_parts.py:103) -----------------------
 * w [ncls][Cin] fp32, bias [ncls] fp32; logits are fp32 [npix][ncls]. */
int uh_conv1x1_fwd(const void* x, int ldx, const float* w, const float* bias, float* logits,
                   int64_t npix, int Cin, int ncls, int dt, uh_stream stream);
int uh_conv1x1_dgrad(const float* dlogits, const float* w, void* dx, int lddx,
                     int64_t npix, int Cin, int ncls, int dt, uh_stream stream);
size_t uh_conv1x1_wgrad_ws_bytes(int64_t npix, int Cin, int ncls);
int uh_conv1x1_wgrad(const float* dlogits, const void* x, int ldx, float* dw, float* dbias,
                     void* ws, size_t ws_bytes, int64_t npix, int Cin, int ncls, int dt, uh_stream stream);

/* ---- losses ------------------------------------------------------------------------------
 * Binary path (train.py:118-134): t = (mask / mask_div) as float (train.py:119 uses // 2), or t
 * taken from `target_f` when mask is NULL.  sums[0..3] = { sum softplus-BCE, sum sigmoid*t,
 * sum sigmoid, sum t } (fp32, caller zeroes nothing: the kernel overwrites). */
size_t uh_loss_ws_bytes(int64_t n);
int uh_bce_dice_sums(const float* logits, const int64_t* mask, int mask_div, const float* target_f,
                     int64_t n, float* sums, void* ws, size_t ws_bytes, uh_stream stream);
/* dlogits = gscale[0] * ( w_bce*(sigmoid - t)/n + w_dice * d(1 - dice)/dlogit ) with the Dice
 * ratio formed from `sums` (global-batch sums; after a cross-rank all-reduce they give the
 * single-process reference's gradient).  dice_score.py:14-18 including the sets_sum==0 branch.
 * gscale is a device pointer (upstream gradient of the scalar loss); n_mean = element count of
 * the BCE mean (the GLOBAL batch when sharded). */
int uh_bce_dice_grad(const float* logits, const int64_t* mask, int mask_div, const float* target_f,
                     int64_t n, const float* sums, double n_mean, float w_bce, float w_dice,
                     const float* gscale, float* dlogits, uh_stream stream);
/* Multi-class path (train.py:136-142): logits [npix][ncls] fp32, mask int64 class ids.
 * sums = { sum CE, inter[c]..., psum[c]..., tsum[c]... } (1 + 3*ncls floats). */
int uh_ce_dice_sums(const float* logits, const int64_t* mask, int64_t npix, int ncls, float* sums,
                    void* ws, size_t ws_bytes, uh_stream stream);
int uh_ce_dice_grad(const float* logits, const int64_t* mask, int64_t npix, int ncls,
                    const float* sums, double n_mean, float w_ce, float w_dice, const float* gscale,
                    float* dlogits, uh_stream stream);
/* dice_coeff (dice_score.py:5-25) for arbitrary float inputs: per-group sums {sum x*t, sum x, sum t}
 * over `ngroups` contiguous groups of `group_len` elements -> sums [ngroups][3]. */
int uh_dice_sums(const float* x, const float* t, int64_t ngroups, int64_t group_len, float* sums,
                 void* ws, size_t ws_bytes, uh_stream stream);
/* boundary_loss (boundary_loss.py:5-118), value only.  pred [B][H][W] fp32 with element stride
 * `pstride` (4-D [B,C,H,W]-logical channel select = base pointer + stride), target fp32 [B][H][W];
 * out[0] = loss.  ws from uh_loss_ws_bytes(B*H*W). */
int uh_boundary_loss(const float* pred, int64_t pstride, int64_t bstride, const float* target, int B, int H, int W,
                     int edge_width, float edge_weight, float smooth, float* out,
                     void* ws, size_t ws_bytes, uh_stream stream);
/* The same with the target given as the int64 class-index mask and a divisor, target = mask / mask_div (train.py:119 `true_masks
 * //= 2` feeding train.py:134): no float copy of the mask is made. */
int uh_boundary_loss_mask(const float* pred, int64_t pstride, int64_t bstride, const int64_t* mask, int mask_div, int B, int H,
                          int W, int edge_width, float edge_weight, float smooth, float* out, void* ws, size_t ws_bytes,
                          uh_stream stream);

/* ---- clip_grad_norm_ + RMSprop  (train.py:80-81,157-158) -----------------------------------
 * Flat-buffer form: all parameters / gradients / optimizer states live in four equally laid out
 * fp32 buffers of n elements (the Python shim points every nn.Parameter at a view of them).
 * uh_grad_sumsq: norm_out[0] = sqrt(sum g^2)  (clip_grad_norm_'s total_norm).
 * uh_rmsprop_step: coef = min(1, max_norm/(norm+1e-6)); g *= coef (written back, as
 * clip_grad_norm_ does); g += wd*p; v = alpha*v + (1-alpha)*g^2; buf = mu*buf + g/(sqrt(v)+eps);
 * p -= lr*buf.  total_norm is a device pointer (no host sync); max_norm <= 0 disables clipping. */
size_t uh_optim_ws_bytes(int64_t n);
int uh_grad_sumsq(const float* g, int64_t n, float* norm_out, void* ws, size_t ws_bytes, uh_stream stream);
int uh_rmsprop_step(float* p, float* g, float* square_avg, float* momentum_buf, int64_t n,
                    const float* total_norm, float max_norm, float lr, float alpha, float eps,
                    float weight_decay, float momentum, uh_stream stream);

/* ---- scalar assembly ------------------------------------------------------------------------
 * dice_coeff from per-group sums {sum x*t, sum x, sum t} (dice_score.py:14-25): out[0] = mean over
 * groups of (2I+eps)/(S+eps) with S := 2I where S == 0. */
int uh_dice_from_sums(const float* sums, int64_t ngroups, float eps, float* out, uh_stream stream);
/* train.py:121-134: out = { total, bce_mean, dice_loss, boundary } with
 * total = bce_mean + dice_loss + w_boundary*boundary[0]; sums from uh_bce_dice_sums. */
int uh_seg_loss_binary_finish(const float* sums, double n_mean, const float* boundary, float w_boundary,
                              float* out, uh_stream stream);
/* train.py:119-134 for one process in three launches instead of six (+ the torch glue between them): one pass over the
 * logits forms the BCE / Dice partial sums and the prediction's min / max (boundary_loss.py:28), the boundary counts follow,
 * one block finishes.  logits fp32 [B][H][W] dense, mask int64 class ids, target = mask / mask_div.
 * sums[0..3] as uh_bce_dice_sums writes them (uh_bce_dice_grad takes them); out[0..3] = { total, bce_mean, dice_loss,
 * boundary } as uh_seg_loss_binary_finish writes them, out[4] = 1.0 when total is NaN (train.py:149), else 0.0.
 * w_boundary == 0 skips the boundary term (out[3] = 0).  Bit-identical to the separate calls.  Data-parallel runs keep the
 * separate calls (the sums are all-reduced between them).  ws: uh_seg_loss_fused_ws_bytes(). */
size_t uh_seg_loss_fused_ws_bytes(void);
int uh_seg_loss_binary_fused(const float* logits, const int64_t* mask, int mask_div, int B, int H, int W, int edge_width,
                             float edge_weight, float smooth, float w_boundary, double n_mean, float* sums, float* out,
                             void* ws, size_t ws_bytes, uh_stream stream);
/* train.py:137-142: out = { total, ce_mean, dice_loss, boundary }, sums from uh_ce_dice_sums. */
int uh_seg_loss_multiclass_finish(const float* sums, int ncls, double n_mean, const float* boundary,
                                  float w_boundary, float* out, uh_stream stream);

/* ---- connected_component_loss, host part  (utils/connected_component_loss.py:20-59) -----------
 * masks: HOST uint8 [B][H][W], non-zero = (p > 0.5).  Restates cv2.findContours(RETR_EXTERNAL,
 * CHAIN_APPROX_SIMPLE) + cv2.contourArea + cv2.boundingRect (Suzuki-Abe outer borders, shoelace area).
 * out[0] = (sum of small-area and near-edge penalties) / B, out[1] = number of external contours.
 * PARITY UNPINNED (OpenCV is not available in this image). */
int uh_cc_loss_host(const uint8_t* masks, int B, int H, int W, int edge_distance, int min_area, double* out);
/* The same loss with the masks left on the device: hole filling + 8-connected components by union-find,
 * contourArea as the integer sum over 2x2 pixel blocks that the border polygon encloses (full squares + corner halves),
 * boundingRect by integer atomics; out = DEVICE double[2] {sum of penalties / B, number of external contours}. */
size_t uh_cc_loss_ws_bytes(int B, int H, int W);
int uh_cc_loss_device(const uint8_t* masks, int B, int H, int W, int edge_distance, int min_area, void* ws, size_t ws_bytes,
                      double* out, uh_stream stream);

/* ---- inference masks  (predict.py:27, evaluate.py:60-62,111) ------------------------------------
 * uh_argmax_classes: logits fp32 [npix][ncls] -> int64 index of the first maximum per pixel (torch.argmax(dim=1)).
 * uh_threshold_mask: binary head, out = (logit > 0) as 0.0/1.0  (== sigmoid(logit) > 0.5). */
int uh_argmax_classes(const float* logits, int64_t npix, int ncls, int64_t* out, uh_stream stream);
int uh_threshold_mask(const float* logits, int64_t n, float* out, uh_stream stream);

/* ---- mask post-processing on the device  (utils/post_process.py:51-88, used by evaluate.py:71-78) ----
 * mask / out: DEVICE uint8 [B][H][W] class indices {0,1,2}.  Hole filling of the class-2 foreground, k x k opening,
 * 8-connected components with fewer than min_area pixels removed; out is 2 for kept pixels and 0 elsewhere (the
 * reference zeroes the class-1 background too).  ws: uh_postprocess_ws_bytes() bytes.  PARITY UNPINNED (OpenCV). */
size_t uh_postprocess_ws_bytes(int B, int H, int W);
int uh_postprocess_masks(const uint8_t* mask, uint8_t* out, int B, int H, int W, int min_area,
                         int morph_kernel_size, void* ws, size_t ws_bytes, uh_stream stream);

/* ---- input pipeline, device stage  (utils/data_loading.py:65-132, train.py:113-114) ---------------------------
 * What BasicDataset.__getitem__ does to a DECODED image / mask pair, for a whole batch in one pass:
 *   img_u8   DEVICE uint8 [B][Hin][Win][C] (C = 1..4, what np.asarray(PIL image) holds), or NULL (masks only)
 *   mask_u8  DEVICE uint8 [B][Hin][Win] grey levels (255 contour / 128 background / 0 ghost), or NULL (images only)
 *   turns    DEVICE int [B]: quarter turns counter-clockwise of item b (index % 4 of the x4 augmentation,
 *            data_loading.py:100-121), or NULL = no rotation.  odd_turns = 1 when the items' turn counts are odd (every
 *            item of a batch must have the same output shape: Hin x Win for even counts, Win x Hin for odd ones; square
 *            images may mix them: pass odd_turns = 0)
 *   image_out  NHWC [B][Ho][Wo][ld_out >= C] in dt (UH_F32 / UH_BF16): u / 255 when the IMAGE holds a value above 1, else
 *              the raw 0 / 1 value (data_loading.py:86-87); channels C..ld_out-1 are not written
 *   labels_out int64 [B][Ho][Wo]: 255 -> 2, 128 -> 1, anything else -> 0 (data_loading.py:74-78)
 *   flags_ws   DEVICE int [B] workspace (per-image "holds a value above 1")
 * Decode and the BICUBIC / NEAREST rescale of scale < 1 stay on the host. */
int uh_batch_prepare(const uint8_t* img_u8, int C, const uint8_t* mask_u8, const int* turns, int odd_turns, void* image_out,
                     int ld_out, int64_t* labels_out, int* flags_ws, int B, int Hin, int Win, int dt, uh_stream stream);

#ifdef __cplusplus
}
#endif
#endif
