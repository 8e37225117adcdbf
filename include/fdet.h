/*
 * fdet.h -- C-ABI of libfdet_hip.so: the MI355X (gfx950) hot path of a YOLO-style face
 * detector (training step + inference), drop-in for the arithmetic behind the Python
 * surface of smpurkis/PyTorch-Face-Detection-from-Scratch.
 *
 * The reference is pure Python and has no FFI of its own; each entry point below cites
 * the reference function (path:line under /root/reference) or the third-party kernel the
 * reference dispatches to that it replaces.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer (hipMalloc'd / torch-ROCm storage) unless the
 *    parameter name starts with `h_`.  Tensors are dense, row-major, float32 unless noted;
 *    feature maps are NCHW.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Every call only
 *    enqueues work on that stream and returns; nothing synchronises, allocates or frees.
 *    The caller owns all buffers, including workspaces (sizes via the *_ws_bytes helpers).
 *  - Return value: 0 on success, negative FDET_E* on error (nothing is enqueued then);
 *    fdet_last_error() returns a thread-local message.  Never throws across the boundary.
 *  - No global mutable state: calls from different threads (e.g. the autograd worker) on
 *    different streams are safe.
 */
#ifndef FDET_H
#define FDET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDET_VERSION 100           /* 0.1.0 */
#define FDET_OK 0
#define FDET_EINVAL (-1)           /* bad argument / unsupported shape */
#define FDET_ELAUNCH (-2)          /* hipLaunch failed (message holds hipGetErrorString) */
#define FDET_EWORKSPACE (-3)       /* workspace too small */

int fdet_version(void);
const char* fdet_last_error(void);

/* ---------------------------------------------------------------------------------------
 * Detection math (bit-exact integer/index work, fp32 within 1e-4)
 * ------------------------------------------------------------------------------------- */

/* Grid-cell target encode.  Replaces WIDERFaceDataset.convert_bbx_to_feature_map,
 * datasets/WIDERFace/dataset.py:32-64, batched.
 *   boxes      [total,5] rows [conf,x,y,w,h] (pixel units), images concatenated
 *   box_offset [B+1] int32, image n owns rows box_offset[n] .. box_offset[n+1]-1
 *   out        [B,5,S,S]; fully written (zeros where no box)
 * Semantics kept: map dim1 indexes x; offsets use the UNCLAMPED cell index; cells are
 * clamped to [0,S-1]; the LAST box written to a cell wins. */
int fdet_encode_targets(const float* boxes, const int32_t* box_offset, int B, int S,
                        float img_w, float img_h, float* out, void* stream);

/* YOLO loss, forward + analytic backward in one pass.  Replaces losses/YoloLoss.py:4-44
 * called per image and summed over the batch (models/ModelMeta.py:173-176) plus autograd.
 *   pred, gt       [B,5,S,S]
 *   loss_per_image [B]   (written)
 *   loss_sum       [1]   (written; fixed-order sum of loss_per_image, deterministic)
 *   grad_pred      [B,5,S,S] = grad_scale * d loss_sum / d pred, or NULL to skip
 * Keeps: NaN->0.1 when nansum(pred_n)!=0 (:8-9), pred ch1/ch2 swap (:18), weights 3 and 1/S,
 * the x**0.5 backward singularity (0*inf = NaN as autograd produces). */
int fdet_yolo_loss_fwd_bwd(const float* pred, const float* gt, int B, int S,
                           float* loss_per_image, float* loss_sum, float* grad_pred,
                           float grad_scale, void* stream);

/* Threshold + affine decode + xywh->xyxy + round-half-even.  Replaces
 * ReduceBoundingBoxes.scale_batch_bbx_xywh / remove_low_probabilty_bbx /
 * convert_batch_to_xyxy / torch.round, datasets/utils.py:111-126,152-155,162, batched.
 *   maps   [B,5,S,S]
 *   scores [B,S*S]      candidates in row-major (i,j) order of `x[0] > pt` (strict)
 *   boxes  [B,S*S,4]    rounded x1,y1,x2,y2
 *   counts [B] int32    number of candidates per image */
int fdet_decode(const float* maps, int B, int S, float prob_threshold, float img_w, float img_h,
                float* scores, float* boxes, int32_t* counts, void* stream);

/* Greedy NMS, batched, one image per workgroup.  Replaces torchvision.ops.nms 0.11.2
 * (call site datasets/utils.py:164).  Stable descending-score order; fp32 overlap compared
 * against the double threshold; keep indices are returned in visiting order.
 *   boxes [B,Kmax,4] xyxy, scores [B,Kmax], counts [B] (candidates per image, <= Kmax)
 *   keep  [B,Kmax] int32 indices into the image's candidate list, keep_counts [B] int32
 * Kmax <= 4096. */
int fdet_nms(const float* boxes, const float* scores, const int32_t* counts, int B, int Kmax,
             double iou_threshold, int32_t* keep, int32_t* keep_counts, void* stream);

/* Fused ReduceBoundingBoxes.forward (datasets/utils.py:157-170) for a batch of maps:
 * decode -> NMS -> gather -> xyxy->xywh.
 *   out [B,S*S,5] rows [score,x,y,w,h] in keep order, out_counts [B] int32 */
int fdet_reduce_bounding_boxes(const float* maps, int B, int S, float prob_threshold,
                               double iou_threshold, float img_w, float img_h,
                               float* out, int32_t* out_counts, void* stream);

/* Step metrics.  Replaces the per-image block of ModelMeta.step, models/ModelMeta.py:199-214
 * (torchvision.ops.box_iou + counts), given the reduced boxes of targets and predictions.
 *   gt/pred [B,Kmax,5] rows [score,x,y,w,h], gt_counts/pred_counts [B]
 *   per_image [B,3]  (sum IoU, recall, precision) for each image (0 when no prediction)
 *   totals    [3]    fixed-order sums over the batch divided by B (:216-218) */
int fdet_step_metrics(const float* gt, const int32_t* gt_counts, const float* pred,
                      const int32_t* pred_counts, int B, int Kmax, float* per_image,
                      float* totals, void* stream);

/* uint8 -> float32 / 255 (true division).  Replaces `resize(x) / 255.0`,
 * models/PoolResnet.py:95, for inputs already at the model resolution (Resize is then the
 * identity round trip) and `img / 255`, datasets/WIDERFace/dataset.py:146. */
int fdet_u8_to_f32_norm(const uint8_t* in, float* out, size_t n, void* stream);

/* ---------------------------------------------------------------------------------------
 * Optimiser
 * ------------------------------------------------------------------------------------- */

/* Adam on one flat parameter buffer.  Replaces torch.optim._multi_tensor.Adam.step reached
 * through SAMSGD.step, models/ModelMeta.py:12,81 (lr 1e-4, betas .9/.999, eps 1e-8, wd 0).
 * `step` is 1-based.  Hyper-parameters are doubles (Python floats in the reference; bias
 * corrections are formed in double and rounded once).  grad_scale multiplies the gradient
 * first (1.0 normally). */
int fdet_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                   int step, double lr, double beta1, double beta2, double eps, float grad_scale,
                   void* stream);

/* ---------------------------------------------------------------------------------------
 * SSD detection math (next row after the YOLO path; BASELINE.json config 4).  Priors are ordered
 * (scale, i, j); h_patch_sizes is a HOST array (reference: (60,30,15,7) -> P = 4774).
 * ------------------------------------------------------------------------------------- */
int fdet_ssd_num_priors(const int* h_patch_sizes, int nscales);
/* Multi-scale target encode: WIDERFaceDatasetSSD.convert_bbx_to_feature_map per scale + concat
 * (datasets/WIDERFace/dataset_ssd.py:36-76,134-139).  boxes/box_offsets as fdet_encode_targets;
 * out [B,P,5] rows [conf - 0.001*ps, off_x, off_y, w/W, h/H]. */
int fdet_ssd_encode_targets(const float* boxes, const int32_t* box_offsets, int B, const int* h_patch_sizes,
                            int nscales, float img_w, float img_h, float* out, void* stream);
/* ssd_loss(y_hat[:,:,0], y_hat[:,:,1:], y[:,:,0], y[:,:,1:], neg_pos_ratio) and its autograd
 * (losses/SSDLoss.py:25-86, models/ModelMetaSSD.py:127-129): per-image hard negative mining
 * (rank of -log(conf) among the negatives, ties by index), BCE over positives + mined negatives
 * with rounded labels, smooth-L1 over positives, divided by the batch's positive count.
 *   pred/target [B,P,5]; loss [1]; grad [B,P,5] or NULL; mask [B,P] (1 = contributes) or NULL. */
size_t fdet_ssd_loss_ws_bytes(int B);
int fdet_ssd_loss_fwd_bwd(const float* pred, const float* target, int B, int P, int neg_pos_ratio, float* loss,
                          float* grad, uint8_t* mask, void* ws, size_t ws_bytes, void* stream);
/* The same loss split at its only batch-wide quantity, for data-parallel training (the positive count of
 * losses/SSDLoss.py:86 is the count of the WHOLE batch): `parts` leaves the UNSCALED gradient of this rank's shard in
 * grad and the shard's three fp64 sums [BCE, smooth-L1, positive priors] in sums; the caller SUM-all-reduces the three
 * doubles; `finish` writes loss = (sums[1] + sums[0]) / sums[2] and scales grad by 1 / sums[2].  One rank: the pair
 * equals fdet_ssd_loss_fwd_bwd.  ws of `finish`: >= 16 bytes. */
int fdet_ssd_loss_parts(const float* pred, const float* target, int B, int P, int neg_pos_ratio, float* grad, uint8_t* mask,
                        double* sums, void* ws, size_t ws_bytes, void* stream);
int fdet_ssd_loss_finish(const double* sums, float* loss, float* grad, size_t n_grad, void* ws, size_t ws_bytes, void* stream);
/* ReduceSSDBoundingBoxes.forward for a batch (datasets/utils.py:54-92): decode (with_priors: scale by
 * 1/ps and add the cell origin), threshold (strict >), xyxy, round half even, greedy NMS, xywh.
 *   x [B,P,5]; out [B,P,5] rows [score,x,y,w,h] (first out_counts[n] rows valid). */
int fdet_ssd_reduce_bounding_boxes(const float* x, int B, const int* h_patch_sizes, int nscales, int with_priors,
                                   float prob_threshold, double iou_threshold, float img_w, float img_h,
                                   float* out, int32_t* out_counts, void* stream);
/* the same with a caller-supplied prior table: priors [P,4] on the device (ReduceSSDBoundingBoxes(priors=...),
 * datasets/utils.py:31-32: added to x, y, w, h after the 1/ps scaling of x, y), NULL = calculate_priors() */
int fdet_ssd_reduce_bounding_boxes_priors(const float* x, int B, const int* h_patch_sizes, int nscales, int with_priors,
                                          const float* priors, float prob_threshold, double iou_threshold, float img_w,
                                          float img_h, float* out, int32_t* out_counts, void* stream);

/* SSD heads (models/SSD.py:233-245,206-218): z [N,CP,ps,ps] holds Linear(C,5) of one scale in channels 0..4
 * (CP >= 5, extra channels ignored / zeroed); rows prior_start + i*ps + j of y [N,P,5] get
 * [sigmoid(z0), z1/ps + i/ps, z2/ps + j/ps, z3, z4].  _bwd maps d loss/d y back to dz (needs y for the sigmoid). */
int fdet_ssd_head_pack_fwd(const float* z, int N, int CP, int ps, int prior_start, int P, float* y, void* stream);
int fdet_ssd_head_pack_bwd(const float* dy, const float* y, int N, int CP, int ps, int prior_start, int P, float* dz,
                           void* stream);

/* Bilinear Resize fused with the /255 normalisation: replaces `self.resize(x) / 255.0`
 * (models/PoolResnet.py:91,95; models/Resnet.py likewise) and `Resize(...)(x); x / 255.0`
 * (models/BaseModel.py:64-65), i.e. torchvision 0.11.2 transforms.Resize on tensors =
 * F.interpolate(mode="bilinear", align_corners=False), no antialias.
 *   u8 : src [N,C,Hs,Ws] uint8 -> dst [N,C,Hd,Wd] f32 = round_half_even(interp) / 255   (the uint8 round trip)
 *   f32: src [N,C,Hs,Ws] f32   -> dst = interp / divisor                                 (no rounding)
 * Same-size dimensions are the identity. */
int fdet_resize_bilinear_u8_norm(const uint8_t* src, float* dst, int N, int C, int Hs, int Ws,
                                 int Hd, int Wd, void* stream);
int fdet_resize_bilinear_f32_norm(const float* src, float* dst, int N, int C, int Hs, int Ws,
                                  int Hd, int Wd, float divisor, void* stream);

/* ---------------------------------------------------------------------------------------
 * Conv stack (PoolResnet / Resnet): fp32 implicit GEMM on v_mfma_f32_32x32x2_f32
 * ------------------------------------------------------------------------------------- */

/* Repack OIHW 3x3 weights [Cout,Cin,3,3] into the two K-major panels the conv kernels read:
 *   wpk_fwd [Cin*9][CoutP]  k=(ci,tap)          A panel of the forward conv
 *   wpk_bwd [Cout*9][CinP]  k=(co,flipped tap)  A panel of the data-gradient conv
 * CoutP/CinP = channel count rounded up to 32 (zero padded).  Either output may be NULL. */
int fdet_pack_conv3x3_weights(const float* w, int Cout, int Cin, float* wpk_fwd, float* wpk_bwd,
                              void* stream);

/* 3x3 stride-1 pad-1 convolution with fused tail.  Replaces nn.Conv2d + LeakyReLU(0.2)
 * [+ Dropout2d + skip add + MaxPool2d(2)] of ResidualBlock.forward,
 * models/PoolResnet.py:33-43 (models/Resnet.py:30-40).
 *   x [N,Cin,H,W], wpk = wpk_fwd of fdet_pack_conv3x3_weights, bias [Cout]
 *   z = lrelu(conv(x)+bias, slope)
 *   y_full [N,Cout,H,W]  = z                         (NULL to skip; saved for backward)
 *   y_out  [N,Cout,H,W] = z*drop_scale[n,c] + skip      (NULL to skip; un-pooled tail)
 *   skip [N,Cout,H,W] or NULL; drop_scale [N,Cout] or NULL (eval).
 * `pool` must be 1: blocks that pool write y_full here and finish in fdet_block_tail_fwd. */
int fdet_conv3x3_fwd(const float* x, const float* wpk, const float* bias, float* y_full,
                     const float* skip, const float* drop_scale, float* y_out,
                     int N, int Cin, int Cout, int H, int W, int pool, float slope, void* stream);

/* Data gradient of the 3x3 conv with fused tail (autograd of the above):
 *   dx = conv_transpose(dz) * lrelu'(act) + add
 *   dz [N,Cout,H,W], wpk = wpk_bwd, act [N,Cin,H,W] or NULL (lrelu'(a)=1 if a>0 else slope),
 *   add [N,Cin,H,W] or NULL, dx [N,Cin,H,W]. */
int fdet_conv3x3_dgrad(const float* dz, const float* wpk, const float* act, const float* add,
                       float* dx, int N, int Cin, int Cout, int H, int W, float slope, void* stream);

/* bf16x3 variants of the three entry points above: identical arguments and results within ~1e-5
 * (tests: 1e-4), computed on v_mfma_f32_32x32x16_bf16 with every fp32 operand split into two
 * bf16 values (x = hi + lo; products a_hi*b_lo + a_lo*b_hi + a_hi*b_hi, fp32 accumulation).
 * ~5x the MFMA rate of the fp32 path: the convs become HBM-bound.  Channel counts must be
 * multiples of 16.  The packed panels hold [hi | lo] bf16 and are EXACTLY as large as the fp32
 * panels (same buffers can be reused); they are not interchangeable with them. */
int fdet_pack_conv3x3_weights_bf16x3(const float* w, int Cout, int Cin, void* wpk_fwd, void* wpk_bwd,
                                     void* stream);
/* L same-shape layers in one launch; h_* are HOST arrays of L device pointers (either panel array may be NULL). */
int fdet_pack_conv3x3_weights_bf16x3_batched(const float* const* h_w, int L, int Cout, int Cin,
                                             void* const* h_wpk_fwd, void* const* h_wpk_bwd, void* stream);
int fdet_conv3x3_fwd_bf16x3(const float* x, const void* wpk, const float* bias, float* y_full,
                            const float* skip, const float* drop_scale, float* y_out,
                            int N, int Cin, int Cout, int H, int W, int pool, float slope, void* stream);
int fdet_conv3x3_dgrad_bf16x3(const float* dz, const void* wpk, const float* act, const float* add,
                              float* dx, int N, int Cin, int Cout, int H, int W, float slope, void* stream);

/* Weight + bias gradient of the 3x3 conv: dW[co,ci,ky,kx] = sum_{n,y,x} dz*x_shifted,
 * db[co] = sum dz.  Deterministic two-pass (per-workgroup slabs in `ws`, then a fixed-order
 * reduce).  dW [Cout,Cin,3,3], db [Cout] are overwritten. */
size_t fdet_conv3x3_wgrad_ws_bytes(int N, int Cin, int Cout, int H, int W);
int fdet_conv3x3_wgrad(const float* x, const float* dz, float* dW, float* db, void* ws,
                       size_t ws_bytes, int N, int Cin, int Cout, int H, int W, void* stream);

/* bf16x3 variant of the weight gradient (same arguments/results within ~1e-5 of the tensor's scale;
 * see the bf16x3 note above).  Rows of up to 16 vector lanes (VW = 4, 2, 1 floats for W%4==0, W%2==0,
 * else) are staged whole; wider rows need W % 4 == 0 and are cut into 56-column segments.
 * fdet_conv3x3_wgrad_bf16x3_ws_bytes returns 0 when the shape is not supported. */
size_t fdet_conv3x3_wgrad_bf16x3_ws_bytes(int N, int Cin, int Cout, int H, int W);
int fdet_conv3x3_wgrad_bf16x3(const float* x, const float* dz, float* dW, float* db, void* ws,
                              size_t ws_bytes, int N, int Cin, int Cout, int H, int W, void* stream);

/* Batched form: L <= 16 same-shape layers in ONE launch (+ one reduce launch).  h_x / h_dz / h_dW /
 * h_db are HOST arrays of L device pointers.  Small layers (15x15) are launch/reduction-overhead
 * bound one at a time; batched, every workgroup owns bands of a single layer and writes one slab. */
size_t fdet_conv3x3_wgrad_bf16x3_batched_ws_bytes(int L, int N, int Cin, int Cout, int H, int W);
int fdet_conv3x3_wgrad_bf16x3_batched(const float* const* h_x, const float* const* h_dz, float* const* h_dW,
                                      float* const* h_db, int L, void* ws, size_t ws_bytes,
                                      int N, int Cin, int Cout, int H, int W, void* stream);

/* Residual-block CHAIN at one resolution, running activation resident in LDS (bf16x3 arithmetic).
 * `nblocks` consecutive un-pooled ResidualBlocks (models/PoolResnet.py:33-43 / models/Resnet.py:30-40 with
 * pool == 1) in ONE launch, one workgroup per image; HBM only sees the tensors kept for backward.
 * Needs 64 channels and H*roundup4(W+1) <= 256 (15x15, 10x10 ...): fdet_block_chain_supported.
 * All h_* arguments are HOST arrays of `nblocks` DEVICE pointers.
 *   forward : a_k = lrelu(conv1_k(h)); c_k = lrelu(conv2_k(a_k)); h <- c_k*scale_k[n,f] + h
 *     x [N,64,H,W]; h_wpk1/h_wpk2: forward panels of fdet_pack_conv3x3_weights_bf16x3; h_b1/h_b2 biases;
 *     h_scale: [N,64] dropout scales (array or entries NULL = eval); h_a/h_c: where to keep a_k / c_k
 *     (array or entries NULL: not kept); h_out: block outputs (entries may be NULL except the last).
 *   backward: dz2_k = dout*scale_k*lrelu'(c_k); dz1_k = conv2_k^T(dz2_k)*lrelu'(a_k);
 *             dout <- conv1_k^T(dz1_k) + dout, for k = nblocks-1 .. 0; dx = final dout.
 *     h_wpk1b/h_wpk2b: backward panels; h_dz1/h_dz2 [N,64,H,W] are written (operands of the weight gradients). */
int fdet_block_chain_supported(int F, int H, int W);
int fdet_block_chain_fwd_bf16x3(const float* x, const void* const* h_wpk1, const float* const* h_b1,
                                const void* const* h_wpk2, const float* const* h_b2,
                                const float* const* h_scale, float* const* h_a, float* const* h_c,
                                float* const* h_out, int nblocks, int N, int F, int H, int W, float slope,
                                void* stream);
int fdet_block_chain_bwd_bf16x3(const float* dout, const void* const* h_wpk1b, const void* const* h_wpk2b,
                                const float* const* h_scale, const float* const* h_a, const float* const* h_c,
                                float* const* h_dz1, float* const* h_dz2, float* dx, int nblocks, int N, int F,
                                int H, int W, float slope, void* stream);

/* The same chain with the tensors kept per block in the engine's PS layout (csrc/fdet_ps.h; PS image-0 pointers):
 *   forward : h_a_ps[k] (both planes) and h_out_ps[k], k < nblocks-1 (both planes) are operands of the weight gradients,
 *             h_c_ps[k] receives the hi plane only (backward needs the signs of c_k); arrays or entries may be NULL
 *             (not kept).  The last block's output is written to out_last as fp32 NCHW.  x: fp32 NCHW, or PS if x_is_ps.
 *   backward: dout, dx fp32 NCHW; h_dz1_ps / h_dz2_ps are written as PS.
 * No reference counterpart beyond the one of fdet_block_chain_*_bf16x3; values are identical to that flavour's
 * (a PS element is the fp32 value rounded to its bf16 hi + lo parts, which is all the bf16x3 consumers read). */
int fdet_block_chain_fwd_ps(const void* x, int x_is_ps, const void* const* h_wpk1, const float* const* h_b1,
                            const void* const* h_wpk2, const float* const* h_b2, const float* const* h_scale,
                            void* const* h_a_ps, void* const* h_c_ps, void* const* h_out_ps, float* out_last,
                            int nblocks, int N, int F, int H, int W, float slope, void* stream);
int fdet_block_chain_bwd_ps(const float* dout, const void* const* h_wpk1b, const void* const* h_wpk2b,
                            const float* const* h_scale, const void* const* h_a_ps, const void* const* h_c_ps,
                            void* const* h_dz1_ps, void* const* h_dz2_ps, float* dx, int nblocks, int N, int F,
                            int H, int W, float slope, void* stream);

/* Residual-block tail for pooled blocks: out = maxpool_pool(c*drop_scale[n,f] + x)
 * (Dropout2d + skip add + MaxPool2d(2), models/PoolResnet.py:39-42).  pool in {1,2}.
 *   c,x [N,F,H,W]; drop_scale [N,F] or NULL; out [N,F,H/pool,W/pool] (floor: an odd last row/column is dropped). */
int fdet_block_tail_fwd(const float* c, const float* x, const float* drop_scale, float* out,
                        int N, int F, int H, int W, int pool, void* stream);

/* Pooled residual block with the tail fused into the convolutions (bf16x3, maps of even height and even width
 * <= 62; Cout % 32 == 0).  Replaces conv2 + LeakyReLU + Dropout2d + skip + MaxPool2d(2) of
 * models/PoolResnet.py:36-42 (models/Resnet.py:33-39) and their autograd without ever writing c = lrelu(conv2):
 *   forward : out_pooled [N,Cout,H/2,W/2] = maxpool2x2(lrelu(conv(x)+bias) * drop_scale + skip)
 *             route [N,Cout,H/2,W/2] uint8 (NULL in eval): bits 0-3 = (c > 0) of the window's four elements in
 *             ATen scan order (row-major), bits 4-5 = index of the maximum (first maximum wins, NaN is a maximum)
 *   backward: fdet_pool_route_bwd        dz2 [N,F,H,W] = unpool(dout_pooled) * drop_scale * lrelu'(c)
 *             fdet_conv3x3_dgrad_unpool  dx = conv^T(dz) + unpool(dout_pooled)      (conv1's data gradient + skip path)
 * wpk: forward / backward panels of fdet_pack_conv3x3_weights_bf16x3. */
int fdet_conv3x3_pool_fusion_ok(int N, int Cin, int Cout, int H, int W);   /* 1: the two kernels below have a tiling for the shape */
int fdet_conv3x3_fwd_pool_bf16x3(const float* x, const void* wpk, const float* bias, const float* skip,
                                 const float* drop_scale, float* out_pooled, unsigned char* route, int N, int Cin,
                                 int Cout, int H, int W, float slope, void* stream);
int fdet_conv3x3_dgrad_unpool_bf16x3(const float* dz, const void* wpk, const float* dout_pooled,
                                     const unsigned char* route, float* dx, int N, int Cin, int Cout, int H, int W,
                                     float slope, void* stream);
int fdet_pool_route_bwd(const float* dout_pooled, const unsigned char* route, const float* drop_scale, float* dz2,
                        int N, int F, int H, int W, float slope, void* stream);

/* Engine-private PRE-SPLIT ("PS") activations of the bf16x3 conv stack (csrc/fdet_ps.h): a feature map [N,C,H,W]
 * (C % 8 == 0, W <= 62) kept as bf16 hi | lo planes with 8 channels innermost, rows padded to 16/32/64 slots with
 * zero halo slots, zero rows between images -- the layout the MFMA operands have in LDS, so the conv kernels stage
 * it with LDS-DMA and their epilogues write it for the next consumer.  Same bytes per element as fp32.  No
 * counterpart in the reference (its tensors are fp32 NCHW: models/PoolResnet.py:33-43); fp32 NCHW stays the format at
 * every C-ABI boundary that mirrors a reference interface.
 *   fdet_ps_bytes          allocation size in bytes (includes one all-zero guard image on either side; the caller
 *                          zero-fills the allocation ONCE, producers write real elements only); 0 = unsupported shape
 *   fdet_ps_image0_offset  byte offset of image 0 inside the allocation: the pointer every other call takes
 *   fdet_ps_from_f32 / fdet_ps_to_f32   converters (value = float(hi) + float(lo), 16 significant bits) */
size_t fdet_ps_bytes(int N, int C, int H, int W);
size_t fdet_ps_image0_offset(int N, int C, int H, int W);
int fdet_ps_from_f32(const float* x, void* ps, int N, int C, int H, int W, void* stream);
int fdet_ps_to_f32(const void* ps, float* x, int N, int C, int H, int W, void* stream);
/* Column strips (round 4; config 3's 320 / 160 / 80-column maps, models/Resnet.py:30-40): an even map wider than 63 columns is
 * kept as S = fdet_ps_strips(W) strips of <= 62 columns, each a PS image of its own whose edge slots hold the neighbour
 * strip's column.  Every fdet_*_ps entry point takes the FULL width and plans the strips itself; the caller's part is
 *   fdet_ps_halo_exchange(zero_only = 0)  after a producer that writes real elements only (a conv epilogue,
 *                                         fdet_pool_route_bwd_ps) and before a 3x3 conv reads the tensor;
 *   fdet_ps_halo_exchange(zero_only = 1)  before the tensor is the dz operand of fdet_conv3x3_wgrad_ps_batched (a halo slot
 *                                         is not a position of its strip) when its halo slots may hold columns;
 * fdet_ps_from_f32 writes the halos itself.  Both are no-ops on plain (<= 63-column) tensors. */
int fdet_ps_strips(int W);
int fdet_ps_halo_exchange(void* ps, int N, int C, int H, int W, int zero_only, int hi_only, void* stream);
/* 3x3 convs on PS tensors (Cout == 64, Cin % 16 == 0, maps of 15..62 columns, or wider even maps as column strips); wpk: forward / backward panels of
 * fdet_pack_conv3x3_weights_bf16x3.  Same arithmetic as fdet_conv3x3_fwd_bf16x3 / fdet_conv3x3_dgrad_bf16x3
 * (models/PoolResnet.py:33-36 and its autograd):
 *   fwd      : y_ps  = LeakyReLU(conv(x_ps) + bias)
 *   dgrad_act: dx_ps = conv^T(dz_ps) * LeakyReLU'(act_ps) */
int fdet_conv3x3_ps_fwd(const void* x_ps, const void* wpk, const float* bias, void* y_ps, int N, int Cin, int Cout,
                        int H, int W, float slope, void* stream);
int fdet_conv3x3_ps_dgrad_act(const void* dz_ps, const void* wpk, const void* act_ps, void* dx_ps, int N, int Cin,
                              int Cout, int H, int W, float slope, void* stream);

/* Pooled residual block on PS tensors (models/PoolResnet.py:36-42 and its autograd; 64 channels, even maps), the tail
 * fused into the convolutions as in fdet_conv3x3_fwd_pool_bf16x3 / fdet_conv3x3_dgrad_unpool_bf16x3 / fdet_pool_route_bwd:
 *   fwd_pool     : pooled = maxpool2x2(lrelu(conv(x_ps)+bias) * drop_scale + skip_ps) -> pool_ps (PS, may be NULL) and / or
 *                  pool_f32 (fp32 NCHW, may be NULL); route8 [N][8][H/2][W/2][8] uint8 (channel-innermost routing bytes,
 *                  NULL in eval): bits 0-3 = (c > 0) of the window's four elements in scan order, bits 4-5 = argmax
 *                  (first maximum wins, NaN is a maximum)
 *   route_bwd    : dz2_ps = unpool(dout_pooled) * drop_scale * lrelu'(c)        (dout_pooled: fp32 NCHW)
 *   dgrad_unpool : dx (fp32 NCHW) = conv^T(dz_ps) + unpool(dout_pooled) */
int fdet_conv3x3_ps_fwd_pool(const void* x_ps, const void* wpk, const float* bias, const void* skip_ps,
                             const float* drop_scale, void* pool_ps, float* pool_f32, unsigned char* route8, int N, int Cin,
                             int Cout, int H, int W, float slope, void* stream);
int fdet_pool_route_bwd_ps(const float* dout_pooled, const unsigned char* route8, const float* drop_scale, void* dz2_ps,
                           int N, int C, int H, int W, float slope, void* stream);
int fdet_conv3x3_ps_dgrad_unpool(const void* dz_ps, const void* wpk, const float* dout_pooled, const unsigned char* route8,
                                 float* dx, int N, int Cin, int Cout, int H, int W, float slope, void* stream);
/* The PoolResnet stem Conv2d(3,64,10,s8,p2)+bias (models/PoolResnet.py:70-72) writing its output as a PS tensor
 * (image-0 pointer of an (N,64,Ho,Wo) PS allocation): same arithmetic as fdet_stem_fwd_bf16x3. */
int fdet_stem_fwd_ps(const float* x, const float* w, const float* bias, void* y_ps, int N, int Cin, int F, int H, int W,
                     int k, int stride, int pad, void* stream);
/* Weight / bias gradients of L same-shape 64-channel 3x3 layers from PS tensors (h_x[l], h_dz[l]: host arrays of
 * image-0 device pointers): dW[l] [64,64,3,3], db[l] [64] (autograd of models/PoolResnet.py:33-36; same results as
 * fdet_conv3x3_wgrad_bf16x3_batched to the rounding of the PS format).  ws: fdet_conv3x3_wgrad_ps_ws_bytes bytes
 * (0 = unsupported shape).  Deterministic (fixed-order slab reduction). */
size_t fdet_conv3x3_wgrad_ps_ws_bytes(int L, int N, int C, int H, int W);
int fdet_conv3x3_wgrad_ps_batched(const void* const* h_x, const void* const* h_dz, float* const* h_dW, float* const* h_db,
                                  int L, int N, int C, int H, int W, void* ws, size_t ws_bytes, void* stream);

/* Pointwise (1x1) convolution / per-position Linear layer as a dense GEMM on the matrix cores, bf16x3 arithmetic
 * (fp32-level accuracy).  Replaces nn.Conv2d(Cin, Cout, 1) of SeparableResidualBlock.pointwise_conv_skip
 * (models/SSD.py:24-30) and nn.Linear(C, 5) applied at every position (models/SSD.py:183-185) and their autograd.
 * Tensors are [N, C, P] fp32 with P = H*W positions per image (NCHW); any Cin / Cout >= 1.
 *   pack : w [Cout,Cin] -> K-major bf16 hi|lo panels; each buffer fdet_pointwise_packed_bytes(Cout, Cin) bytes
 *   fwd  : y = lrelu_slope(W x + bias)         (slope 1 = identity; bias may be NULL)
 *   dgrad: dx = W^T dz (+ add, may be NULL)
 *   wgrad: dW [Cout,Cin] = sum_{n,p} dz x^T ; db [Cout] = sum dz (db may be NULL); deterministic slab reduction;
 *          ws: fdet_pointwise_wgrad_ws_bytes(N, Cin, Cout, P) bytes of device scratch. */
size_t fdet_pointwise_packed_bytes(int Cout, int Cin);
int fdet_pack_pointwise_weights_bf16x3(const float* w, int Cout, int Cin, void* wpk_fwd, void* wpk_bwd, void* stream);
int fdet_pointwise_fwd_bf16x3(const float* x, const void* wpk_fwd, const float* bias, float* y, int N, int Cin, int Cout,
                              int P, float slope, void* stream);
int fdet_pointwise_dgrad_bf16x3(const float* dz, const void* wpk_bwd, const float* add, float* dx, int N, int Cin,
                                int Cout, int P, void* stream);
size_t fdet_pointwise_wgrad_ws_bytes(int N, int Cin, int Cout, int P);
int fdet_pointwise_wgrad_bf16x3(const float* x, const float* dz, float* dW, float* db, void* ws, size_t ws_bytes, int N,
                                int Cin, int Cout, int P, void* stream);

/* MobileNetV3-small backbone, inference, bf16 (BASELINE.json config 5).  Replaces the forward of
 * models/MobilenetV3Backbone.py:35-60 (timm tf_mobilenetv3_small_100 feature extractor + Conv2d(576,5,3,p1) + sigmoid).
 * Engine-private activations are NHWC bf16 ([N][H][W][C], C % 8 == 0); BatchNorm (eps 1e-3) is folded into the conv
 * weights by the caller.  act: 0 none, 1 ReLU, 2 Hardswish.
 *   fdet_mb_stem      x [N,3,H,W] f32 in [0,1] (x_is_u8: uint8, divided by 255) -> y [N,H/2,W/2,16]; Conv2dSame(3,16,3,s2)
 *                     with folded weights w [16][3][3][3] f32, bias [16], then Hardswish
 *   fdet_mb_depthwise KxK (3|5) depthwise, stride 1 (pad K/2) or 2 (TF "SAME"); w [K*K][C] f32, bias [C]; y [N,Ho,Wo,C];
 *                     pool [N][slots][C] f32 (or NULL), slots = fdet_mb_depthwise_pool_slots(...): per-image partial channel
 *                     sums of y, one row per workgroup column (SqueezeExcite numerators; no atomics: bit-reproducible)
 *   fdet_mb_se_gate   gate [N][C] = hardsigmoid(W2 relu(W1 (sum of the pool rows / HW) + b1) + b2); w1 [R][C], w2 [C][R] f32
 *   fdet_mb_pointwise y [N,P,Cout] = act(W (x * gate) + bias) (+ res); x [N,P,Cin]; w bf16 [ceil32(Cout)][ceil16(Cin)] zero
 *                     padded; bias f32 [ceil32(Cout)]; gate [N][Cin] or NULL; res [N,P,Cout] or NULL
 *   fdet_mb_head      y [N,5,S,S] f32 = sigmoid(Conv2d(C,5,3,p1)(f)); f [N,S,S,C] bf16 (C % 16 == 0); w bf16 [2][64][C]: row
 *                     tap*5 + ch of the fp32 weight (rows 45..63 zero) split into hi = bf16(w) and lo = bf16(w - hi); bias [5]
 *                     f32; ws >= fdet_mb_head_ws_bytes(N, S) (the 45 per-position tap products, f32) */
int fdet_mb_stem(const void* x, int x_is_u8, const float* w, const float* bias, void* y, int N, int H, int W, void* stream);
int fdet_mb_depthwise(const void* x, const float* w, const float* bias, void* y, float* pool, int N, int H, int W, int C,
                      int K, int stride, int act, void* stream);
int fdet_mb_depthwise_pool_slots(int N, int H, int W, int C, int K, int stride);
int fdet_mb_se_gate(const float* pool, int slots, int HW, const float* w1, const float* b1, const float* w2, const float* b2,
                    int N, int C, int R, float* gate, void* stream);
int fdet_mb_pointwise(const void* x, const void* w, const float* bias, const float* gate, const void* res, void* y, int N,
                      int P, int Cin, int Cout, int act, void* stream);
size_t fdet_mb_head_ws_bytes(int N, int S);
int fdet_mb_head(const void* f, const void* w, const float* bias, float* y, void* ws, size_t ws_bytes, int N, int S, int C,
                 void* stream);

/* Backward of the residual-block tail (dropout, skip, max-pool, second LeakyReLU):
 *   e = c*drop_scale + x ; out = maxpool(e) ; given dout [N,F,H/pool,W/pool]:
 *   de = unpool(dout) (first max in window scan order wins, as ATen max_pool2d backward)
 *   dz2 = de * drop_scale * lrelu'(c)
 *   c,x [N,F,H,W]; dz2 [N,F,H,W]; de [N,F,H,W] written only if pool==2 (else de==dout, may be NULL). */
int fdet_block_tail_bwd(const float* dout, const float* c, const float* x, const float* drop_scale,
                        float* dz2, float* de, int N, int F, int H, int W, int pool, float slope,
                        void* stream);

/* Stem convolution (no activation).  Replaces nn.Conv2d(3,F,k,stride,pad) of
 * models/PoolResnet.py:70-76,98 (k10 s8 p2) and models/Resnet.py:64-70 (k3 s2 p1); other
 * (k,stride,pad) return FDET_EINVAL.
 *   x [N,Cin,H,W], w [F,Cin,k,k] (OIHW), bias [F], y [N,F,Ho,Wo].
 * `ws` (fdet_stem_ws_bytes) holds the K-major weight panel the kernel reads (rebuilt per
 * call) and, for the weight gradient, the per-workgroup slabs of the fixed-order reduction. */
size_t fdet_stem_ws_bytes(int N, int Cin, int F, int H, int W, int k, int stride, int pad);
int fdet_stem_fwd(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes,
                  int N, int Cin, int F, int H, int W, int k, int stride, int pad, void* stream);
/* bf16x3 variant of fdet_stem_fwd (PoolResnet geometry only: 3 channels, k10 s8 p2); same
 * arguments, results within ~1e-5.  `ws` is unused. */
int fdet_stem_fwd_bf16x3(const float* x, const float* w, const float* bias, float* y, void* ws, size_t ws_bytes,
                         int N, int Cin, int F, int H, int W, int k, int stride, int pad, void* stream);
/* dW [F,Cin,k,k], db [F] from x and dy [N,F,Ho,Wo] (autograd of the stem; the input image
 * needs no gradient, so there is no data-gradient entry point). */
int fdet_stem_wgrad(const float* x, const float* dy, float* dW, float* db, void* ws, size_t ws_bytes,
                    int N, int Cin, int F, int H, int W, int k, int stride, int pad, void* stream);
/* bf16x3 variant of fdet_stem_wgrad (PoolResnet geometry, W % 16 == 0); same workspace size. */
int fdet_stem_wgrad_bf16x3(const float* x, const float* dy, float* dW, float* db, void* ws, size_t ws_bytes,
                           int N, int Cin, int F, int H, int W, int k, int stride, int pad, void* stream);

/* Head: Dropout2d(0.5) + Conv2d(F,5,k,pad) + Sigmoid.  Replaces models/PoolResnet.py:100-102
 * (k6 p0) and models/Resnet.py:94-96 (k3 p1).
 *   x [N,F,H,W], drop_scale [N,F] or NULL, w [5,F,k,k], bias [5], y [N,5,S,S] (post-sigmoid). */
int fdet_head_fwd(const float* x, const float* drop_scale, const float* w, const float* bias,
                  float* y, int N, int F, int H, int W, int k, int pad, void* stream);
/* Backward of the head given dy = d loss / d y (post-sigmoid):
 *   dx [N,F,H,W] (includes drop_scale), dW [5,F,k,k], db [5]. */
size_t fdet_head_bwd_ws_bytes(int N, int F, int H, int W, int k, int pad);
int fdet_head_bwd(const float* x, const float* drop_scale, const float* w, const float* y,
                  const float* dy, float* dx, float* dW, float* db, void* ws, size_t ws_bytes,
                  int N, int F, int H, int W, int k, int pad, void* stream);

/* precision16 (round 4): the PS kernels with ONE bf16 MFMA pass on the hi planes -- bf16 activations and weights, fp32
 * accumulation, fp32 epilogue arithmetic and fp32 master weights / gradients: the arithmetic of the reference's
 * Trainer(precision=16) (train_model.py:50) with bf16 as the 16-bit type.  Same arguments as the functions without the
 * suffix; the PS tensors they produce hold their hi plane only (the lo plane is left untouched: zero in a buffer that only
 * ever served this mode). */
int fdet_conv3x3_ps_fwd_p16(const void* x_ps, const void* wpk, const float* bias, void* y_ps, int N, int Cin,
                            int Cout, int H, int W, float slope, void* stream);
int fdet_conv3x3_ps_dgrad_act_p16(const void* dz_ps, const void* wpk, const void* act_ps, void* dx_ps, int N,
                                  int Cin, int Cout, int H, int W, float slope, void* stream);
int fdet_conv3x3_ps_fwd_pool_p16(const void* x_ps, const void* wpk, const float* bias, const void* skip_ps,
                                 const float* drop_scale, void* pool_ps, float* pool_f32, unsigned char* route8,
                                 int N, int Cin, int Cout, int H, int W, float slope, void* stream);
int fdet_conv3x3_ps_dgrad_unpool_p16(const void* dz_ps, const void* wpk, const float* dout_pooled,
                                     const unsigned char* route8, float* dx, int N, int Cin, int Cout, int H, int W,
                                     float slope, void* stream);
int fdet_pool_route_bwd_ps_p16(const float* dout_pooled, const unsigned char* route8, const float* drop_scale,
                               void* dz2_ps, int N, int C, int H, int W, float slope, void* stream);
int fdet_stem_fwd_ps_p16(const float* x, const float* w, const float* bias, void* y_ps, int N, int Cin, int F, int H,
                         int W, int k, int stride, int pad, void* stream);
/* Inference: the PoolResnet stem on the uint8 FRAMES themselves: the `x / 255.0` of models/PoolResnet.py:95 (models/BaseModel.py:65)
 * happens in the stem's staging (a 256-entry table of the bf16 hi | lo parts of p / 255), results identical to
 * fdet_u8_to_f32_norm + fdet_stem_fwd_ps, and the fp32 image is never written.  frames: [N][3][H][W] uint8, 4-byte aligned,
 * W % 4 == 0; precision16 != 0: one MFMA pass, hi plane only. */
int fdet_stem_fwd_ps_u8(const unsigned char* frames, const float* w, const float* bias, void* y_ps, int N, int Cin, int F, int H,
                        int W, int k, int stride, int pad, int precision16, void* stream);
int fdet_stem_wgrad_bf16(const float* x, const float* dy, float* dW, float* db, void* ws, size_t ws_bytes,
                         int N, int Cin, int F, int H, int W, int k, int stride, int pad, void* stream);
int fdet_conv3x3_wgrad_ps_batched_p16(const void* const* h_x, const void* const* h_dz, float* const* h_dW,
                                      float* const* h_db, int L, int N, int C, int H, int W, void* ws,
                                      size_t ws_bytes, void* stream);
int fdet_block_chain_fwd_ps_p16(const void* x, int x_is_ps, const void* const* h_wpk1, const float* const* h_b1,
                                const void* const* h_wpk2, const float* const* h_b2, const float* const* h_scale,
                                void* const* h_a_ps, void* const* h_c_ps, void* const* h_out_ps, float* out_last,
                                int nblocks, int N, int F, int H, int W, float slope, void* stream);
int fdet_block_chain_bwd_ps_p16(const float* dout, const void* const* h_wpk1b, const void* const* h_wpk2b,
                                const float* const* h_scale, const void* const* h_a_ps, const void* const* h_c_ps,
                                void* const* h_dz1_ps, void* const* h_dz2_ps, float* dx, int nblocks, int N, int F,
                                int H, int W, float slope, void* stream);

/* MobileNetV3-small backbone, training pieces (round 4; models/MobilenetV3Backbone.py:49-60 trained through
 * models/ModelMeta.py:115-227): fp32 NCHW tensors [N][C][P = H*W].  The 1x1 convs use fdet_pointwise_*_bf16x3, the head
 * fdet_head_fwd / fdet_head_bwd.  PARITY UNPINNED (timm absent): checked against torch autograd on the CPU oracle.
 *   stem     : timm Conv2dSame(3,16,3,stride 2) without bias (TF "SAME" padding), output ceil(H/2) x ceil(W/2); weight gradient
 *   dw       : depthwise conv k = 3 | 5, stride 1 (pad k/2) or 2 (TF "SAME"), no bias; bwd = data + weight gradient
 *   bn       : nn.BatchNorm2d in TRAINING mode (batch statistics over N,H,W; running = (1-momentum)*running + momentum*batch
 *              with the unbiased variance) fused with the activation (0 none, 1 ReLU, 2 Hardswish) and an optional residual
 *              add; bwd takes dy = d/dy and returns dz, dgamma, dbeta.  ws: fdet_mbt_bn_ws_bytes(C)
 *   se       : timm SqueezeExcite y = x * hardsigmoid(W2 relu(W1 mean_hw(x) + b1) + b2); w1 [R,C], w2 [C,R];
 *              pooled [N,C], hidden [N,R], pre [N,C] are kept by the forward for the backward; bwd ws: (2*N*C + N*R) floats */
int fdet_mbt_stem_fwd(const float* x, const float* w, float* z, int N, int H, int W, void* stream);
size_t fdet_mbt_taps_ws_bytes(int C, int k);   /* workspace of the tap-gradient kernels; k = 0: the stem (C = 16, 27 taps) */
int fdet_mbt_stem_wgrad(const float* x, const float* dz, float* dW, void* ws, size_t ws_bytes, int N, int H, int W, void* stream);
int fdet_mbt_dw_fwd(const float* x, const float* w, float* z, int N, int C, int H, int W, int k, int s, void* stream);
int fdet_mbt_dw_bwd(const float* x, const float* dz, const float* w, float* dx, float* dW, void* ws, size_t ws_bytes, int N,
                    int C, int H, int W, int k, int s, void* stream);
size_t fdet_mbt_bn_ws_bytes(int C);
int fdet_mbt_bn_fwd(const float* z, const float* gamma, const float* beta, float* running_mean, float* running_var,
                    float momentum, float eps, float* save_mean, float* save_invstd, const float* residual, float* y,
                    void* ws, size_t ws_bytes, int N, int C, int P, int act, void* stream);
int fdet_mbt_bn_bwd(const float* z, const float* dy, const float* gamma, const float* beta, const float* save_mean,
                    const float* save_invstd, float* dz, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, int N,
                    int C, int P, int act, void* stream);
int fdet_mbt_se_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* pooled,
                    float* hidden, float* pre, float* y, int N, int C, int R, int P, void* stream);
int fdet_mbt_se_bwd(const float* x, const float* dy, const float* pooled, const float* hidden, const float* pre,
                    const float* w1, const float* w2, float* dx, float* dw1, float* db1, float* dw2, float* db2, void* ws,
                    size_t ws_bytes, int N, int C, int R, int P, void* stream);

/* Training head fused with the loss (round 4): ONE launch sequence for what models/PoolResnet.py:100-102 +
 * losses/YoloLoss.py:4-44 (called per image and summed, models/ModelMeta.py:173-176) + their autograd compute:
 *   y = sigmoid(conv(x * drop_scale, w) + bias);  loss_per_image[n] = yolo_loss(y[n], gt[n]);  loss_sum = sum_n;
 *   dx = d loss_sum / d x (includes drop_scale),  dW [5,F,k,k],  db [5].
 * Equivalent to fdet_head_fwd + fdet_yolo_loss_fwd_bwd(grad_scale 1) + fdet_head_bwd; loss_per_image / loss_sum are
 * bit-identical to that sequence GIVEN y, y itself is computed in bf16x3 arithmetic (~1e-5 of the fp32 head).
 * Supported geometry (fdet_head_loss_fused_supported): F = 64, k = 6, pad = 0, maps of at most 15x15 (PoolResnet).
 * ws: fdet_head_loss_fused_ws_bytes() bytes, 16-byte aligned; its LAST 64 bytes are a ticket counter that must be
 * zero at the first call (the kernel leaves it zero); do not share ws between launches that may overlap. */
int fdet_head_loss_fused_supported(int F, int H, int W, int k, int pad);
size_t fdet_head_loss_fused_ws_bytes(int N, int F, int H, int W, int k, int pad);
int fdet_head_loss_fused(const float* x, const float* drop_scale, const float* w, const float* bias,
                         const float* gt, float* y, float* loss_per_image, float* loss_sum, float* dx,
                         float* dW, float* db, void* ws, size_t ws_bytes, int N, int F, int H, int W, int k,
                         int pad, void* stream);

/* Dropout2d scale factors: out[i] = (u_i >= p) ? 1/(1-p) : 0 with u from a counter-based
 * generator keyed by (seed, offset+i).  Replaces nn.Dropout2d's per-(n,c) Bernoulli draw
 * (models/PoolResnet.py:31,69).  The stream differs from ATen's Philox usage, so parity
 * tests inject masks instead. */
int fdet_dropout_scales(float* out, size_t n, float p, uint64_t seed, uint64_t offset, void* stream);

/* Dropout2d scales of every dropout layer of a model in one launch (nn.Dropout2d of models/PoolResnet.py:31,69,
 * models/SSD.py:47; per-rank streams of SURVEY.md 8e).  The counter is indexed by the GLOBAL image number:
 *   counter(image g, layer k, channel c) = base + g*LS + sum_{j<k} channels[j] + c,  LS = sum_k channels[k]
 * so that a data-parallel rank that owns images [first_image, first_image+n) draws exactly the planes a single
 * process draws for the concatenated batch.  channels / p: HOST arrays of nlayers (<= 32) entries.
 * out: layer k is a dense [n][channels[k]] array at float offset n * sum_{j<k} channels[j]. */
int fdet_dropout_scales_layers(float* out, int n, int nlayers, const int* channels, const float* p,
                               uint64_t seed, uint64_t base, uint64_t first_image, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FDET_H */
