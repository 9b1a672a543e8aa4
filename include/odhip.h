/*
 * odhip.h — C ABI of libodhip.so, the MI355X (gfx950) drop-in for the hot path of
 * ak110/object_detector (Darknet53 backbone + multi-scale prior-box head -> decode -> per-class NMS,
 * plus prior-box assignment / loss for training).
 *
 * The reference has NO FFI for this path: everything below `ObjectDetector.predict`
 * (reference voc_validate.py:27, voc_evaluate.py:27) and `od.pb.encode_truth / decode_locs`
 * (reference check_assign.py:21,27) lives in the un-vendored `pytoolkit` submodule.
 * The entry points here are therefore BUILD-DEFINED; each one names the reference call site whose
 * device work it replaces.  INTEGRATION.md shows the ctypes binding a pytoolkit maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative OD_ERR_* otherwise; od_last_error() gives text
 *   - all tensor arguments are raw DEVICE pointers + explicit dims; `stream` is a hipStream_t passed as void*
 *   - no hidden allocation in any per-step call: the caller owns every buffer, including workspaces whose
 *     size is returned by the matching *_workspace_bytes() query.  od_ctx owns only an 8 KiB zero page and an 8 KiB
 *     vector of ones (allocated once in od_ctx_create).
 *   - kernels are asynchronous on `stream`; launch errors are mapped to return codes right after launch
 *   - activations are NHWC; f16 storage, f32 accumulation (MFMA v_mfma_f32_16x16x32_f16)
 */
#ifndef ODHIP_H
#define ODHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OD_OK 0
#define OD_ERR_INVALID (-1)   /* bad argument / unsupported shape */
#define OD_ERR_HIP (-2)       /* HIP runtime error (text in od_last_error) */
#define OD_ERR_WORKSPACE (-3) /* workspace too small */
#define OD_ERR_COMM (-4)      /* RCCL error */

/* activation enum of the fused conv epilogue (backbone leaky(0.1), neck/head ELU) */
#define OD_ACT_LINEAR 0
#define OD_ACT_LEAKY 1
#define OD_ACT_ELU 2

#define OD_RES_NONE 0
#define OD_RES_SAME 1 /* residual tensor has the output's shape: Darknet53 shortcut */
#define OD_RES_UP2 2  /* residual tensor is [B,Ho/2,Wo/2,Cout], nearest-upsampled 2x: FPN top-down add */

#define OD_DT_F16 0
#define OD_DT_F32 1
#define OD_DT_BF16 2 /* od_allreduce payload only */

typedef struct od_ctx od_ctx;

const char* od_last_error(void);
int od_version(void);

int od_ctx_create(int device, od_ctx** out);
int od_ctx_destroy(od_ctx* ctx);

/* ABI self-description, so that a foreign-language binding (ctypes, cgo, JNI ...) can verify its mirror of the structs
 * below against THIS build of the library instead of against a copy of the header: size in bytes of a struct of this
 * header by name ("od_conv_desc", ...), and the byte offset of one of its fields ("od_conv_desc", "tile_cfg").
 * -1 = unknown struct / field.  od_struct_fields writes the comma-separated field names of a struct, in declaration
 * order, into buf (returns the number of fields, -1 unknown struct or buffer too small). */
long od_sizeof(const char* struct_name);
long od_offsetof(const char* struct_name, const char* field_name);
int od_struct_fields(const char* struct_name, char* buf, int buf_bytes);

/* A stream whose kernels run on a subset of the compute units only (bit i of cu_bits[i / 32] = CU i enabled).  The training
 * step (Trainer, OD_TRAIN_WSTREAM_CUS) can confine its weight-gradient chain to one; destroy with od_stream_destroy. */
int od_stream_create_cu_mask(od_ctx* ctx, const uint32_t* cu_bits, int n_words, void** out_stream);
int od_stream_destroy(od_ctx* ctx, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K1/K2: fused conv2d forward.  Replaces the Keras Conv2D+BatchNormalization+activation(+Add) layers
 * that `ObjectDetector.predict` executes (reference voc_validate.py:27; network per docs/MODEL.md:5-21).
 * out = act(scale[c] * conv(x, w)[c] + bias[c]) (+ residual), 'same' padding (ksize/2).
 *   x      f16 [B,H,W,Cin] NHWC, Cin % 8 == 0
 *   w      f16 packed [Cout_pad][Kpad], k = (dy*ksize+dx)*Cin + cin, zero padded
 *          (dims from od_conv_weight_dims; rows beyond Cout and k beyond ksize*ksize*Cin are zero)
 *   scale, bias  f32 [Cout_pad]  (inference: BatchNorm folded; training fwd: identity/conv bias)
 *   res    f16 residual or NULL (res_mode)
 *   out    f16 or f32; element (b, ho, wo, c) is written at
 *          out[b*out_batch_stride + (ho*Wo+wo)*out_pix_stride + c]  (0 strides = dense NHWC)
 * ---------------------------------------------------------------------------------------------- */
typedef struct od_conv_desc {
  const void* x;
  const void* w;
  const float* scale;
  const float* bias;
  const void* res;
  void* out;
  int32_t B, H, W, Cin, Cout;
  int32_t ksize;  /* 1 or 3 */
  int32_t stride; /* 1 or 2 */
  int32_t act;    /* OD_ACT_* */
  float alpha;    /* leaky slope / ELU alpha */
  int32_t res_mode;
  int32_t out_dtype; /* OD_DT_* */
  int64_t out_batch_stride;
  int64_t out_pix_stride;
  int32_t tile_cfg; /* -1 = auto (fastest launch on an idle chip); -2 = auto for launches that overlap other work (batches
                       in flight): least CU x time; otherwise index into the tile-config table (od_conv_num_tile_cfgs) */
  int32_t transposed; /* != 0: backward-data of a 3x3 stride-2 conv: x is [B,H,W,Cin] = dZ, out is [B,2H,2W,Cout];
                         w must be the flipped / channel-swapped pack written by od_pack_weights */
  int32_t splitk;     /* split-K factor for small-M layers: 0 = library decides, 1 = off; needs splitk_workspace */
  void* splitk_workspace; /* f32 scratch for the partial slabs [splitk][B*Ho*Wo*Cout] (NULL = never split) */
  int64_t splitk_workspace_bytes; /* its size; the factor is clamped so that the slabs fit */
  float* bn_partials; /* training forward (NULL otherwise): the epilogue also writes per-channel partial sums of the f16
                         values it stores, one row per m-tile: f32 [rows][2][Cout] = (sum z, sum z^2), rows =
                         od_conv2d_fwd_bn_rows(); od_bn_stats_from_partials turns them into the batch statistics, so the
                         separate pass of od_bn_stats over z is not needed.  Needs out_dtype f16; disables split-K */
  int64_t bn_partials_bytes;
  /* optional (w2 == NULL: none): the pointwise (1x1, stride 1) layer that consumes this launch's output, e.g. the first
   * conv of the next residual block:   out2 = act2(scale2 * conv1x1(out, w2) + bias2),  f16 dense [B,Ho,Wo,Cout2].
   * `out` is written as usual (f16, dense, no transposed / bn_partials).  When the selected kernel holds all channels of
   * a pixel in one workgroup (8-wave kernel, Cout == 256, Cout2 == 128) the second layer runs in its epilogue on the
   * rounded f16 rows -- one launch, `out` is not read back; otherwise the library issues the 1x1 as a second launch on
   * the same stream.  Same rounding points either way.  w2 is packed like w (od_conv_weight_dims(Cout2, Cout, 1)). */
  const void* w2;
  const float* scale2;
  const float* bias2;
  void* out2;
  int32_t Cout2;
  int32_t act2;
  float alpha2;
  int32_t pad2_;
  /* optional (nseg <= 1: none): GROUPED launch -- the same layer (w, scale, bias, activation, B, Cin, Cout, 3x3, stride 1,
   * out_dtype / out_batch_stride / out_pix_stride) applied to nseg <= 3 input maps of different sizes, the prediction
   * module whose weights the pyramid levels share (reference docs/MODEL.md:8): segment i reads seg_x[i] f16
   * [B, seg_H[i], seg_W[i], Cin] and writes seg_out[i]; x / out / H / W are ignored.  No residual, w2, bn_partials or
   * transposed mode.  One launch when the 8-wave kernel takes the layer (every m-tile lies inside one segment), otherwise
   * the library issues the nseg launches itself -- identical results either way. */
  int32_t nseg;
  int32_t pad3_;
  const void* seg_x[3];
  void* seg_out[3];
  int32_t seg_H[3];
  int32_t seg_W[3];
} od_conv_desc;

int od_conv_weight_dims(int cout, int cin, int ksize, int* cout_pad, int* kpad);
int od_conv_num_tile_cfgs(void);
int od_conv2d_fwd(od_ctx* ctx, const od_conv_desc* d, void* stream);
/* number of partial rows a launch of this descriptor writes into bn_partials (depends on the tile the library picks);
 * <= 0 on error */
int od_conv2d_fwd_bn_rows(od_ctx* ctx, const od_conv_desc* d);

/* K1+K2 fused residual block of the early Darknet53 stages (one launch instead of two layers + add):
 *   out = x + act(scale3 * conv3x3(act(scale1 * conv1x1(x) + bias1)) + bias3)
 * x, out f16 [B,H,W,C]; the middle tensor has C/2 channels and never leaves the CU.  w1 / w3 are the packed weights
 * of the 1x1 (C -> C/2) and 3x3 (C/2 -> C) convolutions exactly as od_conv2d_fwd takes them.  Supported: C in {64, 128},
 * H and W multiples of 16 (od_bottleneck_supported); everything else runs as two od_conv2d_fwd calls.
 * Replaces the same Keras layers as od_conv2d_fwd (reference voc_validate.py:27; docs/MODEL.md:15-17). */
typedef struct od_bneck_desc {
  const void* x;
  const void* w1;
  const float* scale1;
  const float* bias1;
  const void* w3;
  const float* scale3;
  const float* bias3;
  void* out;
  int32_t B, H, W, C;
  int32_t act; /* OD_ACT_*, both convolutions */
  float alpha;
} od_bneck_desc;
int od_bottleneck_supported(int H, int W, int C);
int od_bottleneck_fwd(od_ctx* ctx, const od_bneck_desc* d, void* stream);

/* K3+K1 fused stem: the first two Darknet53 layers (uint8 image -> 3x3 conv 3->32 -> 3x3 stride-2 conv 32->64, each with
 * folded BatchNorm + activation) in one launch; the 32-channel full-resolution tensor stays on chip.
 *   x u8 [B,H,W,3]; w0 f16 [32][32] as od_conv_first_fwd takes it; w3 = packed weights of the stride-2 conv as
 *   od_conv2d_fwd takes them; out f16 [B,H/2,W/2,64].  H and W multiples of 32 (od_stem_supported).
 * Replaces the preprocess + first two Conv2D layers of predict (reference voc_validate.py:27; docs/MODEL.md:15-17). */
typedef struct od_stem_desc {
  const uint8_t* x;
  const void* w0;
  const float* scale0;
  const float* bias0;
  const void* w3;
  const float* scale3;
  const float* bias3;
  void* out;
  int32_t B, H, W;
  int32_t act; /* OD_ACT_*, both convolutions */
  float alpha;
} od_stem_desc;
int od_stem_supported(int H, int W);
int od_stem_fwd(od_ctx* ctx, const od_stem_desc* d, void* stream);

/* K3: first layer, uint8 RGB image in, 3x3 stride-1 conv 3->Cout (Cout = 32), input normalisation folded
 * into scale.  x u8 [B,H,W,3]; w f16 packed [Cout][32] (k = (dy*3+dx)*3 + c, k >= 27 zero);
 * out f16 [B,H,W,Cout].  Replaces the image preprocess + first Conv2D of predict (voc_validate.py:27). */
int od_conv_first_fwd(od_ctx* ctx, const uint8_t* x, const void* w, const float* scale, const float* bias,
                      void* out, int B, int H, int W, int Cout, int act, float alpha, void* stream);

/* K4 (standalone form; the planner fuses it into the lateral conv via OD_RES_UP2):
 * out[b,y,x,c] = a[b,y,x,c] + up[b,y/2,x/2,c], all f16 NHWC.  docs/MODEL.md:7 "FPN-like" top-down path. */
int od_upsample2x_add(od_ctx* ctx, const void* a, const void* up, void* out, int B, int H, int W, int C,
                      void* stream);

/* ------------------------------------------------------------------------------------------------
 * K5+K6: head post-process.  pred f32 [B,P,2+NC+4] (col 0/1 = not-object/object logits, 2..2+NC class logits,
 * last 4 = corner-form box offsets: reference check_assign.py:25-27 layout).
 *   conf[b,p,c]  = softmax2(pred[b,p,0:2])[1] * softmax(pred[b,p,2:2+NC])[c]      (docs/MODEL.md:54-58)
 *   boxes[b,p,:] = clip01(priors[p] + loc * loc_scale * [pw,ph,pw,ph])              (od.pb.decode_locs)
 * priors f32 [P,4] corner form, normalised image coordinates.
 * ---------------------------------------------------------------------------------------------- */
int od_head_postprocess(od_ctx* ctx, const float* pred, const float* priors, float* conf, float* boxes,
                        int B, int P, int NC, float loc_scale, int clip, void* stream);

/* K6 alone: od.pb.decode_locs(locs) (reference check_assign.py:27): zero offsets decode to the priors. */
int od_decode_locs(od_ctx* ctx, const float* locs, const float* priors, float* boxes, int N, int P,
                   float loc_scale, int clip, void* stream);

/* K7: per image, the K candidates with the largest (conf desc, flat index asc) among conf > conf_threshold,
 * flat index = p*NC + c.  keys u64 [B,K] = (float_bits(conf) << 32) | (0xFFFFFFFF - flat), unsorted,
 * unused slots 0; counts i32 [B].  Exact radix select: no ties, no approximation. */
size_t od_topk_workspace_bytes(int B, int N, int K);
int od_topk_scores(od_ctx* ctx, const float* conf, int B, int N, int K, float conf_threshold,
                   uint64_t* keys, int32_t* counts, void* workspace, size_t workspace_bytes, void* stream);

/* K8: sort the K keys (bitonic), class-aware greedy NMS with a 64-bit wavefront suppression bitmask
 * (docs/MODEL.md:78-82).  A candidate is suppressed by a kept higher-ranked candidate of the SAME class when
 * inter > iou_threshold * (area_a + area_b - inter), all f32, no FMA contraction.
 *   boxes f32 [B,P,4]; keys u64 [B,K]; counts i32 [B]
 *   keep_flat i32 [B,max_det] (flat index p*NC+c in rank order, -1 padded); keep_count i32 [B]
 *   strict != 0: suppression ignores the class (strict_nms kwarg, voc_validate.py:26; BUILD-DEFINED). K <= 1024. */
size_t od_nms_workspace_bytes(int B, int K);
int od_nms(od_ctx* ctx, const float* boxes, const uint64_t* keys, const int32_t* counts, int B, int P, int NC,
           int K, float iou_threshold, int strict, int max_det, int32_t* keep_flat, int32_t* keep_count,
           void* workspace, size_t workspace_bytes, void* stream);

/* The kept detections of a batch as ONE dense record block, so that `predict` copies a single buffer back per batch
 * (reference voc_validate.py:27 returns per-image classes / confs / bboxes):
 *   out f32 [B][1 + 6*max_det]: out[b][0] = keep_count[b] (int bits); row r = out[b][1 + 6r ...] =
 *   {flat index p*NC+c (int bits; -1 beyond the count), conf[b][flat], boxes[b][p][0..3]} in rank order. */
int od_gather_detections(od_ctx* ctx, const float* conf, const float* boxes, const int32_t* keep_flat,
                         const int32_t* keep_count, int B, int P, int NC, int max_det, float* out, void* stream);

/* K5-K8 as ONE call, the path `predict` runs (reference voc_validate.py:27; docs/MODEL.md:54-58,78-82): pred -> decoded boxes,
 * the exact top-K key set, class-aware NMS kept indices -- five launches, the confidence tensor is never materialised
 * (one pass over pred computes the confidences on chip, their first-digit histogram and one max per prior; the second pass
 * recomputes only the priors that can hold a top-K candidate).  Results are bit-identical to od_head_postprocess ->
 * od_topk_scores -> od_nms on the same pred (keys come back SORTED descending here, unused slots 0).
 *   conf: optional dense f32 [B,P,NC] output (NULL in the product path);  workspace: od_detect_workspace_bytes, zeroed once
 *   with od_detect_workspace_init (every call leaves it ready for the next);  nms_workspace: od_nms_workspace_bytes(B, K).
 *   P even, NC <= 76 (the limit of od_head_postprocess, whose dense form stays available for the same pred), K <= 1024. */
size_t od_detect_workspace_bytes(int B, int P, int NC, int K);
int od_detect_workspace_init(od_ctx* ctx, void* workspace, size_t workspace_bytes, int B, int P, int NC, void* stream);
int od_detect(od_ctx* ctx, const float* pred, const float* priors, int B, int P, int NC, float loc_scale, int clip,
              float conf_threshold, int K, float iou_threshold, int strict, int max_det, float* boxes, float* conf,
              uint64_t* keys, int32_t* counts, int32_t* keep_flat, int32_t* keep_count, void* workspace,
              size_t workspace_bytes, void* nms_workspace, size_t nms_workspace_bytes, void* stream);
/* od_gather_detections without a confidence tensor: the kept detections' confidences are recomputed from pred (same code,
 * same bits). */
int od_gather_detections_pred(od_ctx* ctx, const float* pred, const float* boxes, const int32_t* keep_flat,
                              const int32_t* keep_count, int B, int P, int NC, int max_det, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K9: prior-box assignment + target encoding = od.pb.encode_truth (reference check_assign.py:21,25-27).
 *   priors f32 [P,4]; gt_boxes f32 [B,Gmax,4] corner form, normalised; gt_classes i32 [B,Gmax]; gt_counts i32 [B]
 *   y f32 [B,P,2+NC+4]: col 0 = background, col 1 = assigned ("y[:,1]==1", :25), cols 2..2+NC one-hot class (:26),
 *   last 4 = regression target = inverse of od_decode_locs (:27).  An all-zero row = ignored prior.
 *   assigned_gt i32 [B,P] (may be NULL): GT index, -1 background, -2 ignore;  npos i32 [B] assigned priors per image.
 * Rule [BUILD-DEFINED]: best GT per prior with IoU >= pos_thr; neg_thr <= IoU < pos_thr ignored; every GT force-takes
 * its best prior (ascending GT index, later wins).  Gmax <= 128.
 * ---------------------------------------------------------------------------------------------- */
size_t od_assign_workspace_bytes(int B, int P, int Gmax);
int od_assign_anchors(od_ctx* ctx, const float* priors, const float* gt_boxes, const int32_t* gt_classes,
                      const int32_t* gt_counts, int B, int P, int Gmax, int NC, float pos_thr, float neg_thr,
                      float loc_scale, float* y, int32_t* assigned_gt, int32_t* npos, void* workspace,
                      size_t workspace_bytes, void* stream);

/* K10: loss forward + gradient (reference docs/MODEL.md:33-52): focal(objectness, 2-class softmax) +
 * softmax-CE(classes, assigned priors) + box loss (box_mode 0 smooth-L1 / 1 MSE, assigned priors), each weighted and
 * divided by max(1, #assigned).  pred, y, grad f32 [B,P,2+NC+4]; losses f32 [4] = obj, cls, box, total. */
size_t od_loss_workspace_bytes(int B, int P);
int od_loss_fwd_bwd(od_ctx* ctx, const float* pred, const float* y, float* grad, float* losses, int B, int P, int NC,
                    float focal_alpha, float focal_gamma, int box_mode, float w_obj, float w_cls, float w_box,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K11/K12: training-side kernels (the Keras fit() machinery behind tk.dl.od.ObjectDetector: Conv2D backward,
 * BatchNormalization in training mode, activation backward, SGD with per-layer LR multipliers -- docs/MODEL.md:19-21,
 * 84-90).  Activations / gradients f16 NHWC (gradients loss-scaled), statistics and parameters f32.
 *   backward-data  = od_conv2d_fwd on dZ with the w_bwd pack of od_pack_weights (stride 1), or with
 *                    od_conv_desc.transposed = 1 (stride 2); the epilogue's residual input accumulates gradients
 *   backward-weight= od_conv2d_bwd_weight: dw f32 [Cout_pad][Kpad] += dZ^T . shifted(X)   (atomic f32 adds; zero it first)
 * ---------------------------------------------------------------------------------------------- */
/* backward-data as ONE call (SURVEY.md §8b names it as an export): dx[B, Ho*stride, Wo*stride, Cin] f16
 * (= dx_accumulate + ..., when dx_accumulate != NULL; it may alias dx) from dz [B,Ho,Wo,Cout] f16 and the w_bwd pack of
 * od_pack_weights.  A thin wrapper: it fills an od_conv_desc (x = dz, w = w_bwd, identity scale / zero bias owned by the
 * context, transposed = stride == 2) and runs the SAME implicit-GEMM kernels as od_conv2d_fwd -- there is no separate
 * backward-data kernel, which is why the trainer calls od_conv2d_fwd directly. */
int od_conv2d_bwd_data(od_ctx* ctx, const void* dz, const void* w_bwd, const void* dx_accumulate, void* dx, int B, int Ho,
                       int Wo, int Cin, int Cout, int ksize, int stride, void* stream);
/* inference-time BatchNorm folding on the device: scale = gamma / sqrt(var + eps), bias = beta - mean*scale,
 * f32 [C], bit-identical to the numpy f32 expression the loader (object_detector_amd/weights.py fold_bn) evaluates on the
 * host -- the loader folds once on the host at load time, this export is for callers that keep the statistics on the
 * device (e.g. evaluating during training).  The training-mode forward/backward of BatchNorm are od_bn_stats +
 * od_scale_act and od_bn_bwd below (§8b's od_bn_train_fwd / od_bn_train_bwd). */
int od_bn_fold(od_ctx* ctx, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
               float* scale, float* bias, int C, void* stream);
size_t od_bn_workspace_bytes(long long M, int C);
/* batch statistics of z [M,C] -> mean, rstd, and the fused (scale, shift) = (gamma*rstd, beta - mean*scale);
 * run_mean/run_var (may be NULL) updated with `momentum` */
int od_bn_stats(od_ctx* ctx, const void* z, long long M, int C, const float* gamma, const float* beta, float eps,
                float* mean, float* rstd, float* scale, float* shift, float* run_mean, float* run_var, float momentum,
                void* workspace, size_t workspace_bytes, void* stream);
/* the same statistics from the partial rows a conv epilogue wrote (od_conv_desc.bn_partials): fixed-order sum over the
 * rows, then mean / rstd / scale / shift / running statistics exactly as od_bn_stats computes them */
int od_bn_stats_from_partials(od_ctx* ctx, const float* partials, int rows, long long M, int C, const float* gamma,
                              const float* beta, float eps, float* mean, float* rstd, float* scale, float* shift,
                              float* run_mean, float* run_var, float momentum, void* stream);
/* y = act(scale*z + shift) (+ res / up2(res)), f16 */
int od_scale_act(od_ctx* ctx, const void* z, const float* scale, const float* shift, const void* res, int res_mode,
                 void* y, int B, int H, int W, int C, int act, float alpha, void* stream);
/* dz from dy through activation and (bn != 0) BatchNorm; dgamma/dbeta f32 [C] are ACCUMULATED (shared layers);
 * workspace >= od_bn_workspace_bytes(M,C) + 2*C*4 */
int od_bn_bwd(od_ctx* ctx, const void* z, const void* dy, const float* scale, const float* shift, const float* mean,
              const float* rstd, long long M, int C, int act, float alpha, int bn, float* dgamma, float* dbeta,
              void* dz, void* workspace, size_t workspace_bytes, void* stream);
int od_conv2d_bwd_weight(od_ctx* ctx, const void* x, const void* dz, float* dw, int B, int H, int W, int Cin, int Cout,
                         int ksize, int stride, void* stream);
/* gradient of the nearest-2x upsample add: dup[b,y,x,c] (+)= sum of the 2x2 block of d */
int od_down2_sum_add(od_ctx* ctx, const void* d, void* dup, int B, int Hh, int Wh, int C, int accumulate, void* stream);
int od_add_f16(od_ctx* ctx, void* a, const void* b, long long n, void* stream);
/* loss gradient rows of one pyramid level (f32 [B,P,C]) -> loss-scaled f16 gradient of that level's prediction conv */
int od_pred_grad_to_level(od_ctx* ctx, const float* grad_pred, void* dz, int B, int P, int C, int row_off, int rows,
                          float loss_scale, void* stream);
/* w -= lr*(m = momentum*m + g*inv_loss_scale + weight_decay*w) on a flat f32 segment */
int od_sgd_step(od_ctx* ctx, float* w, float* m, const float* g, long long n, float lr, float momentum,
                float weight_decay, float inv_loss_scale, void* stream);
/* master f32 [Cout][k*k*Cin] -> f16 forward pack [Cout_pad][Kpad] and (w_bwd != NULL) backward-data pack
 * [Cin_pad][Kpad_t] (taps flipped, channels swapped); both destinations must be pre-zeroed once (padding) */
int od_pack_weights(od_ctx* ctx, const float* w, void* w_fwd, void* w_bwd, int Cout, int Cin, int ksize, void* stream);

/* Deterministic weight gradients: od_conv2d_bwd_weight_slabs writes the per-split partial sums as
 * od_conv2d_bwd_weight_splits(...) dense f32 slabs [split][Cout][k*k*Cin] (plain stores, no atomics); one
 * od_wgrad_reduce_multi launch per step then sums every layer's slabs in ascending order into the flat gradient buffer
 * (`table` = DEVICE array; a layer shared by several pyramid levels lists all of their slabs). */
typedef struct od_wgrad_red {
  int64_t dw_offset; /* into the flat gradient buffer */
  int64_t count;     /* Cout * k*k*Cin */
  const float* slabs;
  int32_t nslabs;
  int32_t pad_;
} od_wgrad_red;
int od_conv2d_bwd_weight_splits(od_ctx* ctx, int B, int H, int W, int Cin, int Cout, int ksize, int stride);
int od_conv2d_bwd_weight_slabs(od_ctx* ctx, const void* x, const void* dz, float* slabs, int B, int H, int W, int Cin,
                               int Cout, int ksize, int stride, void* stream);
int od_wgrad_reduce_multi(od_ctx* ctx, const od_wgrad_red* table, int nlayers, float* grads, void* stream);

/* Multi-tensor forms (one launch for the whole parameter set instead of one per tensor -- ~240 launches per step):
 *   od_sgd_step_multi: `segs` is a DEVICE array of od_sgd_seg over one flat f32 parameter / momentum / gradient buffer;
 *   od_pack_weights_multi: `layers` is a DEVICE array of od_pack_layer (w_offset into the same flat buffer). */
typedef struct od_sgd_seg {
  int64_t offset; /* first element of the segment in the flat buffers */
  int64_t count;
  float lr;
  float weight_decay;
} od_sgd_seg;
typedef struct od_pack_layer {
  int64_t w_offset; /* master weights f32 [Cout][k*k*Cin] inside the flat parameter buffer */
  void* w_fwd;      /* f16 [Cout_pad][Kpad] */
  void* w_bwd;      /* f16 [Cin_pad][Kpad_t] or NULL */
  int32_t Cout, Cin, ksize, pad_;
} od_pack_layer;
/* skip_if_nonzero (DEVICE int32, may be NULL): when *skip_if_nonzero != 0 the launch leaves w and m untouched -- the flag
 * od_grad_nonfinite writes, so that one f16 overflow in the loss-scaled backward pass cannot poison the master weights */
int od_sgd_step_multi(od_ctx* ctx, float* w, float* m, const float* g, const od_sgd_seg* segs, int nseg, float momentum,
                      float inv_loss_scale, const int32_t* skip_if_nonzero, void* stream);
/* flag[0] (DEVICE int32) = 1 when any of g[0..n) is Inf / NaN, else 0.  Run it on the flat gradient buffer AFTER the
 * all-reduce: a non-finite value on one rank reaches every rank through the sum, so all ranks skip the same step. */
int od_grad_nonfinite(od_ctx* ctx, const float* g, long long n, int32_t* flag, void* stream);
/* dst[0..n) = src[0..n) when *flag (DEVICE int32) != 0, untouched otherwise.  The trainer keeps a copy of the BatchNorm
 * running statistics taken before forward and restores it with this call when od_grad_nonfinite flagged the step: the
 * skipped step's forward has already folded its (possibly overflowed) batch statistics into them. */
int od_copy_if_nonzero(od_ctx* ctx, float* dst, const float* src, long long n, const int32_t* flag, void* stream);
/* f32 <-> bf16 (round to nearest even) for the bf16 gradient payload of od_allreduce(OD_DT_BF16) */
int od_cast_f32_bf16(od_ctx* ctx, const float* src, void* dst, long long n, void* stream);
int od_cast_bf16_f32(od_ctx* ctx, const void* src, float* dst, long long n, void* stream);
int od_pack_weights_multi(od_ctx* ctx, const float* w, const od_pack_layer* layers, int nlayers, void* stream);

/* first layer's weight gradient: dw f32 [32][27] += dz^T . shifted(x_u8) * in_scale (no dX: the input is the image).
 * Streaming kernel (W a multiple of 32: the uint8 window is read in place, one 32 x 32 f32 partial slab per workgroup);
 * other widths run the MFMA weight-gradient kernel over an f16 x 8-channel copy of the image kept in `workspace`.  Either way
 * a fixed-order slab sum, no atomics: bit-reproducible.  workspace: 16-byte aligned, caller-owned, sized by
 * od_conv_first_bwd_weight_workspace_bytes (which knows which of the two forms a shape takes). */
size_t od_conv_first_bwd_weight_workspace_bytes(od_ctx* ctx, int B, int H, int W);
int od_conv_first_bwd_weight(od_ctx* ctx, const uint8_t* x, const void* dz, float* dw, int B, int H, int W, int Cout,
                             float in_scale, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K13: data-parallel gradient exchange over RCCL / xGMI (reference knob use_multi_gpu=True, check_assign.py:19).
 * One process per GPU; the launcher (Python) creates the unique id on rank 0 (od_comm_get_unique_id) and ships it to
 * the other ranks itself.  od_allreduce: in-place sum over ranks, asynchronous on `stream`.
 * ---------------------------------------------------------------------------------------------- */
typedef struct od_comm od_comm;
int od_comm_unique_id_bytes(void);
int od_comm_get_unique_id(void* out, int bytes);
int od_comm_init(od_ctx* ctx, int rank, int nranks, const void* unique_id, od_comm** out);
int od_allreduce(od_comm* comm, void* buf, long long count, int dtype, void* stream); /* dtype: OD_DT_F32 / F16 / BF16 */
int od_comm_count(od_comm* comm, int* rank, int* nranks); /* as RCCL reports them (ncclCommUserRank / ncclCommCount) */
int od_comm_destroy(od_comm* comm);

/* ------------------------------------------------------------------------------------------------
 * K14: device-side pixel work of the training generator (reference check_generator.py:17-18; docs/MODEL.md:60-64).
 * Parameters are sampled on the host; one od_aug_params per output image (device array).  src = packed uint8 RGB source
 * images (any sizes, located by src_offset/src_h/src_w); out uint8 [B,H,W,3].
 * crop -> bilinear resize -> flip -> saturation/contrast/brightness -> up to 3 Random-Erasing rectangles.
 * ---------------------------------------------------------------------------------------------- */
typedef struct od_aug_params {
  int64_t src_offset;
  int32_t src_h, src_w;
  float crop_x1, crop_y1, crop_x2, crop_y2;
  int32_t flip;
  float brightness, contrast, saturation;
  int32_t n_erase;
  float erase[3][4];
  uint8_t erase_rgb[3][4];
} od_aug_params;
int od_aug_params_bytes(void);
int od_augment_batch(od_ctx* ctx, const uint8_t* src, const void* params, uint8_t* out, int B, int H, int W,
                     void* stream);

/* Wide (f32) add paths of the mixed-precision inference plan (ObjectDetector(precision="mixed"); replaces the Keras `Add`
 * layers of the residual blocks, reference docs/MODEL.md:15-17, where the reference's fp32 path keeps the sum in fp32):
 *   v = y (+ res);  out32 = v;  out16 = f16(v);  out_hilo[r] = [f16(v) | f16(v - f16(v))]  (row of 2*C halves)
 * y f32 [M,C] is the producing conv's f32 output; res is the f32 (res_f32 != 0) or f16 tensor it is added to -- the residual
 * stream, or with res_up2 the half-size map of the FPN sum (docs/MODEL.md:5-8) -- or NULL;
 * any of the three outputs may be NULL.  A conv over the [hi | lo] tensor with its weights repeated along Cin sees ~22
 * significant bits of v on f16 MFMA operands. */
typedef struct od_wide_desc {
  const float* y;
  const void* res;
  float* out32;
  void* out16;
  void* out_hilo;
  int64_t M;
  int32_t C;
  int32_t res_f32;
  int32_t res_up2; /* != 0: res is [B, H/2, W/2, C] and every element adds its nearest-neighbour parent (the FPN sums) */
  int32_t H, W;    /* the OUTPUT map (res_up2 only; M = B * H * W) */
  int32_t pad_;
} od_wide_desc;
int od_wide_add(od_ctx* ctx, const od_wide_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Native forward plan: the whole layer list of one network executed from C++ (one call per batch, optional
 * hipGraph replay) so Python is not in the per-layer loop.  ops is an array of od_plan_op.
 * ---------------------------------------------------------------------------------------------- */
#define OD_OP_CONV 1
#define OD_OP_CONV_FIRST 2
#define OD_OP_BNECK 3
#define OD_OP_STEM 4
#define OD_OP_WIDE 5

typedef struct od_plan_op {
  int32_t kind; /* OD_OP_* */
  int32_t pad_;
  od_conv_desc conv; /* OD_OP_CONV; OD_OP_CONV_FIRST uses x(u8), w, scale, bias, out, B,H,W,Cout,act,alpha */
  od_bneck_desc bneck; /* OD_OP_BNECK */
  od_stem_desc stem;   /* OD_OP_STEM */
  od_wide_desc wide;   /* OD_OP_WIDE */
} od_plan_op;

typedef struct od_plan od_plan;
int od_plan_create(od_ctx* ctx, const od_plan_op* ops, int n_ops, od_plan** out);
int od_plan_run(od_plan* plan, void* stream);   /* eager launches */
int od_plan_capture(od_plan* plan, void* stream); /* capture into a hipGraph (stream must be capturable) */
int od_plan_replay(od_plan* plan, void* stream);
int od_plan_destroy(od_plan* plan);
/* per-op timing for bench.py: runs the plan once with hipEvents around every op; ms[n_ops] */
int od_plan_time_ops(od_plan* plan, void* stream, float* ms, int n_ops);
/* name of the device kernel an op launches (for matching rocprofv3 rows) */
const char* od_plan_op_kernel_name(od_plan* plan, int op_index);

#ifdef __cplusplus
}
#endif
#endif /* ODHIP_H */
