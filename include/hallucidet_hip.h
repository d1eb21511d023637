/*
 * hallucidet_hip.h -- C ABI of libhallucidet_hip.so (gfx950 / MI355X only).
 *
 * Drop-in boundary for the HalluciDet data-parallel hot path.  Every entry
 * point takes borrowed DEVICE pointers, explicit shapes and a hipStream_t
 * (passed as void*), allocates nothing persistent and returns an int status
 * (0 = ok, <0 = HD_E_*).  No exception and no torch type crosses this line.
 *
 * The reference (heitorrapela/HalluciDet) has no native code of its own: the
 * arithmetic it runs is ATen/cuDNN/torchvision kernels reached from Python.
 * Each entry point therefore cites the *Python call site* whose kernel it
 * replaces (paths relative to the reference tree) and, where that call lands
 * in un-vendored torchvision 0.12, says so with [EXT].
 *
 * Tensor conventions
 *   activations : NHWC, IEEE fp16 ("f16"), channel count a multiple of 8
 *   weights     : [Cout][KH][KW][Cin] f16 (K-contiguous per output channel)
 *   statistics, losses, optimizer state, box maths : fp32
 */
#ifndef HALLUCIDET_HIP_H
#define HALLUCIDET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HD_OK 0
#define HD_E_ARG (-1)     /* bad argument (shape / alignment / null)            */
#define HD_E_LAUNCH (-2)  /* hipLaunch / runtime error, see hd_last_error()      */
#define HD_E_UNSUPPORTED (-3)

#define HD_ACT_NONE 0
#define HD_ACT_RELU 1
#define HD_ACT_SIGMOID 2

#define HD_OUT_NHWC_F16 0
#define HD_OUT_NCHW_F32 1
#define HD_OUT_NHWC_F32 2 /* [N][Ho][Wo][Cout] fp32 (any Cout): head outputs whose [N, HWA, C] view must be free */

/* library identity / diagnostics */
int hd_abi_version(void); /* 6: struct layouts of this header (hd_conv_args incl. y2 = 224 bytes, hd_wgrad_args incl. dw_oihw / dw_scale); a binding
                             must refuse a library that reports another number (hallucidet_amd/_abi.py: ABI_VERSION) */
const char* hd_last_error(void);
const char* hd_arch(void); /* "gfx950" */

/* ------------------------------------------------------------------------
 * Convolution as implicit GEMM on MFMA (v_mfma_f32_32x32x16_f16).
 * Replaces: nn.Conv2d forward inside
 *   src/segmentation_models/base/modules.py:28-35   (Conv2dReLU conv)
 *   src/segmentation_models/decoders/unet/decoder.py:38-46 (upsample+cat+conv)
 *   src/segmentation_models/base/heads.py:23-27      (head conv + activation)
 *   src/segmentation_models/encoders/resnet.py:47-65 (ResNet stages [EXT])
 *   src/utils/eval_forward_fasterrcnn.py:55,76       (detector backbone / RPN head [EXT])
 * and, with flipped/transposed weights, the data-gradient of the same convs.
 *
 * Logical input = concat_c( up1 ? nearest2x(x) : x , x2 ) of size [N,Hin,Win,C1+C2].
 * If in_dil > 1 the logical input is x zero-dilated by in_dil (used for the
 * data-gradient of stride-in_dil convolutions): element (h,w) is x[h/in_dil,w/in_dil]
 * when both are divisible and in range, else 0; Hin/Win are then ignored and the
 * extent is taken from Hsrc/Wsrc.
 * Epilogue: v = acc + bias[co] + res[pix,co]; v = 0 where mask[pix,co] <= 0 (fused ReLU backward);
 * stats on fp16(v); y = act(v).
 * -------------------------------------------------------------------- */
typedef struct hd_conv_args {
  const void* x;      /* f16 NHWC [N,Hsrc,Wsrc,C1]                       */
  const void* x2;     /* f16 NHWC [N,Hin,Win,C2] or NULL                  */
  const void* w;      /* f16 [Cout][KH*KW*(C1+C2)]                        */
  const float* bias;  /* [Cout] or NULL                                   */
  const void* res;    /* f16 NHWC [N,Ho,Wo,Cout] or NULL                  */
  const void* mask;   /* f16 NHWC [N,Ho,Wo,Cout] or NULL                  */
  void* y;            /* out: f16 NHWC [N,Ho,Wo,Cout] or f32 NCHW         */
  float* stats;       /* out: [gridM][2][Cout] per-tile (sum,sumsq) or NULL */
  int32_t N, Hsrc, Wsrc; /* stored extent of x                           */
  int32_t Hin, Win;      /* logical extent (== Hsrc*2 when up1)            */
  int32_t C1, C2;
  int32_t Ho, Wo, Cout;
  int32_t KH, KW, stride, pad;
  int32_t up1, in_dil;
  int32_t act, out_mode;
  /* Consumer-side BatchNorm (both NULL: off).  `x` then holds the RAW output y of the producing convolution and the kernel reads
   * fp16(fma(y, in_scale[c], in_shift[c])) (+ ReLU when in_relu) in its place -- the Conv2dReLU unit's BatchNorm2d + ReLU
   * (src/segmentation_models/base/modules.py:10-47) folded into the operand staging of the NEXT convolution, so that the normalised
   * activation never makes a round trip through HBM; zero padding stays zero.  Bit-identical to hd_bn_apply followed by the plain call.
   * Implemented where the operand passes through registers: the small-channel 3x3 kernel (C1 in {8,16,32}); other shapes -> HD_E_ARG. */
  const float* in_scale; /* [C1] or NULL */
  const float* in_shift; /* [C1] or NULL */
  int32_t in_relu;
  /* out_pool2 = c > 0: the first c output channels leave 2 x 2 SUM-POOLED, y = [N, Ho/2, Wo/2, c]; the remaining Cout - c channels (if
   * any) go unpooled to y2 = [N, Ho, Wo, Cout - c] -- the data gradient of a decoder block's first convolution over
   * cat([nearest_2x(a), skip]) (src/segmentation_models/decoders/unet/decoder.py:38-41): the gradient of `a` is the 2 x 2 sum of the
   * upsampled half's gradient (formed in the epilogue in fp32), the skip's gradient is the other half; the concatenated full-resolution
   * gradient is never written and hd_concat_up_bwd is not needed.  Implemented by the small-channel 3x3 kernel (c == Cout, no y2: a
   * block without a skip), by the 32 -> 128-channel kernel (c == 64, y2 given) and by the 8-wave 3x3 families (c a multiple of 128, or of
   * 64 on the 160-pixel tile; y2 given unless c == Cout); hd_conv2d_pool2_ok says whether a problem
   * qualifies, other problems -> HD_E_ARG. */
  int32_t out_pool2;
  /* Producer-side sums of a BatchNorm's backward pass (bs_y NULL: off).  The tensor this call writes (y, f16 NHWC) is then the incoming
   * gradient dz of a Conv2dReLU unit (src/segmentation_models/base/modules.py:10-47) whose raw convolution output is bs_y, and `stats`
   * receives, per M tile, the rows hd_bn_bwd_reduce would produce from the stored dz: [gridM][2][Cout] = (sum dz*m, sum dz*m*xhat) with
   * xhat = (bs_y - mean) * invstd and the ReLU mask m = (bs_z > 0) when bs_z is given (residual units), else
   * fp16(fma(bs_y, gamma*invstd, beta - mean*gamma*invstd)) > 0 -- the same expressions on the same fp16-rounded dz; only the fp32
   * summation order differs.  One full read of dz and one launch per unit disappear (the rows go to hd_bn_bwd_apply unchanged).
   * Implemented in the 8-wave 3x3 kernels (hd_conv2d_bstat_ok says whether this problem is routed there); excludes act / mask / the
   * forward statistics. */
  const void* bs_y;       /* f16 NHWC [N,Ho,Wo,Cout] or NULL */
  const void* bs_z;       /* f16 NHWC [N,Ho,Wo,Cout] or NULL */
  const float* bs_mean;   /* [Cout] */
  const float* bs_invstd; /* [Cout] */
  const float* bs_gamma;  /* [Cout] or NULL (1) */
  const float* bs_beta;   /* [Cout] or NULL (0) */
  int32_t bs_relu, reserved1;
  void* y2;               /* out_pool2: f16 NHWC [N,Ho,Wo,Cout - out_pool2] or NULL */
} hd_conv_args;

int hd_conv2d(const hd_conv_args* a, void* stream);
/* 1 if hd_conv2d / hd_conv2d_wgrad route this problem to a kernel that implements the bs_* sums (the answer does not depend on the
 * bs_* fields themselves), else 0: the caller then runs hd_bn_bwd_reduce as before */
int hd_conv2d_bstat_ok(const hd_conv_args* a);
/* 1 if hd_conv2d implements out_pool2 for this problem, else 0 */
int hd_conv2d_pool2_ok(const hd_conv_args* a);
/* number of M tiles (rows of `stats`) hd_conv2d will use for this problem */
int hd_conv2d_stats_rows(const hd_conv_args* a);
/* Data gradient of a 7x7 / stride-2 / pad-3 convolution with 64 output and <= 4 input channels (torchvision ResNet.conv1 [EXT] of the
 * frozen detector: the last step of the gradient that trains the hallucination network, src/models/detector.py:24-141) in sub-pixel
 * form: the four output pixels (2I + a, 2J + b) of a low-resolution position are one GEMM row block over the 4 x 4 window
 * dy[I - 1 .. I + 2, J - 1 .. J + 2].  dy: f16 NHWC [N,Hl,Wl,64]; mask_z (or NULL): the stem's output, dy is taken as dy * (mask_z > 0)
 * (the ReLU backward fused into the operand staging); w16: f16 [16][1024], row (2a + b) * 4 + c, column ((di + 1) * 4 + dj + 1) * 64 + ch
 * holds w[ch][a + 3 - 2 di][b + 3 - 2 dj][c] (0 outside the 7 x 7 support and for c = 3); dx: f16 NHWC [N,H,W,8] (channels 3..7 zero).
 * Same products as hd_conv2d's in_dil = 2 route, fp32 sums in another order. */
int hd_conv7x7s2_dgrad_thin(const void* dy, const void* mask_z, const void* w16, void* dx, int N, int Hl, int Wl, int H, int W, void* stream);
/* tuning hook for hd_conv2d's tile choice (tools/tune_conv.py): bm in {64,128}, bn in {32,64,128}, bk in {32,64}, deep in {0,1};
 * -1 = the built-in heuristic.  Process-wide; not for production use. */
int hd_conv_tune_override(int bm, int bn, int bk, int deep);
/* tuning hook of the 8-wave patch-staged 3x3 family (conv3x3_w8.hip, conv3x3_m160.hip): cfg -1 = the built-in cost model, -2 = never,
 * -3 = the cost model without the 160-pixel tile, 10..13 = force tile {256x128, 128x128, 256x64, 128x64} wherever eligible, 15..17 = the
 * step-split main loop of 11..13, 18 / 19 / 20 = force the 160- / 320- / 96-pixel x 64-channel tile (4 x 40, 8 x 40, 4 x 24 pixels,
 * v_mfma_f32_16x16x32_f16; 20: single-source problems).  `nslices` is ignored (kept for the call's shape: the im2col 8-wave
 * family with split-K that used it was measured no faster than the 4-wave kernels on any shape and removed in round 3). */
int hd_conv_tune_w8(int cfg, int nslices);
/* test / tuning hook of the same cost model: n > 0 evaluates it at batch n whatever the launch's batch (then image i of a batched launch is
 * bit-identical to the same image launched alone: the tiles split K differently and do not round identically); 0 (default) = the launch's
 * own batch -- a given problem always gets the same tile (run-to-run identical), differently batched launches agree to fp16 rounding. */
int hd_conv_nominal_batch(int n);
/* test / tuning hook of the large-tile GEMM path (gemm_w8.hip: the plain-GEMM problems of hd_conv2d -- 1x1 / stride-1 convolutions
 * and fully connected layers -- on 256 x 128 (8 waves) / 128 x 128 (4 waves) tiles, register-only epilogue; bit-identical to the
 * implicit-GEMM family): -1 = the built-in rule, 0 = never, 128 / 1128 = that tile wherever the problem is eligible.  Process-wide. */
int hd_gemm_w8_mode(int mode);

/* ------------------------------------------------------------------------
 * Weight-gradient implicit GEMM: dW[co][kh][kw][ci] = sum_pix dY[pix,co] * X[pix@(kh,kw),ci]
 * Replaces the autograd weight-gradient of every trainable Conv2d of the
 * hallucination net (train_hallucidet.py:431-435 holds only those params).
 * Split over `nsplit` pixel ranges; partials go to `slab` [nsplit][Cout][K]
 * fp32 and are reduced by hd_wgrad_reduce into the OIHW fp32 gradient.
 * -------------------------------------------------------------------- */
typedef struct hd_wgrad_args {
  const void* x;   /* f16 NHWC [N,Hsrc,Wsrc,C1]                          */
  const void* x2;  /* f16 NHWC [N,Hin,Win,C2] or NULL                     */
  const void* dy;  /* f16 NHWC [N,Ho,Wo,Cout]                             */
  float* slab;     /* [nsplit][Cout][KH*KW*(C1+C2)] fp32                 */
  int32_t N, Hsrc, Wsrc, Hin, Win, C1, C2, Ho, Wo, Cout;
  int32_t KH, KW, stride, pad, up1;
  int32_t nsplit;
  /* consumer-side BatchNorm of the x operand, as in hd_conv_args (small-channel 3x3 kernel only: C1 in {16,32}, Cout <= 32) */
  const float* in_scale; /* [C1] or NULL */
  const float* in_shift; /* [C1] or NULL */
  int32_t in_relu, reserved0;
  /* direct output (round 5, appended): with nsplit == 1 the one partial IS the gradient, so where the 8-wave 3x3 kernel takes the problem
   * (hd_wgrad_direct_ok) it writes dw_oihw[co][ci][kh][kw] = dw_scale * sum itself -- what hd_wgrad_reduce(slab, dw_oihw, 1, ..., dw_scale,
   * accumulate = 0) would produce, bit for bit -- and `slab` is neither written nor needed (may be NULL).  NULL: the slab, as before. */
  float* dw_oihw;
  float dw_scale;
  int32_t reserved1;
} hd_wgrad_args;
int hd_wgrad(const hd_wgrad_args* a, void* stream);
/* n independent weight gradients (the Conv2d layers of one ResNet stage / one decoder block in the backward pass of
 * train_hallucidet.py:448-451 -> autograd) as ONE grid when every entry runs in the 8-wave patch-staged 3x3 kernel, n hd_wgrad launches
 * otherwise.  Results are bit-identical to n hd_wgrad calls with the same nsplit.  With the layers of a stage in one grid, nsplit = 1
 * (one block per 64 x 64 weight tile and layer) already fills the chip: 1 / 16 ... 1 / 256 of the fp32 slab bytes of per-layer launches.
 * n <= HD_WGRAD_MULTI_MAX for the single grid. */
/* 1 if hd_wgrad / hd_wgrad_multi honour args->dw_oihw for this problem (8-wave 3x3 kernel, nsplit == 1, one source or both of a concat) */
int hd_wgrad_direct_ok(const hd_wgrad_args* a);
#define HD_WGRAD_MULTI_MAX 24
int hd_wgrad_multi(const hd_wgrad_args* args, int n, void* stream);
/* blocks per pixel slice the 8-wave 3x3 weight-gradient kernel uses for this problem ((Cin/64) * (Cout/64)), 0 if hd_wgrad
 * will not route it there: lets the caller choose `nsplit` so that nsplit * blocks fills the GPU */
int hd_wgrad_w8_blocks(const hd_wgrad_args* a);
/* tuning hook (tools/tune_wgrad.py): force hd_wgrad's Cout tile (32 / 64 / 128 rows); -1 = by channel count. Process-wide. */
int hd_wgrad_tune_override(int tm);
/* n independent hd_conv2d problems (an array of argument blocks) as ONE grid where they all run in the same 4-wave implicit-GEMM
 * variant, n separate launches otherwise: the same Conv2d applied per feature level -- torchvision FeaturePyramidNetwork inner_blocks /
 * layer_blocks, RPNHead.conv / cls_logits / bbox_pred, the RetinaNet / FCOS towers [EXT], called per level from
 * src/utils/eval_forward_*.py via model.backbone / model.rpn.head / model.head -- and their data gradients.  Where that is not the case, the
 * members the tile model sends to the 4 x 24-pixel tile (the small pyramid levels) still share one grid.  Results are bit-identical to
 * n hd_conv2d calls.  n <= 10 for a single grid. */
int hd_conv2d_multi(const hd_conv_args* args, int n, void* stream);

/* The data gradient (an hd_conv2d over dY with the flipped weights) and the weight gradient of ONE layer -- the two consumers of the
 * same dY in the backward pass of every trainable Conv2d (train_hallucidet.py:448-451 -> autograd) -- issued together: one grid when
 * both run in the 8-wave kernels (the conv tiles first, the weight-gradient blocks behind them: the 160-tile data gradients of the deep
 * layers leave 96 of the 256 CUs idle on their own), two launches otherwise.  Results are bit-identical to hd_conv2d + hd_wgrad. */
int hd_conv2d_wgrad(const hd_conv_args* dgrad, const hd_wgrad_args* wgrad, void* stream);

/* dw_oihw[co][ci][kh][kw] (=|+=) scale * sum_s slab[s][co][(kh,kw,ci)] ; Cin_real <= Cin, Cout <= Cout_slab
 * (slab rows/channels beyond the real extents are layout padding and are dropped) */
int hd_wgrad_reduce(const float* slab, float* dw_oihw, int nsplit, int Cout_slab, int Cout, int KH, int KW,
                    int Cin, int Cin_real, float scale, int accumulate, void* stream);

/* Every weight tensor of a backward segment reduced in ONE launch (src/segmentation_models: one hd_wgrad slab per conv of
 * decoder / layer4 / ... / stem, 47 per training step).  Entries mean what hd_wgrad_reduce's arguments mean; `mode` and
 * `first_block` are filled by hd_wgrad_reduce_plan (host side, no GPU work), which returns the grid size.  The table lives in
 * HOST memory (n <= HD_WRED_MAX entries) and travels in the launch's kernel arguments.  Each tensor is summed in the same
 * order as hd_wgrad_reduce would sum it alone: results are bit-identical. */
#define HD_WRED_MAX 16
typedef struct hd_wred_desc {
  const float* slab;
  float* dw_oihw;
  int32_t nsplit, Cout_slab, Cout, KH, KW, Cin, Cin_real, accumulate;
  float scale;
  int32_t mode, first_block, reserved;
} hd_wred_desc;
int hd_wgrad_reduce_plan(hd_wred_desc* table_host, int n);      /* -> total blocks (> 0) or a negative status */
int hd_wgrad_reduce_multi(const hd_wred_desc* table_host, int n, int total_blocks, void* stream);

/* one descriptor per layer for hd_weight_prep_multi (device-resident table; same meaning as hd_weight_prep's arguments,
 * no output scale) */
typedef struct hd_wprep_desc {
  const float* w_oihw;
  void* w_fwd;    /* f16 [Cout][KH][KW][Cin_pad] or NULL */
  void* w_dgrad;  /* f16 [Cin_pad][KH][KW][Cout_pad] flipped, or NULL */
  int32_t Cout, Cin, KH, KW, Cin_pad, Cout_pad;
} hd_wprep_desc;

/* ------------------------------------------------------------------------
 * Weight preparation: fp32 OIHW master -> f16 OHWI forward layout (+ optional
 * per-output-channel scale fold and channel padding) and the flipped/transposed
 * layout the data-gradient conv consumes.
 * -------------------------------------------------------------------- */
int hd_weight_prep(const float* w_oihw, const float* out_scale /*[Cout] or NULL*/, void* w_fwd /*f16 [Cout][KH][KW][Cin_pad] or NULL*/,
                   void* w_dgrad /*f16 [Cin_pad][KH][KW][Cout_pad] flipped, or NULL*/, int Cout, int Cin, int KH, int KW,
                   int Cin_pad, int Cout_pad, void* stream);
/* every trainable conv of the hallucination net re-packed in ONE launch (the fp32 masters move every optimizer step:
 * train_hallucidet.py:431-435); table_dev: n_layers descriptors in device memory; grid = (blocks_per_layer, n_layers) */
int hd_weight_prep_multi(const hd_wprep_desc* table_dev, int n_layers, int blocks_per_layer, void* stream);

/* ------------------------------------------------------------------------
 * BatchNorm2d, training mode (batch statistics), split in the three steps the
 * fused pipeline needs.  Replaces nn.BatchNorm2d inside Conv2dReLU
 * (src/segmentation_models/base/modules.py:41-42) and torchvision BasicBlock [EXT].
 * -------------------------------------------------------------------- */
/* deterministic column sum of a [rows][W] fp32 slab -> out[W]; ws: >= 128*W floats (needed when rows > 32) */
int hd_colsum(const float* in, int rows, int W, float* out, float* ws, void* stream);
/* one deterministic reduction stage: out[r][W] = sum of the r-th slice of in's rows (out_rows <= rows) */
int hd_rowsum(const float* in, int rows, int W, float* out, int out_rows, void* stream);
/* part[rows][2][C] partial (sum x, sum x^2) rows (any number of rows, summed in-kernel in a fixed order) -> mean/invstd/scale/shift
 * (+ running stats update when running_mean != NULL; unbiased variance for the running estimate, momentum as
 * nn.BatchNorm2d) */
int hd_bn_finalize(const float* part, int rows, int C, double count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                   float* scale, float* shift, void* stream);
/* eval mode: scale/shift from running statistics */
int hd_bn_eval_scale_shift(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                           float eps, int C, float* scale, float* shift, void* stream);
/* z = act(y*scale[c] + shift[c] (+ res)) , all f16 NHWC, n = number of elements (multiple of 8) */
int hd_bn_apply(const void* y, const void* res, const float* scale, const float* shift, void* z, int64_t n, int C,
                int relu, void* stream);
/* backward of z = relu(bn(y) (+ res)):  part[rows][2][C] <- (sum g, sum g*xhat) with g = dz*(z>0).
 * z == NULL (non-residual unit): the mask is recomputed as (f16)(y*gamma*invstd + beta - mean*gamma*invstd) > 0,
 * bit-identical to the forward's activation. */
int hd_bn_bwd_reduce(const void* dz, const void* z, const void* y, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, float* part, int rows, int64_t npix, int C, int relu, void* stream);
/* dy = gamma*invstd*(g - sum_g/M - xhat*sum_gx/M); dres = g (optional); also emits dgamma/dbeta (fp32, scaled by gscale).
 * part[rows][2][C] (any number of rows) are the partial rows of hd_bn_bwd_reduce: a coefficient launch sums them once (fixed
 * order) into coef_ws [5][C] (caller-owned scratch), the apply launch streams the tensor. */
int hd_bn_bwd_apply(const void* dz, const void* z, const void* y, const float* mean, const float* invstd,
                    const float* gamma, const float* beta, const float* part, int rows, float* coef_ws, void* dy, void* dres,
                    float* dgamma, float* dbeta, float gscale, int accumulate, int64_t npix, int C, int relu, void* stream);

/* ------------------------------------------------------------------------
 * Pooling / resampling / layout
 * -------------------------------------------------------------------- */
/* MaxPool2d(3, stride 2, pad 1) NHWC f16 (encoders/resnet.py:51 via torchvision ResNet.maxpool [EXT]) */
int hd_maxpool3x3s2(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, void* stream);
int hd_maxpool3x3s2_bwd(const void* x, const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo, void* stream);
/* same pooling, additionally recording the winning window position (0..8, first maximum in (kh,kw) scan order = ATen's
 * rule) per output element in idx_u8 [N,Ho,Wo,C]; the _bwd_idx form routes dy with it (1 byte + 1 gradient read per
 * window instead of 9 inputs) */
int hd_maxpool3x3s2_idx(const void* x, void* y, void* idx_u8, int N, int H, int W, int C, int Ho, int Wo, void* stream);
int hd_maxpool3x3s2_bwd_idx(const void* idx_u8, const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo, void* stream);
/* the same with a second gradient of the pooled tensor's input added in the same pass: dx = f16(f16(routed) + add), the rounding of
 * hd_maxpool3x3s2_bwd_idx followed by hd_add_f16 (the U-Net's stem output feeds the max-pool AND the last decoder skip,
 * src/segmentation_models/encoders/resnet.py:50-51 + decoders/unet/decoder.py:37-41) */
int hd_maxpool3x3s2_bwd_idx_add(const void* idx_u8, const void* dy, const void* add, void* dx, int N, int H, int W, int C, int Ho, int Wo,
                                void* stream);
/* generic strided-subsample (LastLevelMaxPool k=1,s=2 [EXT]) */
int hd_subsample2(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, void* stream);
int hd_subsample2_bwd(const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo, int accumulate, void* stream);
/* NCHW f32 [N,Cr,H,W] -> NHWC f16 [N,Ho,Wo,Cp] with nearest resize src=floor(dst*in/out)
 * (custom_generalized_transform.py:80-87: F.interpolate default mode) ; channels >= Cr zero-filled.
 * in_scale multiplies the value (normalize (x-0)/1 is identity: custom_generalized_transform.py:177-186). */
int hd_nchw_to_nhwc_resize(const float* x, void* y, int N, int Cr, int H, int W, int Ho, int Wo, int Cp, void* stream);
/* the same with explicit image / channel strides of x (in elements; rows stay dense).  cstride == 0 broadcasts plane 0 to every
 * channel: the reference's 1 -> 3 channel repeat of the IR image (src/utils/utils.py:52-53, train_hallucidet.py:171) folded into
 * the layout conversion, so that the repeated tensor is never materialised */
int hd_nchw_to_nhwc_resize_strided(const float* x, int64_t nstride, int64_t cstride, void* y, int N, int Cr, int H, int W, int Ho,
                                   int Wo, int Cp, void* stream);
/* backward of the above: dx NCHW f32 (zero where no destination pixel selects the source) */
int hd_nchw_to_nhwc_resize_bwd(const void* dy, float* dx, int N, int Cr, int H, int W, int Ho, int Wo, int Cp,
                               float gscale, void* stream);
/* NHWC f16 [N,H,W,Cp] -> NCHW f32 [N,Cr,H,W] */
int hd_nhwc_to_nchw(const void* x, float* y, int N, int Cr, int H, int W, int Cp, void* stream);
/* y = a + nearest_resize(b -> a's size)  (FPN top-down path, torchvision FeaturePyramidNetwork [EXT]) */
int hd_upsample_add(const void* a, const void* b, void* y, int N, int H, int W, int C, int Hb, int Wb, void* stream);
/* db[n,hb,wb,c] (+)= sum over (h,w) mapping to (hb,wb) of dy */
int hd_upsample_add_bwd(const void* dy, void* db, int N, int H, int W, int C, int Hb, int Wb, int accumulate, void* stream);
/* 2x2 sum-pool of the gradient of a nearest-2x upsample: dx_low[n,h,w,c] (+)= sum dy_up[n,2h+i,2w+j, c_off + c] */
int hd_upsample2_bwd(const void* dy_up, void* dx_low, int N, int Hl, int Wl, int C, int Ctot, int c_off, int accumulate,
                     void* stream);
/* gradient of cat([nearest_2x(a), skip], channel) (decoders/unet/decoder.py:37-41) in one launch: dlow [N,Hl,Wl,Cup] = 2x2 sum-pool of
 * dcat[..., :Cup] (hd_upsample2_bwd's arithmetic), dskip [N,2Hl,2Wl,Cskip] = dcat[..., Cup:] (dskip may be NULL when Cskip == 0) */
int hd_concat_up_bwd(const void* dcat, void* dlow, void* dskip, int N, int Hl, int Wl, int Cup, int Cskip, void* stream);
/* out = a + b ; out = copy channel slice ; all f16, vectors of 8 */
int hd_add_f16(const void* a, const void* b, void* out, int64_t n, void* stream);
int hd_slice_channels(const void* x, void* y, int64_t npix, int Ctot, int c_off, int C, int accumulate, void* stream);
/* elementwise: dlogit = dy * s * (1-s) where s = sigmoid output (NCHW f32 both) -> NHWC f16 [N,H,W,Cp] scaled */
int hd_sigmoid_bwd_nchw_to_nhwc(const float* dy, const float* s, void* dlogit, int N, int Cr, int H, int W, int Cp,
                                float gscale, void* stream);
/* relu mask: dx = dy * (z > 0) */
int hd_relu_bwd(const void* dy, const void* z, void* dx, int64_t n, void* stream);
int hd_f32_to_f16(const float* x, void* y, int64_t n, float scale, void* stream);
int hd_f16_to_f32(const void* x, float* y, int64_t n, float scale, void* stream);
/* x fp32 rows of C channels -> y [P][Cp] fp16 with zero channels C..Cp-1 (gradient of an HD_OUT_NHWC_F32 head output entering a
 * data-gradient conv); row p of x lives at (p / rows_per_image) * image_stride + (p % rows_per_image) * C floats (a slice of a larger
 * buffer; image_stride == rows_per_image * C for a dense source) */
int hd_pad_cast_f32_f16(const float* x, void* y, int64_t P, int C, int Cp, int64_t rows_per_image, int64_t image_stride, void* stream);
/* n <= 16 such problems in one launch (arrays of the scalar call's arguments): the per-level head gradients of an RPN / RetinaNet
 * backward pass */
int hd_pad_cast_f32_f16_multi(const float* const* x, void* const* y, const int64_t* P, const int* C, const int* Cp,
                              const int64_t* rows_per_image, const int64_t* image_stride, int n, void* stream);
/* per-channel sums of an NHWC f16 tensor -> part[rows][C] (bias gradients); reduce with hd_colsum */
int hd_channel_sum_f16(const void* x, int64_t npix, int C, float* part, int rows, void* stream);
/* out[i] (=|+=) in[i]*scale, small fp32 vectors */
int hd_scale_store(const float* in, float* out, int n, float scale, int accumulate, void* stream);

/* ------------------------------------------------------------------------
 * Detection kernels (torchvision.ops [EXT], reached from
 * src/utils/eval_forward_fasterrcnn.py:88,122,136 and eval_forward_retinanet.py:157)
 * -------------------------------------------------------------------- */
/* batched greedy NMS; boxes [B][nmax][4] already sorted by descending score per image, counts[B] (device int32)
 * valid boxes each.  mask workspace: [B][nmax][ceil(nmax/64)] u64.  keep[b][i] = 1/0 in sorted order.
 * IoU test: inter/(a_i+a_j-inter) > thr, fp32, no FMA contraction (torchvision nms [EXT]). */
int hd_nms_sorted_batched(const float* boxes, const int* counts, int B, int nmax, float iou_thr, uint64_t* mask_ws,
                          uint8_t* keep, void* stream);
/* same, for callers that only use the first max_keep survivors of each image (RPN post_nms_top_n, detections_per_img):
 * keep[] equals the full result up to the 64-box chunk in which the max_keep-th survivor falls and is 0 after it. */
int hd_nms_sorted_batched_topk(const float* boxes, const int* counts, int B, int nmax, float iou_thr, uint64_t* mask_ws,
                               uint8_t* keep, int max_keep, void* stream);
/* RoIAlign (aligned=False), NHWC f16 features -> [R][PH][PW][C] f16. rois: [R][5] (batch, x1,y1,x2,y2) fp32 */
int hd_roi_align(const void* feat, const float* rois, void* out, int R, int N, int H, int W, int C, int PH, int PW,
                 float spatial_scale, int sampling_ratio, void* stream);
/* multi-level form (torchvision MultiScaleRoIAlign [EXT]): level[r] in [0,L) selects feats[l] ([N,H[l],W[l],C] f16)
 * and scale[l].  All arrays are HOST arrays of length L <= 4. */
int hd_roi_align_ml(const void* const* feats, const int* H, const int* W, const float* scale, int L, const float* rois,
                    const int* level, void* out, int R, int C, int PH, int PW, int sampling_ratio, void* stream);
int hd_roi_align_ml_bwd(const void* dout, const float* rois, const int* level, float* const* dfeat_f32, const int* H,
                        const int* W, const float* scale, int L, int R, int C, int PH, int PW, int sampling_ratio,
                        void* stream);
/* gather-form backward of hd_roi_align_ml for the 7x7 / sampling_ratio-2 pooler (MultiScaleRoIAlign as built at
 * torchvision faster_rcnn.py [EXT]; reached from src/utils/eval_forward_fasterrcnn.py:120 `box_roi_pool`): every element of
 * dfeat_f16[l][0:n_images] ([.,H[l],W[l],C] f16) is WRITTEN once (no atomics, no zero-fill needed, deterministic);
 * RoIs may come in any order, their image index (rois[r][0]) must be < n_images. */
int hd_roi_align_ml_bwd_gather(const void* dout, const float* rois, const int* level, void* const* dfeat_f16, const int* H,
                               const int* W, const float* scale, int L, int R, int n_images, int C, int PH, int PW,
                               int sampling_ratio, void* stream);
/* backward: dfeat must be zeroed by caller; fp32 atomics into dfeat_f32 [N,H,W,C] */
int hd_roi_align_bwd(const void* dout, const float* rois, float* dfeat_f32, int R, int N, int H, int W, int C, int PH,
                     int PW, float spatial_scale, int sampling_ratio, void* stream);
/* pairwise IoU [G][A] fp32 (torchvision.ops.box_iou [EXT]) */
int hd_box_iou(const float* gt, int G, const float* boxes, int A, float* iou, void* stream);
/* iou[n][g][a] for N images; shared_boxes != 0: one [A][4] box set (anchors) for every image, else boxes is [N][A][4] */
int hd_box_iou_batched(const float* gt, int G, const float* boxes, int A, int N, int shared_boxes, float* iou, void* stream);
/* torchvision.ops.batched_nms (coordinate-offset form [EXT]) over B padded candidate lists, from the unsorted candidates to
 * the ordered list of survivors: boxes [B][n][4], idxs [B][n] i64 (level / class), valid [B][n] u8, order [B][n] i64 =
 * candidate indices by descending score (stable) with the invalid ones last.  pick [B][min(top_n, n)] i64 = candidate
 * index of the k-th surviving box in score order (order[b][0] beyond picked[b]); picked [B] i64.  Workspaces: sorted_ws [B][n][4] f32,
 * counts_ws [B] i32, mask_ws [B][n][ceil(n/64)] u64, keep_ws [B][n] u8.  n <= 16384. */
int hd_batched_nms_pick(const float* boxes, const int64_t* idxs, const uint8_t* valid, const int64_t* order, int B, int n,
                        float iou_thr, int top_n, float* sorted_ws, int* counts_ws, uint64_t* mask_ws, uint8_t* keep_ws,
                        int64_t* pick, int64_t* picked, void* stream);
/* The same selection when the candidates of a row come in SEGMENTS that are the categories of the NMS (the RPN's feature levels:
 * rpn.filter_proposals -> batched_nms(boxes, scores, lvl, ...), torchvision [EXT], reached from src/utils/eval_forward_fasterrcnn.py:86)
 * and are already in descending score order inside each segment (rpn._get_top_n_idx): one independent greedy scan per (row, segment)
 * instead of one per row, then a merge of the segments' survivor lists by (score descending, candidate index ascending).  seg_sizes
 * (HOST) add up to n, <= 8 segments.  Work arrays: S = the largest segment, sorted_ws [B*L*S*4] f32, order_ws / pick_ws [B*L*S] i64,
 * counts_ws [B*L] i32, mask_ws [B*L*S*ceil(S/64)] u64, keep_ws [B*L*S] u8, picked_seg [B*L] i64.  Outputs as hd_batched_nms_pick:
 * pick [B, min(top_n, n)] and picked [B].  Same survivors, same order, same padding, bit for bit. */
int hd_batched_nms_pick_segments(const float* boxes, const float* scores, const uint8_t* valid, int B, int n, const int* seg_sizes, int L,
                                 float iou_thr, int top_n, float* sorted_ws, int64_t* order_ws, int* counts_ws, uint64_t* mask_ws,
                                 uint8_t* keep_ws, int64_t* pick_ws, int64_t* picked_seg, int64_t* pick, int64_t* picked, void* stream);
/* RPN proposals of the selected anchors only (RegionProposalNetwork.filter_proposals [EXT], reached from
 * src/utils/eval_forward_fasterrcnn.py:86): for image n, candidate t, a = top[n][t]:
 *   boxes[n][t] = clip_boxes_to_image(BoxCoder(1,1,1,1).decode(deltas[n][a], anchors[a]), (img_h, img_w));
 *   prob[n][t] = sigmoid(objectness[n][a]);  valid[n][t] = w >= min_size && h >= min_size && prob >= score_thresh.
 * deltas [N][A][4], objectness [N][A], anchors [A][4] (shared by the images), top [N][K] i64. */
int hd_rpn_decode_filter(const float* deltas, const float* objectness, const float* anchors, const int64_t* top, int N, int A, int K,
                         float bbox_xform_clip, float img_h, float img_w, float min_size, float score_thresh, float* boxes,
                         float* prob, uint8_t* valid, void* stream);
/* RoI-head boxes (RoIHeads.postprocess_detections [EXT]): boxes[r][k] = clip(BoxCoder(coder_weights).decode(codes[r][k],
 * rois[r])); codes [R][K*4], rois rows of `roi_stride` floats with the box in the LAST four, coder_weights = host float[4]. */
int hd_roi_decode_clip(const float* codes, const float* rois, long roi_stride, int R, int K, const float* coder_weights,
                       float bbox_xform_clip, float img_h, float img_w, float* boxes, void* stream);
/* RegionProposalNetwork.compute_loss [EXT] (src/utils/eval_forward_fasterrcnn.py:93): objectness [T] f32 logits, deltas /
 * reg_t [T][4], labels [T] f32 (0/1), pos / samp [T] u8 (sampled positives / all sampled).  out2 = (sum_samp BCEWithLogits,
 * sum_pos smooth_l1(beta)) / max(n_sampled, 1); n_sampled = *n_sampled_dev (i64 on the device) when given, else
 * n_sampled_host.  part_ws: 512 floats.  _bwd: gradients w.r.t. objectness [T] and deltas [T][4] for upstream scalars
 * *g_obj, *g_box (device pointers, NULL = 0). */
int hd_rpn_loss(const float* objectness, const float* deltas, const float* labels, const float* reg_t, const uint8_t* pos,
                const uint8_t* samp, int64_t T, float beta, const int64_t* n_sampled_dev, float n_sampled_host, float* part_ws,
                float* out2, void* stream);
int hd_rpn_loss_bwd(const float* objectness, const float* deltas, const float* labels, const float* reg_t, const uint8_t* pos,
                    const uint8_t* samp, int64_t T, float beta, const float* g_obj, const float* g_box,
                    const int64_t* n_sampled_dev, float n_sampled_host, float* d_objectness, float* d_deltas, void* stream);
/* RetinaNet losses for B images sharing one anchor set (reference src/utils/eval_forward_retinanet.py:163-244 with its own
 * sigmoid_focal_loss :22-50 and smooth-L1 box loss :53-80 folded in): cls_logits [B][A][K], bbox_regression [B][A][4] (fp32),
 * matched [B][A] int64 (>= 0 GT index, -1 background, -2 between thresholds = not counted), gt [B][G][4] / glab [B][G] padded
 * targets, anchors [A][4], coder_weights [4] (HOST array: BoxCoder weights).  out2 = {classification, bbox_regression}, each
 * the mean over images of (sum over the image / max(1, #foreground)); num_fg [B] is kept for the backward pass.
 * part_ws: B * 32 * 3 floats. */
int hd_retinanet_loss(const float* cls_logits, const float* bbox_regression, const int64_t* matched, const float* gt, const int64_t* glab,
                      const float* anchors, int B, int A, int K, int G, float alpha, float gamma, float beta, const float* coder_weights,
                      float* part_ws, float* num_fg, float* out2, void* stream);
int hd_retinanet_loss_bwd(const float* cls_logits, const float* bbox_regression, const int64_t* matched, const float* gt, const int64_t* glab,
                          const float* anchors, int B, int A, int K, int G, float alpha, float gamma, float beta, const float* coder_weights,
                          const float* num_fg, const float* g_cls, const float* g_reg, float* d_cls_logits, float* d_bbox_regression,
                          void* stream);
/* element-wise focal loss of the reference's `sigmoid_focal_loss(inputs, targets, alpha, gamma, reduction="none")`
 * (eval_forward_retinanet.py:22-50; targets are 0/1); with grad_out != NULL writes d loss / d inputs * grad_out instead */
int hd_sigmoid_focal_loss(const float* inputs, const float* targets, int64_t n, float alpha, float gamma, const float* grad_out, float* out,
                          void* stream);
/* roi_heads.fastrcnn_loss [EXT] (src/utils/eval_forward_fasterrcnn.py:141): logits [R][K], box_regression [R][K*4],
 * labels [R] i64, reg_t [R][4].  out2 = (mean cross entropy, sum over labels>0 of smooth_l1(beta) of the label's box / R). */
int hd_fastrcnn_loss(const float* logits, const float* box_regression, const int64_t* labels, const float* reg_t, int R, int K,
                     float beta, float* part_ws, float* out2, void* stream);
int hd_fastrcnn_loss_bwd(const float* logits, const float* box_regression, const int64_t* labels, const float* reg_t, int R, int K,
                         float beta, const float* g_cls, const float* g_box, float* d_logits, float* d_box_regression,
                         void* stream);
/* the same two losses over a FIXED-SIZE RoI list (images x batch_size_per_image rows): rows with label < 0 are padding (no loss, no
 * gradient) and the divisor is the number of real rows read from the device (n_valid_dev, one int64) -- nothing is sized on the host */
int hd_fastrcnn_loss_masked(const float* logits, const float* box_regression, const int64_t* labels, const float* reg_t, int R, int K,
                            float beta, const int64_t* n_valid_dev, float* part_ws, float* out2, void* stream);
int hd_fastrcnn_loss_masked_bwd(const float* logits, const float* box_regression, const int64_t* labels, const float* reg_t, int R, int K,
                                float beta, const int64_t* n_valid_dev, const float* g_cls, const float* g_box, float* d_logits,
                                float* d_box_regression, void* stream);
/* BalancedPositiveNegativeSampler [EXT] for N images (reached from src/utils/eval_forward_fasterrcnn.py:90,127): labels [N][A]
 * i64 (>= 1 positive, 0 negative, < 0 ignored), keys [N][A] i32 >= 0 = one random key per candidate.  Per image
 * num_pos = min(#pos, cap_pos), num_neg = min(#neg, batch_size - num_pos); pos_sel / neg_sel [N][A] u8 mark the num_pos /
 * num_neg members with the smallest keys (equal keys at the cut: lowest index first); counts [N][2] i64 = (num_pos, num_neg). */
int hd_sample_pos_neg(const int64_t* labels, const int32_t* keys, int N, int A, int batch_size, int cap_pos, uint8_t* pos_sel,
                      uint8_t* neg_sel, int64_t* counts, void* stream);
/* Tail of RoIHeads.select_training_samples [EXT] (src/utils/eval_forward_fasterrcnn.py:127 in the reference): for the r-th
 * sampled candidate, sel[r] = flat index into the [N][T] candidate arrays: rois[r] = (image, box) [R][5], labels[r] = lab[sel],
 * reg_t[r] = BoxCoder(coder_weights).encode(gt[image][max(matched[sel], 0)] or zeros for an image without GT, box).
 * comb [N*T][4] f32, lab / matched [N*T] i64, gt [N][G][4] f32, gvalid [N][G] u8, coder_weights host float[4]. */
/* RoIHeads.postprocess_detections up to the NMS, for the fixed-size RoI list (row r is real iff r % S < counts[r / S]): softmax over the C
 * class logits, decode + clip of the C - 1 foreground boxes (rois: box in the last four columns of a roi_stride-wide row), candidate
 * test (score > score_thresh, both sides >= min_size) -> boxes [R][C-1][4], scores [R][C-1], valid [R][C-1] u8; padding rows are zero. */
int hd_roi_postprocess(const float* class_logits, const float* box_regression, const float* rois, long roi_stride, const int64_t* counts,
                       int R, int C, int S, const float* coder_weights, float bbox_xform_clip, float img_h, float img_w, float score_thresh,
                       float min_size, float* boxes, float* scores, uint8_t* valid, void* stream);
int hd_roi_samples_finish(const int64_t* sel, int R, const float* comb, const int64_t* lab, const int64_t* matched, const float* gt,
                          const uint8_t* gvalid, int T, int G, const float* coder_weights, float* rois, int64_t* labels, float* reg_t,
                          void* stream);
/* fixed-size form with the compaction included: image n owns rows [n*S, (n+1)*S) of the outputs -- its selected candidates (pos_sel |
 * neg_sel over [N][T], ascending candidate index) first, then padding rows (empty box, label -1, zero target); counts[n] = real rows.
 * Nothing is sized on the host (no synchronisation). */
int hd_roi_samples_padded(const uint8_t* pos_sel, const uint8_t* neg_sel, const float* comb, const int64_t* lab, const int64_t* matched,
                          const float* gt, const uint8_t* gvalid, int N, int T, int G, int S, const float* coder_weights, float* rois,
                          int64_t* labels, float* reg_t, int64_t* counts, void* stream);
/* torchvision.ops.poolers.LevelMapper [EXT]: levels[r] = clamp(floor(canonical_level + log2(sqrt(area_r) / canonical_scale) +
 * eps), k_min, k_max) - k_min; boxes = pointer to x1 of the first box, `stride` floats between boxes. */
int hd_roi_levels(const float* boxes, long stride, int R, float canonical_scale, float canonical_level, float eps, int k_min, int k_max,
                  int* levels, void* stream);
/* Per-row, per-segment top-k (RegionProposalNetwork._get_top_n_idx [EXT]: `ob.topk(pre_nms_top_n, dim=1)` per feature
 * level): scores [B][sum(seg_sizes)] with row stride `row_stride`; for segment s (columns off_s .. off_s+n_s) the row's
 * min(k, n_s) largest entries in descending score order, equal scores by ascending index (= a stable descending sort cut
 * at k), as indices into the ROW; the segments' lists are concatenated in out [B][sum(min(k, n_s))] (row stride
 * out_stride).  seg_sizes is a host array, 1 <= nseg <= 8, min(k, n_s) <= 4096. */
int hd_topk_select_rows(const float* scores, int B, long row_stride, const int* seg_sizes, int nseg, int k, int64_t* out,
                        long out_stride, void* stream);
/* Fused target assignment for N images: box_iou + Matcher(high, low, allow_low_quality) + label lookup + BoxCoder.encode
 * (torchvision RegionProposalNetwork.assign_targets_to_anchors / RoIHeads.assign_targets_to_proposals + box_coder.encode
 * [EXT], called from src/utils/eval_forward_fasterrcnn.py:88-93,127 in the reference).
 *   gt [N][G][4] f32, gvalid [N][G] u8 (padding rows 0), glabels [N][G] i64 or NULL (NULL: every match is class 1),
 *   boxes [A][4] (shared_boxes != 0) or [N][A][4];
 *   matched [N][A] i64: GT index, -1 below `low`, -2 between the thresholds (first maximal GT on ties);
 *   labels  [N][A] i64 or NULL: class of the match, 0 below, -1 between, 0 for images without GT;
 *   reg_t   [N][A][4] f32 or NULL: encode(gt[matched.clamp(0)], box) with coder_weights[4] (both NULL or both set);
 *   best_ws [N*G] f32 workspace, needed when allow_low_quality != 0. */
int hd_match_targets(const float* gt, const uint8_t* gvalid, const int64_t* glabels, int G, const float* boxes, int A, int N,
                     int shared_boxes, float high, float low, int allow_low_quality, const float* coder_weights,
                     float* best_ws, int64_t* matched, int64_t* labels, float* reg_t, void* stream);

/* ------------------------------------------------------------------------
 * FCOS (torchvision.models.detection.fcos [EXT], driven by src/utils/eval_forward_fcos.py:54-83; selected at
 * src/models/detector.py:113-114,135-136)
 * -------------------------------------------------------------------- */
/* GroupNorm with eight channels per group (GroupNorm(32, 256) of FCOSClassificationHead / FCOSRegressionHead.conv) + optional
 * ReLU over NHWC f16 [N][HW][C]; mean_rstd [N][C/8][2] f32 is written for the backward pass.  Replaces nn.GroupNorm + nn.ReLU. */
int hd_groupnorm8_relu(const void* x, const float* gamma, const float* beta, void* y, float* mean_rstd, int N, int HW, int C,
                       float eps, int relu, void* stream);
/* data gradient of the above: dy, x, y (the forward output: ReLU mask; may be NULL when relu == 0) -> dx */
int hd_groupnorm8_relu_bwd(const void* dy, const void* x, const void* y, const float* gamma, const float* mean_rstd, void* dx, int N,
                           int HW, int C, int relu, void* stream);
/* parameter gradients of the above (detector fine-tuning): dgamma[c] (+)= scale * sum g * xhat, dbeta[c] (+)= scale * sum g */
int hd_groupnorm8_param_grad(const void* dy, const void* x, const void* y, const float* mean_rstd, float* dgamma, float* dbeta, int N,
                             int HW, int C, int relu, float scale, int accumulate, void* stream);
/* FCOS.compute_loss target assignment: anchors [A][4] (one stride-sized square per location), gt [B][G][4], gvalid [B][G] u8 ->
 * matched [B][A] i64 (-1 = background): centre sampling (radius x anchor size), location inside the box, scale range
 * (4, 8) x anchor size (lower bound 0 for the first first_level_count locations, no upper bound from last_level_start on),
 * smallest box wins. */
int hd_fcos_match(const float* anchors, const float* gt, const uint8_t* gvalid, int B, int A, int G, int first_level_count,
                  int last_level_start, float center_sampling_radius, int64_t* matched, void* stream);
/* FCOSHead.compute_loss: cls_logits [B][A][K], bbox_regression [B][A][4] (after the head's ReLU), bbox_ctrness [B][A], matched
 * [B][A], gt [B][G][4], glab [B][G] -> out3 = (sigmoid focal, generalized-IoU of the decoded boxes, centre-ness BCE), each summed
 * over the batch and divided by max(1, #foreground) (= num_fg[0], kept for the backward pass).  part_ws: 64 * 4 floats. */
int hd_fcos_loss(const float* cls_logits, const float* bbox_regression, const float* bbox_ctrness, const int64_t* matched, const float* gt,
                 const int64_t* glab, const float* anchors, int B, int A, int K, int G, float alpha, float gamma, float* part_ws,
                 float* num_fg, float* out3, void* stream);
/* g3 = upstream gradients of the three losses (device, 3 floats) */
int hd_fcos_loss_bwd(const float* cls_logits, const float* bbox_regression, const float* bbox_ctrness, const int64_t* matched,
                     const float* gt, const int64_t* glab, const float* anchors, int B, int A, int K, int G, float alpha, float gamma,
                     const float* num_fg, const float* g3, float* d_cls_logits, float* d_bbox_regression, float* d_bbox_ctrness,
                     void* stream);

/* ------------------------------------------------------------------------
 * Optimizer: unscale + clip_grad_value_ + Adam in one pass over a flat buffer
 * (train_hallucidet.py:431-435,498-499; config.py:204-245)
 * -------------------------------------------------------------------- */
int hd_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, float clip_value, float inv_scale, float bias_corr1, float bias_corr2,
                 const float* found_inf /* device flag, step skipped when != 0 */, void* stream);
/* found_inf[0] = 1 if any element of g is inf/nan */
int hd_check_finite(const float* g, int64_t n, float* found_inf, void* stream);

/* ------------------------------------------------------------------------
 * fp32 STORAGE (`--precision 32`, the reference's default: src/config/config.py:149, handed to pl.Trainer at
 * train_hallucidet.py:507, train_detector.py:387, eval_hallucidet.py:230).  Every entry point above that reads or writes an
 * ACTIVATION tensor (documented there as "f16") exists a second time with the suffix _f32: same arguments, same semantics,
 * the activation / weight / gradient tensors stored as fp32 (statistics, losses, box maths and the optimizer are fp32 in
 * both).  A parity mode, not a fast path: one untuned vector-ALU instance of the convolution / data gradient / FC
 * (hd_conv2d_f32), of the weight gradient (hd_wgrad_f32; slabs reduced by hd_wgrad_reduce as before) and of the weight
 * re-pack; the bandwidth-bound kernels are the fp16 sources compiled for fp32 storage.  out_mode 0 and 2 coincide (NHWC
 * fp32).  What it buys: the whole training step agrees with the reference's fp32 evaluation without shared decisions and
 * without a rounding schedule (tests/test_fp32_mode_gpu.py).
 * -------------------------------------------------------------------- */
int hd_conv2d_f32(const hd_conv_args* a, void* stream);
int hd_conv2d_stats_rows_f32(const hd_conv_args* a);
int hd_wgrad_f32(const hd_wgrad_args* a, void* stream);
int hd_weight_prep_f32(const float* w_oihw, const float* out_scale, void* w_fwd, void* w_dgrad, int Cout, int Cin, int KH, int KW,
                       int Cin_pad, int Cout_pad, void* stream);
int hd_bn_apply_f32(const void* y, const void* res, const float* scale, const float* shift, void* z, int64_t n, int C, int relu, void* stream);
int hd_bn_bwd_reduce_f32(const void* dz, const void* z, const void* y, const float* mean, const float* invstd, const float* gamma,
                         const float* beta, float* part, int rows, int64_t npix, int C, int relu, void* stream);
int hd_bn_bwd_apply_f32(const void* dz, const void* z, const void* y, const float* mean, const float* invstd, const float* gamma,
                        const float* beta, const float* part, int rows, float* coef_ws, void* dy, void* dres, float* dgamma, float* dbeta,
                        float gscale, int accumulate, int64_t npix, int C, int relu, void* stream);
int hd_maxpool3x3s2_f32(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, void* stream);
int hd_maxpool3x3s2_bwd_f32(const void* x, const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo, void* stream);
int hd_maxpool3x3s2_idx_f32(const void* x, void* y, void* idx_u8, int N, int H, int W, int C, int Ho, int Wo, void* stream);
int hd_maxpool3x3s2_bwd_idx_f32(const void* idx_u8, const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo, void* stream);
int hd_maxpool3x3s2_bwd_idx_add_f32(const void* idx_u8, const void* dy, const void* add, void* dx, int N, int H, int W, int C, int Ho, int Wo,
                                    void* stream);
int hd_concat_up_bwd_f32(const void* dcat, void* dlow, void* dskip, int N, int Hl, int Wl, int Cup, int Cskip, void* stream);
int hd_subsample2_f32(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, void* stream);
int hd_subsample2_bwd_f32(const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo, int accumulate, void* stream);
int hd_nchw_to_nhwc_resize_f32(const float* x, void* y, int N, int Cr, int H, int W, int Ho, int Wo, int Cp, void* stream);
int hd_nchw_to_nhwc_resize_strided_f32(const float* x, int64_t nstride, int64_t cstride, void* y, int N, int Cr, int H, int W, int Ho,
                                       int Wo, int Cp, void* stream);
int hd_nchw_to_nhwc_resize_bwd_f32(const void* dy, float* dx, int N, int Cr, int H, int W, int Ho, int Wo, int Cp, float gscale, void* stream);
int hd_nhwc_to_nchw_f32(const void* x, float* y, int N, int Cr, int H, int W, int Cp, void* stream);
int hd_upsample_add_f32(const void* a, const void* b, void* y, int N, int H, int W, int C, int Hb, int Wb, void* stream);
int hd_upsample_add_bwd_f32(const void* dy, void* db, int N, int H, int W, int C, int Hb, int Wb, int accumulate, void* stream);
int hd_upsample2_bwd_f32(const void* dy_up, void* dx_low, int N, int Hl, int Wl, int C, int Ctot, int c_off, int accumulate, void* stream);
int hd_add_f16_f32(const void* a, const void* b, void* out, int64_t n, void* stream);
int hd_slice_channels_f32(const void* x, void* y, int64_t npix, int Ctot, int c_off, int C, int accumulate, void* stream);
int hd_sigmoid_bwd_nchw_to_nhwc_f32(const float* dy, const float* s, void* dlogit, int N, int Cr, int H, int W, int Cp, float gscale, void* stream);
int hd_relu_bwd_f32(const void* dy, const void* z, void* dx, int64_t n, void* stream);
int hd_f32_to_f16_f32(const float* x, void* y, int64_t n, float scale, void* stream);
int hd_f16_to_f32_f32(const void* x, float* y, int64_t n, float scale, void* stream);
int hd_pad_cast_f32_f16_f32(const float* x, void* y, int64_t P, int C, int Cp, int64_t rows_per_image, int64_t image_stride, void* stream);
int hd_pad_cast_f32_f16_multi_f32(const float* const* x, void* const* y, const int64_t* P, const int* C, const int* Cp,
                                  const int64_t* rows_per_image, const int64_t* image_stride, int n, void* stream);
int hd_channel_sum_f16_f32(const void* x, int64_t npix, int C, float* part, int rows, void* stream);
int hd_roi_align_f32(const void* feat, const float* rois, void* out, int R, int N, int H, int W, int C, int PH, int PW, float spatial_scale,
                     int sampling_ratio, void* stream);
int hd_roi_align_bwd_f32(const void* dout, const float* rois, float* dfeat_f32, int R, int N, int H, int W, int C, int PH, int PW,
                         float spatial_scale, int sampling_ratio, void* stream);
int hd_roi_align_ml_f32(const void* const* feats, const int* H, const int* W, const float* scale, int L, const float* rois, const int* level,
                        void* out, int R, int C, int PH, int PW, int sampling_ratio, void* stream);
int hd_roi_align_ml_bwd_f32(const void* dout, const float* rois, const int* level, float* const* dfeat_f32, const int* H, const int* W,
                            const float* scale, int L, int R, int C, int PH, int PW, int sampling_ratio, void* stream);
int hd_roi_align_ml_bwd_gather_f32(const void* dout, const float* rois, const int* level, void* const* dfeat_f16, const int* H, const int* W,
                                   const float* scale, int L, int R, int n_images, int C, int PH, int PW, int sampling_ratio, void* stream);
int hd_groupnorm8_relu_f32(const void* x, const float* gamma, const float* beta, void* y, float* mean_rstd, int N, int HW, int C, float eps,
                           int relu, void* stream);
int hd_groupnorm8_relu_bwd_f32(const void* dy, const void* x, const void* y, const float* gamma, const float* mean_rstd, void* dx, int N,
                               int HW, int C, int relu, void* stream);
int hd_groupnorm8_param_grad_f32(const void* dy, const void* x, const void* y, const float* mean_rstd, float* dgamma, float* dbeta, int N,
                                 int HW, int C, int relu, float scale, int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HALLUCIDET_HIP_H */
