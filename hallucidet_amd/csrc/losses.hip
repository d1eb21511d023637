// Detector losses as single launches, forward and backward (torchvision RegionProposalNetwork.compute_loss and
// roi_heads.fastrcnn_loss [EXT], reached from src/utils/eval_forward_fasterrcnn.py:93,141 in the reference).
// The per-op torch form is ~25 elementwise / reduction launches forward and ~25 backward over [N*A] / [R] tensors.
// Per element the arithmetic is ATen's: BCE-with-logits = (1-t)*x + m + log(exp(-m) + exp(-x-m)), m = max(-x, 0);
// smooth-L1(beta) = 0.5*d*d/beta below beta, |d| - 0.5*beta above; cross entropy through a max-shifted log-softmax.
// Sums are two-stage (fixed block partials, then one block in index order): deterministic.
#include "hd_common.h"

namespace {

constexpr int LB = 256;       // threads per block
constexpr int NPART = 256;    // partial-sum blocks

__device__ __forceinline__ float smooth_l1(float d, float beta) {
  const float z = fabsf(d);
  return z < beta ? 0.5f * z * z / beta : z - 0.5f * beta;
}
__device__ __forceinline__ float smooth_l1_grad(float d, float beta) { return d < -beta ? -1.f : (d > beta ? 1.f : d / beta); }

__device__ __forceinline__ void block_sum2(float& a, float& b, float* sm) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    a += __shfl_xor(a, d);
    b += __shfl_xor(b, d);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sm[w * 2] = a;
    sm[w * 2 + 1] = b;
  }
  __syncthreads();
  a = sm[0] + sm[2] + sm[4] + sm[6];
  b = sm[1] + sm[3] + sm[5] + sm[7];
}

__global__ __launch_bounds__(LB) void rpn_loss_fwd_kernel(const float* __restrict__ obj, const float* __restrict__ deltas,
                                                          const float* __restrict__ labels, const float* __restrict__ reg_t,
                                                          const uint8_t* __restrict__ pos, const uint8_t* __restrict__ samp, int64_t T,
                                                          float beta, float* __restrict__ part) {
  __shared__ float sm[8];
  float so = 0.f, sb = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * LB + threadIdx.x; i < T; i += (int64_t)gridDim.x * LB) {
    if (samp[i]) {
      const float x = obj[i], t = labels[i];
      const float m = fmaxf(-x, 0.f);
      so += (1.f - t) * x + m + logf(expf(-m) + expf(-x - m));
    }
    if (pos[i]) {
      const float4 d = *reinterpret_cast<const float4*>(deltas + i * 4);
      const float4 r = *reinterpret_cast<const float4*>(reg_t + i * 4);
      sb += ((smooth_l1(d.x - r.x, beta) + smooth_l1(d.y - r.y, beta)) + smooth_l1(d.z - r.z, beta)) + smooth_l1(d.w - r.w, beta);
    }
  }
  block_sum2(so, sb, sm);
  if (threadIdx.x == 0) {
    part[blockIdx.x * 2] = so;
    part[blockIdx.x * 2 + 1] = sb;
  }
}

// out[j] = sum_b part[b][j] / max(denominator, 1);  denominator = *denom_dev (int64) when given, else denom_host
__global__ __launch_bounds__(LB) void loss_finish_kernel(const float* __restrict__ part, int nb, const int64_t* __restrict__ denom_dev,
                                                         float denom_host, float* __restrict__ out) {
  __shared__ float sm[8];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < nb; i += LB) {
    a += part[i * 2];
    b += part[i * 2 + 1];
  }
  block_sum2(a, b, sm);
  if (threadIdx.x == 0) {
    float dn = denom_dev ? (float)*denom_dev : denom_host;
    dn = dn < 1.f ? 1.f : dn;
    out[0] = a / dn;
    out[1] = b / dn;
  }
}

__global__ __launch_bounds__(LB) void rpn_loss_bwd_kernel(const float* __restrict__ obj, const float* __restrict__ deltas,
                                                          const float* __restrict__ labels, const float* __restrict__ reg_t,
                                                          const uint8_t* __restrict__ pos, const uint8_t* __restrict__ samp, int64_t T,
                                                          float beta, const float* __restrict__ g_obj, const float* __restrict__ g_box,
                                                          const int64_t* __restrict__ denom_dev, float denom_host, float* __restrict__ d_obj,
                                                          float* __restrict__ d_deltas) {
  float dn = denom_dev ? (float)*denom_dev : denom_host;
  dn = dn < 1.f ? 1.f : dn;
  const float go = (g_obj ? *g_obj : 0.f) / dn, gb = (g_box ? *g_box : 0.f) / dn;
  for (int64_t i = (int64_t)blockIdx.x * LB + threadIdx.x; i < T; i += (int64_t)gridDim.x * LB) {
    float v = 0.f;
    if (samp[i]) {
      const float x = obj[i];
      v = (1.f / (1.f + expf(-x)) - labels[i]) * go;
    }
    d_obj[i] = v;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pos[i]) {
      const float4 d = *reinterpret_cast<const float4*>(deltas + i * 4);
      const float4 r = *reinterpret_cast<const float4*>(reg_t + i * 4);
      o.x = smooth_l1_grad(d.x - r.x, beta) * gb;
      o.y = smooth_l1_grad(d.y - r.y, beta) * gb;
      o.z = smooth_l1_grad(d.z - r.z, beta) * gb;
      o.w = smooth_l1_grad(d.w - r.w, beta) * gb;
    }
    *reinterpret_cast<float4*>(d_deltas + i * 4) = o;
  }
}

// Fast R-CNN head: logits [R][K], box_regression [R][K*4], labels [R] i64, reg_t [R][4]
__global__ __launch_bounds__(LB) void frcnn_loss_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ breg,
                                                            const int64_t* __restrict__ labels, const float* __restrict__ reg_t, int R, int K,
                                                            float beta, float* __restrict__ part) {
  __shared__ float sm[8];
  float sc = 0.f, sb = 0.f;
  for (int r = blockIdx.x * LB + threadIdx.x; r < R; r += gridDim.x * LB) {
    const float* lr = logits + (size_t)r * K;
    float mx = lr[0];
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, lr[k]);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += expf(lr[k] - mx);
    const int lab = (int)labels[r];
    if (lab < 0) continue;                      // padding row of the fixed-size RoI stage (hd_fastrcnn_loss_masked)
    sc += -((lr[lab] - mx) - logf(se));
    if (lab > 0) {
      const float4 d = *reinterpret_cast<const float4*>(breg + ((size_t)r * K + lab) * 4);
      const float4 t = *reinterpret_cast<const float4*>(reg_t + (size_t)r * 4);
      sb += ((smooth_l1(d.x - t.x, beta) + smooth_l1(d.y - t.y, beta)) + smooth_l1(d.z - t.z, beta)) + smooth_l1(d.w - t.w, beta);
    }
  }
  block_sum2(sc, sb, sm);
  if (threadIdx.x == 0) {
    part[blockIdx.x * 2] = sc;
    part[blockIdx.x * 2 + 1] = sb;
  }
}

__global__ __launch_bounds__(LB) void frcnn_loss_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ breg,
                                                            const int64_t* __restrict__ labels, const float* __restrict__ reg_t, int R, int K,
                                                            float beta, const float* __restrict__ g_cls, const float* __restrict__ g_box,
                                                            const int64_t* __restrict__ n_dev, float* __restrict__ d_logits,
                                                            float* __restrict__ d_breg) {
  float dn = n_dev ? (float)*n_dev : (float)R;
  dn = dn < 1.f ? 1.f : dn;
  const float gc = (g_cls ? *g_cls : 0.f) / dn, gb = (g_box ? *g_box : 0.f) / dn;
  for (int r = blockIdx.x * LB + threadIdx.x; r < R; r += gridDim.x * LB) {
    const float* lr = logits + (size_t)r * K;
    float mx = lr[0];
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, lr[k]);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += expf(lr[k] - mx);
    const int lab = (int)labels[r];
    for (int k = 0; k < K; ++k) {
      const float p = expf(lr[k] - mx) / se;
      d_logits[(size_t)r * K + k] = lab < 0 ? 0.f : (p - (k == lab ? 1.f : 0.f)) * gc;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k == lab && lab > 0) {
        const float4 d = *reinterpret_cast<const float4*>(breg + ((size_t)r * K + k) * 4);
        const float4 t = *reinterpret_cast<const float4*>(reg_t + (size_t)r * 4);
        o.x = smooth_l1_grad(d.x - t.x, beta) * gb;
        o.y = smooth_l1_grad(d.y - t.y, beta) * gb;
        o.z = smooth_l1_grad(d.z - t.z, beta) * gb;
        o.w = smooth_l1_grad(d.w - t.w, beta) * gb;
      }
      *reinterpret_cast<float4*>(d_breg + ((size_t)r * K + k) * 4) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// RetinaNet losses (reference src/utils/eval_forward_retinanet.py:22-50 focal, :53-80 box loss, :181-244 the two heads) for B images
// that share one anchor set, ONE forward launch (+ finish) and ONE backward launch.  Per anchor a and image b:
//   matched[b][a] >= 0 : foreground, target class = glab[b][matched], box target = BoxCoder.encode(gt[b][matched], anchor)
//   matched == -1      : background (all class targets 0)          matched == -2 : between thresholds, not counted
//   focal(x, t) = alpha_t * bce(x, t) * (1 - p_t)^gamma            (ATen's BCE-with-logits form, p = sigmoid(x))
//   losses: sum over the image / max(1, #foreground), then the mean over images.
// Sums are fixed-order block partials per image: deterministic.
struct FocalT {
  float ce, p;
};
__device__ __forceinline__ FocalT focal_terms(float x) {
  FocalT r;
  r.p = 1.f / (1.f + expf(-x));
  const float m = fmaxf(-x, 0.f);
  r.ce = m + logf(expf(-m) + expf(-x - m));     // bce(x, t) = (1 - t) * x + this
  return r;
}
__device__ __forceinline__ float focal_value(float x, bool t, float alpha, float gamma) {
  const FocalT f = focal_terms(x);
  const float ce = t ? f.ce : x + f.ce;
  const float pt = t ? f.p : 1.f - f.p;
  const float w = gamma == 2.f ? (1.f - pt) * (1.f - pt) : powf(1.f - pt, gamma);
  float l = ce * w;
  if (alpha >= 0.f) l *= t ? alpha : 1.f - alpha;
  return l;
}
__device__ __forceinline__ float focal_grad(float x, bool t, float alpha, float gamma) {
  const FocalT f = focal_terms(x);
  const float p = f.p, q = 1.f - f.p;
  float g;
  if (t) {
    const float w = gamma == 2.f ? q * q : powf(q, gamma);
    g = w * (-gamma * p * f.ce - q);
  } else {
    const float w = gamma == 2.f ? p * p : powf(p, gamma);
    g = w * (p + gamma * q * (x + f.ce));
  }
  if (alpha >= 0.f) g *= t ? alpha : 1.f - alpha;
  return g;
}
__device__ __forceinline__ float4 encode_box(float4 g, float4 a, float wx, float wy, float ww, float wh) {
  const float ew = a.z - a.x, eh = a.w - a.y, ecx = a.x + 0.5f * ew, ecy = a.y + 0.5f * eh;
  const float gw = g.z - g.x, gh = g.w - g.y, gcx = g.x + 0.5f * gw, gcy = g.y + 0.5f * gh;
  return make_float4(wx * (gcx - ecx) / ew, wy * (gcy - ecy) / eh, ww * logf(gw / ew), wh * logf(gh / eh));
}

constexpr int RB = 32;     // blocks per image

// part [B][RB][3] = (focal sum, smooth-L1 sum, #foreground) of the block's anchor range
__global__ __launch_bounds__(LB) void retina_loss_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ breg,
                                                             const int64_t* __restrict__ matched, const float* __restrict__ gt,
                                                             const int64_t* __restrict__ glab, const float* __restrict__ anchors, int A, int K, int G,
                                                             float alpha, float gamma, float beta, float wx, float wy, float ww, float wh,
                                                             float* __restrict__ part) {
  __shared__ float sm[8];
  const int b = blockIdx.y;
  float sc = 0.f, sb = 0.f, nf = 0.f;
  for (int a = blockIdx.x * LB + threadIdx.x; a < A; a += RB * LB) {
    const int64_t m = matched[(size_t)b * A + a];
    if (m == -2) continue;
    const bool fg = m >= 0;
    const int lab = fg ? (int)glab[(size_t)b * G + m] : -1;
    const float* lr = logits + ((size_t)b * A + a) * K;
    for (int k = 0; k < K; ++k) sc += focal_value(lr[k], k == lab, alpha, gamma);
    if (fg) {
      nf += 1.f;
      const float4 t = encode_box(*reinterpret_cast<const float4*>(gt + ((size_t)b * G + m) * 4), *reinterpret_cast<const float4*>(anchors + (size_t)a * 4),
                                  wx, wy, ww, wh);
      const float4 d = *reinterpret_cast<const float4*>(breg + ((size_t)b * A + a) * 4);
      sb += ((smooth_l1(d.x - t.x, beta) + smooth_l1(d.y - t.y, beta)) + smooth_l1(d.z - t.z, beta)) + smooth_l1(d.w - t.w, beta);
    }
  }
  block_sum2(sc, sb, sm);
  __syncthreads();
  float dummy = 0.f;
  block_sum2(nf, dummy, sm);
  if (threadIdx.x == 0) {
    float* o = part + ((size_t)b * RB + blockIdx.x) * 3;
    o[0] = sc;
    o[1] = sb;
    o[2] = nf;
  }
}

// out[0] = mean_b focal_b / max(1, nfg_b), out[1] = mean_b l1_b / max(1, nfg_b); nfg[b] saved for the backward pass
__global__ void retina_loss_finish_kernel(const float* __restrict__ part, int B, float* __restrict__ nfg, float* __restrict__ out) {
  if (threadIdx.x != 0) return;
  float c = 0.f, r = 0.f;
  for (int b = 0; b < B; ++b) {
    float sc = 0.f, sb = 0.f, nf = 0.f;
    for (int i = 0; i < RB; ++i) {
      const float* o = part + ((size_t)b * RB + i) * 3;
      sc += o[0];
      sb += o[1];
      nf += o[2];
    }
    const float dn = nf < 1.f ? 1.f : nf;
    nfg[b] = dn;
    c += sc / dn;
    r += sb / dn;
  }
  out[0] = c / (float)B;
  out[1] = r / (float)(B < 1 ? 1 : B);
}

__global__ __launch_bounds__(LB) void retina_loss_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ breg,
                                                             const int64_t* __restrict__ matched, const float* __restrict__ gt,
                                                             const int64_t* __restrict__ glab, const float* __restrict__ anchors, int A, int K, int G,
                                                             float alpha, float gamma, float beta, float wx, float wy, float ww, float wh,
                                                             const float* __restrict__ nfg, int B, const float* __restrict__ g_cls,
                                                             const float* __restrict__ g_reg, float* __restrict__ d_logits, float* __restrict__ d_breg) {
  const int b = blockIdx.y;
  const float dn = nfg[b] * (float)B;
  const float gc = (g_cls ? *g_cls : 0.f) / dn, gr = (g_reg ? *g_reg : 0.f) / dn;
  for (int a = blockIdx.x * LB + threadIdx.x; a < A; a += gridDim.x * LB) {
    const int64_t m = matched[(size_t)b * A + a];
    const bool fg = m >= 0;
    const int lab = fg ? (int)glab[(size_t)b * G + m] : -1;
    const float* lr = logits + ((size_t)b * A + a) * K;
    float* dl = d_logits + ((size_t)b * A + a) * K;
    for (int k = 0; k < K; ++k) dl[k] = m == -2 ? 0.f : focal_grad(lr[k], k == lab, alpha, gamma) * gc;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (fg) {
      const float4 t = encode_box(*reinterpret_cast<const float4*>(gt + ((size_t)b * G + m) * 4), *reinterpret_cast<const float4*>(anchors + (size_t)a * 4),
                                  wx, wy, ww, wh);
      const float4 d = *reinterpret_cast<const float4*>(breg + ((size_t)b * A + a) * 4);
      o.x = smooth_l1_grad(d.x - t.x, beta) * gr;
      o.y = smooth_l1_grad(d.y - t.y, beta) * gr;
      o.z = smooth_l1_grad(d.z - t.z, beta) * gr;
      o.w = smooth_l1_grad(d.w - t.w, beta) * gr;
    }
    *reinterpret_cast<float4*>(d_breg + ((size_t)b * A + a) * 4) = o;
  }
}

// element-wise focal loss (the reference's `sigmoid_focal_loss(..., reduction="none")`, eval_forward_retinanet.py:22-50) + gradient
__global__ __launch_bounds__(LB) void focal_elem_kernel(const float* __restrict__ x, const float* __restrict__ t, int64_t n, float alpha, float gamma,
                                                        const float* __restrict__ gout, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * LB + threadIdx.x; i < n; i += (int64_t)gridDim.x * LB) {
    const bool on = t[i] > 0.5f;
    out[i] = gout ? focal_grad(x[i], on, alpha, gamma) * gout[i] : focal_value(x[i], on, alpha, gamma);
  }
}

}  // namespace

extern "C" int hd_retinanet_loss(const float* cls_logits, const float* bbox_regression, const int64_t* matched, const float* gt, const int64_t* glab,
                                 const float* anchors, int B, int A, int K, int G, float alpha, float gamma, float beta, const float* coder_weights,
                                 float* part_ws, float* num_fg, float* out2, void* stream) {
  HD_CHECK_ARG(cls_logits && bbox_regression && matched && gt && glab && anchors && coder_weights && part_ws && num_fg && out2 && B > 0 && A > 0 &&
               K > 0 && G > 0 && beta > 0.f, "hd_retinanet_loss: bad args");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(retina_loss_fwd_kernel, dim3(RB, B), dim3(LB), 0, s, cls_logits, bbox_regression, matched, gt, glab, anchors, A, K, G, alpha, gamma,
                     beta, coder_weights[0], coder_weights[1], coder_weights[2], coder_weights[3], part_ws);
  hipLaunchKernelGGL(retina_loss_finish_kernel, dim3(1), dim3(64), 0, s, (const float*)part_ws, B, num_fg, out2);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_retinanet_loss_bwd(const float* cls_logits, const float* bbox_regression, const int64_t* matched, const float* gt, const int64_t* glab,
                                     const float* anchors, int B, int A, int K, int G, float alpha, float gamma, float beta,
                                     const float* coder_weights, const float* num_fg, const float* g_cls, const float* g_reg, float* d_cls_logits,
                                     float* d_bbox_regression, void* stream) {
  HD_CHECK_ARG(cls_logits && bbox_regression && matched && gt && glab && anchors && coder_weights && num_fg && d_cls_logits && d_bbox_regression &&
               B > 0 && A > 0 && K > 0 && G > 0, "hd_retinanet_loss_bwd: bad args");
  int g = (A + LB - 1) / LB;
  if (g > 256) g = 256;
  hipLaunchKernelGGL(retina_loss_bwd_kernel, dim3(g, B), dim3(LB), 0, (hipStream_t)stream, cls_logits, bbox_regression, matched, gt, glab, anchors, A, K, G,
                     alpha, gamma, beta, coder_weights[0], coder_weights[1], coder_weights[2], coder_weights[3], num_fg, B, g_cls, g_reg, d_cls_logits,
                     d_bbox_regression);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_sigmoid_focal_loss(const float* inputs, const float* targets, int64_t n, float alpha, float gamma, const float* grad_out, float* out,
                                     void* stream) {
  HD_CHECK_ARG(inputs && targets && out && n >= 0, "hd_sigmoid_focal_loss: bad args");
  if (n == 0) return HD_OK;
  int g = (int)((n + LB - 1) / LB);
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(focal_elem_kernel, dim3(g), dim3(LB), 0, (hipStream_t)stream, inputs, targets, n, alpha, gamma, grad_out, out);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_rpn_loss(const float* objectness, const float* deltas, const float* labels, const float* reg_t, const uint8_t* pos,
                           const uint8_t* samp, int64_t T, float beta, const int64_t* n_sampled_dev, float n_sampled_host, float* part_ws,
                           float* out2, void* stream) {
  HD_CHECK_ARG(objectness && deltas && labels && reg_t && pos && samp && part_ws && out2 && T >= 0 && beta > 0.f, "hd_rpn_loss: bad args");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(rpn_loss_fwd_kernel, dim3(NPART), dim3(LB), 0, s, objectness, deltas, labels, reg_t, pos, samp, T, beta, part_ws);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(LB), 0, s, (const float*)part_ws, NPART, n_sampled_dev, n_sampled_host, out2);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_rpn_loss_bwd(const float* objectness, const float* deltas, const float* labels, const float* reg_t, const uint8_t* pos,
                               const uint8_t* samp, int64_t T, float beta, const float* g_obj, const float* g_box,
                               const int64_t* n_sampled_dev, float n_sampled_host, float* d_objectness, float* d_deltas, void* stream) {
  HD_CHECK_ARG(objectness && deltas && labels && reg_t && pos && samp && d_objectness && d_deltas && T >= 0, "hd_rpn_loss_bwd: bad args");
  if (T == 0) return HD_OK;
  int g = (int)((T + LB - 1) / LB);
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(rpn_loss_bwd_kernel, dim3(g), dim3(LB), 0, (hipStream_t)stream, objectness, deltas, labels, reg_t, pos, samp, T, beta, g_obj,
                     g_box, n_sampled_dev, n_sampled_host, d_objectness, d_deltas);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_fastrcnn_loss(const float* logits, const float* box_regression, const int64_t* labels, const float* reg_t, int R, int K,
                                float beta, float* part_ws, float* out2, void* stream) {
  HD_CHECK_ARG(logits && box_regression && labels && reg_t && part_ws && out2 && R >= 0 && K >= 1 && beta > 0.f, "hd_fastrcnn_loss: bad args");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(frcnn_loss_fwd_kernel, dim3(NPART), dim3(LB), 0, s, logits, box_regression, labels, reg_t, R, K, beta, part_ws);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(LB), 0, s, (const float*)part_ws, NPART, (const int64_t*)nullptr, (float)R, out2);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_fastrcnn_loss_bwd(const float* logits, const float* box_regression, const int64_t* labels, const float* reg_t, int R, int K,
                                    float beta, const float* g_cls, const float* g_box, float* d_logits, float* d_box_regression,
                                    void* stream) {
  HD_CHECK_ARG(logits && box_regression && labels && reg_t && d_logits && d_box_regression && R >= 0 && K >= 1, "hd_fastrcnn_loss_bwd: bad args");
  if (R == 0) return HD_OK;
  int g = (R + LB - 1) / LB;
  hipLaunchKernelGGL(frcnn_loss_bwd_kernel, dim3(g), dim3(LB), 0, (hipStream_t)stream, logits, box_regression, labels, reg_t, R, K, beta, g_cls,
                     g_box, (const int64_t*)nullptr, d_logits, d_box_regression);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

// Fixed-size RoI stage: R = images x batch_size_per_image rows, rows with label < 0 are padding (no loss, no gradient); both losses
// divide by the number of REAL rows, read from the device (n_valid_dev) -- no host synchronisation sizes anything.
extern "C" int hd_fastrcnn_loss_masked(const float* logits, const float* box_regression, const int64_t* labels, const float* reg_t, int R, int K,
                                       float beta, const int64_t* n_valid_dev, float* part_ws, float* out2, void* stream) {
  HD_CHECK_ARG(logits && box_regression && labels && reg_t && n_valid_dev && part_ws && out2 && R >= 0 && K >= 1 && beta > 0.f,
               "hd_fastrcnn_loss_masked: bad args");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(frcnn_loss_fwd_kernel, dim3(NPART), dim3(LB), 0, s, logits, box_regression, labels, reg_t, R, K, beta, part_ws);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(LB), 0, s, (const float*)part_ws, NPART, n_valid_dev, 0.f, out2);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_fastrcnn_loss_masked_bwd(const float* logits, const float* box_regression, const int64_t* labels, const float* reg_t, int R, int K,
                                           float beta, const int64_t* n_valid_dev, const float* g_cls, const float* g_box, float* d_logits,
                                           float* d_box_regression, void* stream) {
  HD_CHECK_ARG(logits && box_regression && labels && reg_t && n_valid_dev && d_logits && d_box_regression && R >= 0 && K >= 1,
               "hd_fastrcnn_loss_masked_bwd: bad args");
  if (R == 0) return HD_OK;
  int g = (R + LB - 1) / LB;
  hipLaunchKernelGGL(frcnn_loss_bwd_kernel, dim3(g), dim3(LB), 0, (hipStream_t)stream, logits, box_regression, labels, reg_t, R, K, beta, g_cls,
                     g_box, n_valid_dev, d_logits, d_box_regression);
  HD_CHECK_LAUNCH();
  return HD_OK;
}
