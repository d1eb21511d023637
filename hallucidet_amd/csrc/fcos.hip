// FCOS pieces that the other two detectors do not have (torchvision.models.detection.fcos [EXT], driven by the reference's
// src/utils/eval_forward_fcos.py:54-83): GroupNorm(32) + ReLU of the head towers (forward and data gradient), the
// center-sampling target assignment of FCOS.compute_loss, and the three losses of FCOSHead.compute_loss (sigmoid focal,
// generalized IoU on decoded boxes, centre-ness BCE) as one forward and one backward launch.
// All of it is HBM / latency bound work on small tensors (5 pyramid levels of a 300 x 300 image: 1 939 locations).
#include "hd_common.h"

namespace {

constexpr int GB = 1024;   // GroupNorm block: (C/8) channel vectors x pixel lanes

// ---------------------------------------------------------------------------------------------------------------------
// GroupNorm over NHWC f16 with EIGHT channels per group (GroupNorm(32, 256): a group is one 16-byte channel vector of a pixel).
// One block per image: thread = (pixel lane, channel vector); consecutive threads read consecutive 16 B (a pixel's 512 B row).
// Statistics in fp32 (biased variance, as torch.nn.functional.group_norm), saved as stat[n][g] = (mean, rstd).
// The second pass re-reads x from L2 (a level of one image is at most 740 KB).
__global__ __launch_bounds__(GB) void groupnorm8_fwd_kernel(const f16* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, f16* __restrict__ y, float* __restrict__ stat,
                                                            int HW, int C, float eps, int relu) {
  __shared__ float red[GB * 2];
  __shared__ float st[256];
  const int vecs = C >> 3, PL = GB / vecs;
  const int v = threadIdx.x % vecs, pl = threadIdx.x / vecs;
  const int n = blockIdx.x;
  const f16* xn = x + (size_t)n * HW * C;
  float s = 0.f, q = 0.f;
  for (int p = pl; p < HW; p += PL) {
    const f16x8 a = *reinterpret_cast<const f16x8*>(xn + (size_t)p * C + v * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float f = (float)a[k];
      s += f;
      q += f * f;
    }
  }
  red[threadIdx.x * 2] = s;
  red[threadIdx.x * 2 + 1] = q;
  __syncthreads();
  if (pl == 0) {
    float ts = 0.f, tq = 0.f;
    for (int i = 0; i < PL; ++i) {               // fixed order: deterministic
      ts += red[(i * vecs + v) * 2];
      tq += red[(i * vecs + v) * 2 + 1];
    }
    const float m = ts / (float)(HW * 8);
    float var = tq / (float)(HW * 8) - m * m;
    if (var < 0.f) var = 0.f;
    const float r = rsqrtf(var + eps);
    st[v * 2] = m;
    st[v * 2 + 1] = r;
    stat[((size_t)n * vecs + v) * 2] = m;
    stat[((size_t)n * vecs + v) * 2 + 1] = r;
  }
  __syncthreads();
  const float m = st[v * 2], r = st[v * 2 + 1];
  float ga[8], be[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    ga[k] = gamma[v * 8 + k] * r;
    be[k] = beta[v * 8 + k] - m * ga[k];
  }
  f16* yn = y + (size_t)n * HW * C;
  for (int p = pl; p < HW; p += PL) {
    const f16x8 a = *reinterpret_cast<const f16x8*>(xn + (size_t)p * C + v * 8);
    f16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float f = (float)a[k] * ga[k] + be[k];
      if (relu) f = fmaxf(f, 0.f);
      o[k] = (f16)f;
    }
    *reinterpret_cast<f16x8*>(yn + (size_t)p * C + v * 8) = o;
  }
}

// dx of y = relu(gn(x)):  g = dy * [y > 0];  xh = (x - mean) * rstd;  dx = rstd * (g*gamma - mean_grp(g*gamma) - xh * mean_grp(g*gamma*xh))
__global__ __launch_bounds__(GB) void groupnorm8_bwd_kernel(const f16* __restrict__ dy, const f16* __restrict__ x, const f16* __restrict__ y,
                                                            const float* __restrict__ gamma, const float* __restrict__ stat,
                                                            f16* __restrict__ dx, int HW, int C, int relu) {
  __shared__ float red[GB * 2];
  __shared__ float st[256];
  const int vecs = C >> 3, PL = GB / vecs;
  const int v = threadIdx.x % vecs, pl = threadIdx.x / vecs;
  const int n = blockIdx.x;
  const size_t base = (size_t)n * HW * C;
  const float m = stat[((size_t)n * vecs + v) * 2], r = stat[((size_t)n * vecs + v) * 2 + 1];
  float ga[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) ga[k] = gamma[v * 8 + k];
  float s1 = 0.f, s2 = 0.f;
  for (int p = pl; p < HW; p += PL) {
    const size_t off = base + (size_t)p * C + v * 8;
    const f16x8 g = *reinterpret_cast<const f16x8*>(dy + off), a = *reinterpret_cast<const f16x8*>(x + off);
    f16x8 o;
    if (relu) o = *reinterpret_cast<const f16x8*>(y + off);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float gk = (float)g[k];
      if (relu && !((float)o[k] > 0.f)) gk = 0.f;
      gk *= ga[k];
      s1 += gk;
      s2 += gk * (((float)a[k] - m) * r);
    }
  }
  red[threadIdx.x * 2] = s1;
  red[threadIdx.x * 2 + 1] = s2;
  __syncthreads();
  if (pl == 0) {
    float t1 = 0.f, t2 = 0.f;
    for (int i = 0; i < PL; ++i) {
      t1 += red[(i * vecs + v) * 2];
      t2 += red[(i * vecs + v) * 2 + 1];
    }
    st[v * 2] = t1 / (float)(HW * 8);
    st[v * 2 + 1] = t2 / (float)(HW * 8);
  }
  __syncthreads();
  const float m1 = st[v * 2], m2 = st[v * 2 + 1];
  for (int p = pl; p < HW; p += PL) {
    const size_t off = base + (size_t)p * C + v * 8;
    const f16x8 g = *reinterpret_cast<const f16x8*>(dy + off), a = *reinterpret_cast<const f16x8*>(x + off);
    f16x8 o, d;
    if (relu) o = *reinterpret_cast<const f16x8*>(y + off);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float gk = (float)g[k];
      if (relu && !((float)o[k] > 0.f)) gk = 0.f;
      gk *= ga[k];
      const float xh = ((float)a[k] - m) * r;
      d[k] = (f16)(r * (gk - m1 - xh * m2));
    }
    *reinterpret_cast<f16x8*>(dx + off) = d;
  }
}

// dgamma[c] (+)= scale * sum_{n,p} g * xh,  dbeta[c] (+)= scale * sum g   (g = dy * [y > 0]); one block per 8-channel group, all
// images and pixels walked by 1 024 threads, fixed-order LDS reduction (deterministic).  Detector fine-tuning only.
__global__ __launch_bounds__(GB) void groupnorm8_param_grad_kernel(const f16* __restrict__ dy, const f16* __restrict__ x, const f16* __restrict__ y,
                                                                   const float* __restrict__ stat, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, int N, int HW, int C, int relu, float scale,
                                                                   int accumulate) {
  __shared__ float red[GB][17];
  const int v = blockIdx.x, vecs = C >> 3;
  float sg[8], sb[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) sg[k] = sb[k] = 0.f;
  const int64_t T = (int64_t)N * HW;
  for (int64_t i = threadIdx.x; i < T; i += GB) {
    const int n = (int)(i / HW);
    const float m = stat[((size_t)n * vecs + v) * 2], r = stat[((size_t)n * vecs + v) * 2 + 1];
    const size_t off = (size_t)i * C + v * 8;
    const f16x8 g = *reinterpret_cast<const f16x8*>(dy + off), a = *reinterpret_cast<const f16x8*>(x + off);
    f16x8 o;
    if (relu) o = *reinterpret_cast<const f16x8*>(y + off);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float gk = (float)g[k];
      if (relu && !((float)o[k] > 0.f)) gk = 0.f;
      sb[k] += gk;
      sg[k] += gk * (((float)a[k] - m) * r);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    red[threadIdx.x][k] = sg[k];
    red[threadIdx.x][8 + k] = sb[k];
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    float t = 0.f;
    for (int i = 0; i < GB; ++i) t += red[i][threadIdx.x];
    const int c = v * 8 + (threadIdx.x & 7);
    float* dst = threadIdx.x < 8 ? dgamma : dbeta;
    dst[c] = accumulate ? dst[c] + t * scale : t * scale;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// FCOS.compute_loss target assignment [EXT]: location a (anchor box = stride-sized square around the location) is matched to
// ground-truth box j iff  max(|cx_a - cx_j|, |cy_a - cy_j|) < radius * size_a  (centre sampling),  the location lies strictly
// inside the box,  and the largest of its four distances to the box sides lies in (4 size_a, 8 size_a)  (the first level's
// lower bound is 0, the last level's upper bound is inf);  among several matches the box of SMALLEST area wins
// (arg max of match * (1e8 - area), first index on ties), none -> -1.
__global__ void fcos_match_kernel(const float* __restrict__ anchors, const float* __restrict__ gt, const uint8_t* __restrict__ gvalid, int A, int G,
                                  int first_n, int last_start, float radius, int64_t* __restrict__ matched) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (a >= A) return;
  const float4 an = *reinterpret_cast<const float4*>(anchors + (size_t)a * 4);
  const float cx = (an.x + an.z) / 2.f, cy = (an.y + an.w) / 2.f, size = an.z - an.x;
  const float lower = a < first_n ? 0.f : size * 4.f;
  const float upper = a >= last_start ? __builtin_huge_valf() : size * 8.f;
  float best = 0.f;
  int bi = 0;
  for (int j = 0; j < G; ++j) {
    if (!gvalid[(size_t)b * G + j]) continue;
    const float4 g = *reinterpret_cast<const float4*>(gt + ((size_t)b * G + j) * 4);
    const float gcx = (g.x + g.z) / 2.f, gcy = (g.y + g.w) / 2.f;
    bool ok = fmaxf(fabsf(cx - gcx), fabsf(cy - gcy)) < radius * size;
    const float l = cx - g.x, t = cy - g.y, r = g.z - cx, bt = g.w - cy;
    ok = ok && fminf(fminf(l, t), fminf(r, bt)) > 0.f;
    const float dmax = fmaxf(fmaxf(l, t), fmaxf(r, bt));
    ok = ok && dmax > lower && dmax < upper;
    const float area = (g.z - g.x) * (g.w - g.y);
    const float val = ok ? 1e8f - area : 0.f;
    if (val > best) {
      best = val;
      bi = j;
    }
  }
  matched[(size_t)b * A + a] = best < 1e-5f ? -1 : bi;
}

// ---------------------------------------------------------------------------------------------------------------------
// FCOSHead.compute_loss [EXT] over B images sharing one location set:
//   classification: sigmoid focal (alpha 0.25, gamma 2) of [B, A, K] logits against the one-hot of the matched box's label, summed
//   box:            generalized-IoU loss (eps 1e-7) between BoxLinearCoder.decode(relu'd ltrb, location) and the matched box, foreground
//   centre-ness:    BCE-with-logits against sqrt(min(l,r)/max(l,r) * min(t,b)/max(t,b)) of the matched box's ltrb targets, foreground
//   each divided by max(1, #foreground of the WHOLE batch).
constexpr int LB = 256;
constexpr int FB = 64;      // partial-sum blocks

struct FocalT {
  float ce, p;
};
__device__ __forceinline__ FocalT focal_terms(float x) {
  FocalT r;
  r.p = 1.f / (1.f + expf(-x));
  const float m = fmaxf(-x, 0.f);
  r.ce = m + logf(expf(-m) + expf(-x - m));
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

struct Giou {
  float loss;
  float d[4];      // d loss / d (x1, y1, x2, y2) of the predicted box
};
// torchvision.ops.generalized_box_iou_loss [EXT]; ties of the max / min pairs split the gradient evenly (ATen's maximum / minimum)
__device__ __forceinline__ Giou giou_terms(float x1, float y1, float x2, float y2, float4 g, bool want_grad) {
  constexpr float eps = 1e-7f;
  const float ix1 = fmaxf(x1, g.x), iy1 = fmaxf(y1, g.y), ix2 = fminf(x2, g.z), iy2 = fminf(y2, g.w);
  const bool has = (iy2 > iy1) && (ix2 > ix1);
  const float iw = ix2 - ix1, ih = iy2 - iy1;
  const float I = has ? iw * ih : 0.f;
  const float pw = x2 - x1, ph = y2 - y1;
  const float U = pw * ph + (g.z - g.x) * (g.w - g.y) - I;
  const float iou = I / (U + eps);
  const float cx1 = fminf(x1, g.x), cy1 = fminf(y1, g.y), cx2 = fmaxf(x2, g.z), cy2 = fmaxf(y2, g.w);
  const float cw = cx2 - cx1, ch = cy2 - cy1;
  const float Cc = cw * ch;
  Giou r;
  r.loss = 1.f - (iou - (Cc - U) / (Cc + eps));
  if (!want_grad) return r;
  auto share = [](float a, float b, bool a_wins_if_greater) -> float {      // gradient share of a in max(a, b) / min(a, b)
    if (a == b) return 0.5f;
    return ((a > b) == a_wins_if_greater) ? 1.f : 0.f;
  };
  // per coordinate: dAp, dI, dC
  const float dAp[4] = {-ph, -pw, ph, pw};
  const float dI[4] = {has ? -ih * share(x1, g.x, true) : 0.f, has ? -iw * share(y1, g.y, true) : 0.f,
                       has ? ih * share(x2, g.z, false) : 0.f, has ? iw * share(y2, g.w, false) : 0.f};
  const float dC[4] = {-ch * share(x1, g.x, false), -cw * share(y1, g.y, false), ch * share(x2, g.z, true), cw * share(y2, g.w, true)};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float dU = dAp[k] - dI[k];
    const float diou = (dI[k] * (U + eps) - I * dU) / ((U + eps) * (U + eps));
    const float dpen = ((dC[k] - dU) * (Cc + eps) - (Cc - U) * dC[k]) / ((Cc + eps) * (Cc + eps));
    r.d[k] = -diou + dpen;
  }
  return r;
}

__device__ __forceinline__ float ctr_target(float cx, float cy, float4 g) {
  // the size normalisation of BoxLinearCoder.encode cancels in both ratios
  const float l = cx - g.x, t = cy - g.y, r = g.z - cx, b = g.w - cy;
  return sqrtf((fminf(l, r) / fmaxf(l, r)) * (fminf(t, b) / fmaxf(t, b)));
}

__device__ __forceinline__ void block_sum4(float (&v)[4], float* sm) {
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v[k] += __shfl_xor(v[k], d);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) sm[w * 4 + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = (sm[k] + sm[4 + k]) + (sm[8 + k] + sm[12 + k]);
}

// part [FB][4] = (focal sum, giou sum, centre-ness sum, #foreground) over the block's (image, location) range
__global__ __launch_bounds__(LB) void fcos_loss_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ breg, const float* __restrict__ ctr,
                                                           const int64_t* __restrict__ matched, const float* __restrict__ gt,
                                                           const int64_t* __restrict__ glab, const float* __restrict__ anchors, int B, int A, int K,
                                                           int G, float alpha, float gamma, float* __restrict__ part) {
  __shared__ float sm[16];
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t T = (int64_t)B * A;
  for (int64_t i = (int64_t)blockIdx.x * LB + threadIdx.x; i < T; i += (int64_t)FB * LB) {
    const int b = (int)(i / A), a = (int)(i - (int64_t)b * A);
    const int64_t m = matched[i];
    const bool fg = m >= 0;
    const int lab = fg ? (int)glab[(size_t)b * G + m] : -1;
    const float* lr = logits + i * K;
    for (int k = 0; k < K; ++k) acc[0] += focal_value(lr[k], k == lab, alpha, gamma);
    if (fg) {
      const float4 an = *reinterpret_cast<const float4*>(anchors + (size_t)a * 4);
      const float4 g = *reinterpret_cast<const float4*>(gt + ((size_t)b * G + m) * 4);
      const float4 d = *reinterpret_cast<const float4*>(breg + i * 4);
      const float cx = 0.5f * (an.x + an.z), cy = 0.5f * (an.y + an.w), w = an.z - an.x, h = an.w - an.y;
      acc[1] += giou_terms(cx - d.x * w, cy - d.y * h, cx + d.z * w, cy + d.w * h, g, false).loss;
      const float t = ctr_target(cx, cy, g), x = ctr[i];
      const float mm = fmaxf(-x, 0.f);
      acc[2] += (1.f - t) * x + mm + logf(expf(-mm) + expf(-x - mm));
      acc[3] += 1.f;
    }
  }
  block_sum4(acc, sm);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) part[blockIdx.x * 4 + k] = acc[k];
  }
}

__global__ void fcos_loss_finish_kernel(const float* __restrict__ part, float* __restrict__ nfg, float* __restrict__ out3) {
  if (threadIdx.x != 0) return;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < FB; ++i)
    for (int k = 0; k < 4; ++k) s[k] += part[i * 4 + k];
  const float dn = s[3] < 1.f ? 1.f : s[3];
  nfg[0] = dn;
  out3[0] = s[0] / dn;
  out3[1] = s[1] / dn;
  out3[2] = s[2] / dn;
}

__global__ __launch_bounds__(LB) void fcos_loss_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ breg, const float* __restrict__ ctr,
                                                           const int64_t* __restrict__ matched, const float* __restrict__ gt,
                                                           const int64_t* __restrict__ glab, const float* __restrict__ anchors, int B, int A, int K,
                                                           int G, float alpha, float gamma, const float* __restrict__ nfg,
                                                           const float* __restrict__ g3, float* __restrict__ d_logits, float* __restrict__ d_breg,
                                                           float* __restrict__ d_ctr) {
  const float dn = nfg[0];
  const float gc = g3[0] / dn, gr = g3[1] / dn, gt_ = g3[2] / dn;
  const int64_t T = (int64_t)B * A;
  for (int64_t i = (int64_t)blockIdx.x * LB + threadIdx.x; i < T; i += (int64_t)gridDim.x * LB) {
    const int b = (int)(i / A), a = (int)(i - (int64_t)b * A);
    const int64_t m = matched[i];
    const bool fg = m >= 0;
    const int lab = fg ? (int)glab[(size_t)b * G + m] : -1;
    const float* lr = logits + i * K;
    for (int k = 0; k < K; ++k) d_logits[i * K + k] = focal_grad(lr[k], k == lab, alpha, gamma) * gc;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    float dc = 0.f;
    if (fg) {
      const float4 an = *reinterpret_cast<const float4*>(anchors + (size_t)a * 4);
      const float4 g = *reinterpret_cast<const float4*>(gt + ((size_t)b * G + m) * 4);
      const float4 d = *reinterpret_cast<const float4*>(breg + i * 4);
      const float cx = 0.5f * (an.x + an.z), cy = 0.5f * (an.y + an.w), w = an.z - an.x, h = an.w - an.y;
      const Giou q = giou_terms(cx - d.x * w, cy - d.y * h, cx + d.z * w, cy + d.w * h, g, true);
      o = make_float4(-w * q.d[0] * gr, -h * q.d[1] * gr, w * q.d[2] * gr, h * q.d[3] * gr);
      const float t = ctr_target(cx, cy, g), x = ctr[i];
      dc = (1.f / (1.f + expf(-x)) - t) * gt_;
    }
    *reinterpret_cast<float4*>(d_breg + i * 4) = o;
    d_ctr[i] = dc;
  }
}

}  // namespace

static bool pow2i(int v) { return v > 0 && (v & (v - 1)) == 0; }

extern "C" int HD_API(hd_groupnorm8_relu)(const void* x, const float* gamma, const float* beta, void* y, float* mean_rstd, int N, int HW, int C,
                                  float eps, int relu, void* stream) {
  HD_CHECK_ARG(x && gamma && beta && y && mean_rstd && N > 0 && HW > 0, "hd_groupnorm8_relu: bad args");
  HD_CHECK_ARG(C % 8 == 0 && pow2i(C / 8) && C / 8 <= 128, "hd_groupnorm8_relu: C/8 (= number of 8-channel groups) must be a power of two <= 128 (C=%d)", C);
  hipLaunchKernelGGL(groupnorm8_fwd_kernel, dim3(N), dim3(GB), 0, (hipStream_t)stream, (const f16*)x, gamma, beta, (f16*)y, mean_rstd, HW, C, eps, relu);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_groupnorm8_relu_bwd)(const void* dy, const void* x, const void* y, const float* gamma, const float* mean_rstd, void* dx, int N,
                                      int HW, int C, int relu, void* stream) {
  HD_CHECK_ARG(dy && x && gamma && mean_rstd && dx && (y || !relu) && N > 0 && HW > 0, "hd_groupnorm8_relu_bwd: bad args");
  HD_CHECK_ARG(C % 8 == 0 && pow2i(C / 8) && C / 8 <= 128, "hd_groupnorm8_relu_bwd: C/8 must be a power of two <= 128 (C=%d)", C);
  hipLaunchKernelGGL(groupnorm8_bwd_kernel, dim3(N), dim3(GB), 0, (hipStream_t)stream, (const f16*)dy, (const f16*)x, (const f16*)y, gamma, mean_rstd,
                     (f16*)dx, HW, C, relu);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_groupnorm8_param_grad)(const void* dy, const void* x, const void* y, const float* mean_rstd, float* dgamma, float* dbeta, int N,
                                        int HW, int C, int relu, float scale, int accumulate, void* stream) {
  HD_CHECK_ARG(dy && x && mean_rstd && dgamma && dbeta && (y || !relu) && N > 0 && HW > 0, "hd_groupnorm8_param_grad: bad args");
  HD_CHECK_ARG(C % 8 == 0 && C / 8 <= 128, "hd_groupnorm8_param_grad: C must be a multiple of 8, at most 1024 (C=%d)", C);
  hipLaunchKernelGGL(groupnorm8_param_grad_kernel, dim3(C / 8), dim3(GB), 0, (hipStream_t)stream, (const f16*)dy, (const f16*)x, (const f16*)y,
                     mean_rstd, dgamma, dbeta, N, HW, C, relu, scale, accumulate);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

#ifndef HD_STORE_F32
extern "C" int hd_fcos_match(const float* anchors, const float* gt, const uint8_t* gvalid, int B, int A, int G, int first_level_count,
                             int last_level_start, float center_sampling_radius, int64_t* matched, void* stream) {
  HD_CHECK_ARG(anchors && gt && gvalid && matched && B > 0 && A > 0 && G > 0, "hd_fcos_match: bad args");
  hipLaunchKernelGGL(fcos_match_kernel, dim3(hd_cdiv(A, 256), B), dim3(256), 0, (hipStream_t)stream, anchors, gt, gvalid, A, G, first_level_count,
                     last_level_start, center_sampling_radius, matched);
  HD_CHECK_LAUNCH();
  return HD_OK;
}
#endif

#ifndef HD_STORE_F32
extern "C" int hd_fcos_loss(const float* cls_logits, const float* bbox_regression, const float* bbox_ctrness, const int64_t* matched, const float* gt,
                            const int64_t* glab, const float* anchors, int B, int A, int K, int G, float alpha, float gamma, float* part_ws,
                            float* num_fg, float* out3, void* stream) {
  HD_CHECK_ARG(cls_logits && bbox_regression && bbox_ctrness && matched && gt && glab && anchors && part_ws && num_fg && out3 && B > 0 && A > 0 &&
               K > 0 && G > 0, "hd_fcos_loss: bad args");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(fcos_loss_fwd_kernel, dim3(FB), dim3(LB), 0, s, cls_logits, bbox_regression, bbox_ctrness, matched, gt, glab, anchors, B, A, K, G,
                     alpha, gamma, part_ws);
  hipLaunchKernelGGL(fcos_loss_finish_kernel, dim3(1), dim3(64), 0, s, (const float*)part_ws, num_fg, out3);
  HD_CHECK_LAUNCH();
  return HD_OK;
}
#endif

#ifndef HD_STORE_F32
extern "C" int hd_fcos_loss_bwd(const float* cls_logits, const float* bbox_regression, const float* bbox_ctrness, const int64_t* matched,
                                const float* gt, const int64_t* glab, const float* anchors, int B, int A, int K, int G, float alpha, float gamma,
                                const float* num_fg, const float* g3, float* d_cls_logits, float* d_bbox_regression, float* d_bbox_ctrness,
                                void* stream) {
  HD_CHECK_ARG(cls_logits && bbox_regression && bbox_ctrness && matched && gt && glab && anchors && num_fg && g3 && d_cls_logits &&
               d_bbox_regression && d_bbox_ctrness && B > 0 && A > 0 && K > 0 && G > 0, "hd_fcos_loss_bwd: bad args");
  int g = hd_cdiv((int64_t)B * A, LB);
  if (g > 256) g = 256;
  hipLaunchKernelGGL(fcos_loss_bwd_kernel, dim3(g), dim3(LB), 0, (hipStream_t)stream, cls_logits, bbox_regression, bbox_ctrness, matched, gt, glab,
                     anchors, B, A, K, G, alpha, gamma, num_fg, g3, d_cls_logits, d_bbox_regression, d_bbox_ctrness);
  HD_CHECK_LAUNCH();
  return HD_OK;
}
#endif
