// RoIAlign (aligned=False) forward / backward, single- and multi-level (torchvision.ops.roi_align / MultiScaleRoIAlign [EXT], reached from
// src/utils/eval_forward_fasterrcnn.py:120-123 `box_roi_pool`).  The only detection kernels that touch ACTIVATIONS: this file is built
// twice (fp16 storage, and -DHD_STORE_F32 for precision=32: entry points with the suffix _f32); box maths in fp32 either way.
#include "hd_common.h"
#include <cstdlib>
#pragma clang fp contract(off)

namespace {

// ------------------------------------------------------------------ RoIAlign
struct Bilin {
  int yl, xl, yh, xh;
  float w1, w2, w3, w4;
  bool valid;
};

__device__ __forceinline__ Bilin bilin_setup(float y, float x, int H, int W) {
  Bilin b;
  b.valid = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int yl = (int)y, xl = (int)x, yh, xh;
  if (yl >= H - 1) {
    yh = yl = H - 1;
    y = (float)yl;
  } else
    yh = yl + 1;
  if (xl >= W - 1) {
    xh = xl = W - 1;
    x = (float)xl;
  } else
    xh = xl + 1;
  float ly = y - (float)yl, lx = x - (float)xl;
  float hy = 1.f - ly, hx = 1.f - lx;
  b.yl = yl; b.xl = xl; b.yh = yh; b.xh = xh;
  b.w1 = hy * hx; b.w2 = hy * lx; b.w3 = ly * hx; b.w4 = ly * lx;
  return b;
}

// acc + w * t[K] as ONE fused multiply-add that reads the fp16 element directly (v_fma_mix_f32: no conversion instruction)
template <int K>
__device__ __forceinline__ float mac_tap(float w, const f16x8& t, float acc) {
#ifdef HD_STORE_F32
  return fmaf(w, t[K], acc);
#else
  typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
  const uint32_t pair = __builtin_bit_cast(u32x4_, t)[K >> 1];
  float d;
  if (K & 1)
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(pair), "v"(w), "v"(acc));
  else
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(pair), "v"(w), "v"(acc));
  return d;
#endif
}

// acc[0..7] += w * t[0..7]; FUSED: one v_fma_mix_f32 per element, otherwise convert, multiply, add (the round-3 arithmetic, kept for A/B)
template <bool FUSED = true>
__device__ __forceinline__ void mac_vec(float w, const f16x8& t, float* acc) {
  if (!FUSED) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += w * (float)t[k];
    return;
  }
  acc[0] = mac_tap<0>(w, t, acc[0]); acc[1] = mac_tap<1>(w, t, acc[1]); acc[2] = mac_tap<2>(w, t, acc[2]); acc[3] = mac_tap<3>(w, t, acc[3]);
  acc[4] = mac_tap<4>(w, t, acc[4]); acc[5] = mac_tap<5>(w, t, acc[5]); acc[6] = mac_tap<6>(w, t, acc[6]); acc[7] = mac_tap<7>(w, t, acc[7]);
}

__global__ void roi_align_kernel(const f16* __restrict__ feat, const float* __restrict__ rois, f16* __restrict__ out, int R,
                                 int H, int W, int C, int PH, int PW, float scale, int sr) {
  const int vecs = C / 8;
  const int64_t total = (int64_t)R * PH * PW * vecs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int v = (int)(i % vecs);
    int64_t q = i / vecs;
    int pw = (int)(q % PW);
    int ph = (int)((q / PW) % PH);
    int r = (int)(q / ((int64_t)PW * PH));
    const float* roi = rois + (size_t)r * 5;
    int n = (int)roi[0];
    float rsw = roi[1] * scale, rsh = roi[2] * scale, rew = roi[3] * scale, reh = roi[4] * scale;
    float rw = fmaxf(rew - rsw, 1.f), rh = fmaxf(reh - rsh, 1.f);
    float bh = rh / (float)PH, bw = rw / (float)PW;
    int gh = sr > 0 ? sr : (int)ceilf(rh / (float)PH);
    int gw = sr > 0 ? sr : (int)ceilf(rw / (float)PW);
    float count = fmaxf((float)(gh * gw), 1.f);
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    const f16* fb = feat + (size_t)n * H * W * C + v * 8;
    for (int iy = 0; iy < gh; ++iy) {
      float y = rsh + (float)ph * bh + ((float)iy + .5f) * bh / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        float x = rsw + (float)pw * bw + ((float)ix + .5f) * bw / (float)gw;
        Bilin b = bilin_setup(y, x, H, W);
        if (!b.valid) continue;
        f16x8 v1 = *reinterpret_cast<const f16x8*>(fb + ((size_t)b.yl * W + b.xl) * C);
        f16x8 v2 = *reinterpret_cast<const f16x8*>(fb + ((size_t)b.yl * W + b.xh) * C);
        f16x8 v3 = *reinterpret_cast<const f16x8*>(fb + ((size_t)b.yh * W + b.xl) * C);
        f16x8 v4 = *reinterpret_cast<const f16x8*>(fb + ((size_t)b.yh * W + b.xh) * C);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += b.w1 * (float)v1[k] + b.w2 * (float)v2[k] + b.w3 * (float)v3[k] + b.w4 * (float)v4[k];
      }
    }
    f16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (f16)(acc[k] / count);
    *reinterpret_cast<f16x8*>(out + (size_t)q * C + v * 8) = o;
  }
}

__global__ void roi_align_bwd_kernel(const f16* __restrict__ dout, const float* __restrict__ rois, float* __restrict__ dfeat, int R,
                                     int H, int W, int C, int PH, int PW, float scale, int sr) {
  const int64_t total = (int64_t)R * PH * PW * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    int64_t q = i / C;
    int pw = (int)(q % PW);
    int ph = (int)((q / PW) % PH);
    int r = (int)(q / ((int64_t)PW * PH));
    const float* roi = rois + (size_t)r * 5;
    int n = (int)roi[0];
    float rsw = roi[1] * scale, rsh = roi[2] * scale, rew = roi[3] * scale, reh = roi[4] * scale;
    float rw = fmaxf(rew - rsw, 1.f), rh = fmaxf(reh - rsh, 1.f);
    float bh = rh / (float)PH, bw = rw / (float)PW;
    int gh = sr > 0 ? sr : (int)ceilf(rh / (float)PH);
    int gw = sr > 0 ? sr : (int)ceilf(rw / (float)PW);
    float count = fmaxf((float)(gh * gw), 1.f);
    float g = (float)dout[i] / count;
    if (g == 0.f) continue;
    float* fb = dfeat + (size_t)n * H * W * C + c;
    for (int iy = 0; iy < gh; ++iy) {
      float y = rsh + (float)ph * bh + ((float)iy + .5f) * bh / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        float x = rsw + (float)pw * bw + ((float)ix + .5f) * bw / (float)gw;
        Bilin b = bilin_setup(y, x, H, W);
        if (!b.valid) continue;
        atomicAdd(fb + ((size_t)b.yl * W + b.xl) * C, g * b.w1);
        atomicAdd(fb + ((size_t)b.yl * W + b.xh) * C, g * b.w2);
        atomicAdd(fb + ((size_t)b.yh * W + b.xl) * C, g * b.w3);
        atomicAdd(fb + ((size_t)b.yh * W + b.xh) * C, g * b.w4);
      }
    }
  }
}

struct MLFeat {
  const f16* f[4];
  float* df[4];
  int H[4], W[4];
  float scale[4];
};

__global__ void roi_align_ml_kernel(MLFeat ml, const float* __restrict__ rois, const int* __restrict__ level, f16* __restrict__ out,
                                    int R, int C, int PH, int PW, int sr) {
  const int vecs = C / 8;
  const int64_t total = (int64_t)R * PH * PW * vecs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int v = (int)(i % vecs);
    int64_t q = i / vecs;
    int pw = (int)(q % PW);
    int ph = (int)((q / PW) % PH);
    int r = (int)(q / ((int64_t)PW * PH));
    int l = level[r];
    const int H = ml.H[l], W = ml.W[l];
    const float scale = ml.scale[l];
    const float* roi = rois + (size_t)r * 5;
    int n = (int)roi[0];
    float rsw = roi[1] * scale, rsh = roi[2] * scale, rew = roi[3] * scale, reh = roi[4] * scale;
    float rw = fmaxf(rew - rsw, 1.f), rh = fmaxf(reh - rsh, 1.f);
    float bh = rh / (float)PH, bw = rw / (float)PW;
    int gh = sr > 0 ? sr : (int)ceilf(rh / (float)PH);
    int gw = sr > 0 ? sr : (int)ceilf(rw / (float)PW);
    float count = fmaxf((float)(gh * gw), 1.f);
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    const f16* fb = ml.f[l] + (size_t)n * H * W * C + v * 8;
    if (sr == 2) {
      // the detector's setting (MultiScaleRoIAlign(sampling_ratio=2)): all 16 taps of the 2x2 sample grid are requested
      // before any is used -- the generic loop below waits for each sample's 4 loads in turn (4 dependent L2 round trips
      // per output vector: the launch was latency-bound at 9x its output-write time).  Same accumulation order.
      Bilin bs[4];
      f16x8 t1[4], t2[4], t3[4], t4[4];
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx) {
        const int iy = sidx >> 1, ix = sidx & 1;
        const float y = rsh + (float)ph * bh + ((float)iy + .5f) * bh / 2.f;
        const float x = rsw + (float)pw * bw + ((float)ix + .5f) * bw / 2.f;
        bs[sidx] = bilin_setup(y, x, H, W);
        if (bs[sidx].valid) {
          t1[sidx] = *reinterpret_cast<const f16x8*>(fb + ((size_t)bs[sidx].yl * W + bs[sidx].xl) * C);
          t2[sidx] = *reinterpret_cast<const f16x8*>(fb + ((size_t)bs[sidx].yl * W + bs[sidx].xh) * C);
          t3[sidx] = *reinterpret_cast<const f16x8*>(fb + ((size_t)bs[sidx].yh * W + bs[sidx].xl) * C);
          t4[sidx] = *reinterpret_cast<const f16x8*>(fb + ((size_t)bs[sidx].yh * W + bs[sidx].xh) * C);
        }
      }
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx) {
        if (!bs[sidx].valid) continue;
        const Bilin b = bs[sidx];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          acc[k] += b.w1 * (float)t1[sidx][k] + b.w2 * (float)t2[sidx][k] + b.w3 * (float)t3[sidx][k] + b.w4 * (float)t4[sidx][k];
      }
      f16x8 o2;
#pragma unroll
      for (int k = 0; k < 8; ++k) o2[k] = (f16)(acc[k] / count);
      *reinterpret_cast<f16x8*>(out + (size_t)q * C + v * 8) = o2;
      continue;
    }
    for (int iy = 0; iy < gh; ++iy) {
      float y = rsh + (float)ph * bh + ((float)iy + .5f) * bh / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        float x = rsw + (float)pw * bw + ((float)ix + .5f) * bw / (float)gw;
        Bilin b = bilin_setup(y, x, H, W);
        if (!b.valid) continue;
        f16x8 v1 = *reinterpret_cast<const f16x8*>(fb + ((size_t)b.yl * W + b.xl) * C);
        f16x8 v2 = *reinterpret_cast<const f16x8*>(fb + ((size_t)b.yl * W + b.xh) * C);
        f16x8 v3 = *reinterpret_cast<const f16x8*>(fb + ((size_t)b.yh * W + b.xl) * C);
        f16x8 v4 = *reinterpret_cast<const f16x8*>(fb + ((size_t)b.yh * W + b.xh) * C);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += b.w1 * (float)v1[k] + b.w2 * (float)v2[k] + b.w3 * (float)v3[k] + b.w4 * (float)v4[k];
      }
    }
    f16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (f16)(acc[k] / count);
    *reinterpret_cast<f16x8*>(out + (size_t)q * C + v * 8) = o;
  }
}

// The detector's pooler (7x7 bins, sampling_ratio 2) with ONE BLOCK PER RoI: a thread owns one bin column pw and one 8-channel
// vector and walks the seven bin rows.  roi_align_ml_kernel above spends its time in the VALU, not in memory: every thread
// re-derives the RoI geometry, four 2-D sample set-ups and sixteen 64-bit tap addresses per output vector (the index split alone
// is a dozen quarter-rate integer multiplies).  Here the RoI geometry is block-uniform (scalar loads), the two x samples of a
// thread are set up once, a bin row costs two 1-D y set-ups, and a tap address is one 32-bit add of a row and a column offset
// onto a uniform base.  The sixteen taps of a bin are accumulated in the same order, each as one fused multiply-add that reads the
// fp16 element directly (v_fma_mix_f32: 128 VALU instructions per output vector where convert + multiply + add took 256): a result
// differs from roi_align_ml_kernel's by the roundings the fusion removes -- at most one fp16 ulp of the output (tests/test_kernels_gpu.py).
struct Samp1 {
  unsigned lo, hi;      // element offsets of the two taps along this axis
  float l, h;           // weights of hi / lo tap
  bool valid;
};

__device__ __forceinline__ Samp1 samp1_setup(float y, int H, unsigned pitch) {
  Samp1 s;
  s.valid = !(y < -1.0f || y > (float)H);
  if (y <= 0.f) y = 0.f;
  int yl = (int)y, yh;
  if (yl >= H - 1) {
    yh = yl = H - 1;
    y = (float)yl;
  } else
    yh = yl + 1;
  s.l = y - (float)yl;
  s.h = 1.f - s.l;
  s.lo = (unsigned)yl * pitch;
  s.hi = (unsigned)yh * pitch;
  return s;
}

__global__ __launch_bounds__(256) void roi_align_ml_roi7_kernel(MLFeat ml, const float* __restrict__ rois, const int* __restrict__ level,
                                                                f16* __restrict__ out, int C, int lvecs) {
  const int r = blockIdx.x;
  const int l = level[r];
  const int H = ml.H[l], W = ml.W[l];
  const float scale = ml.scale[l];
  const float* roi = rois + (size_t)r * 5;
  const int n = (int)roi[0];
  const float rsw = roi[1] * scale, rsh = roi[2] * scale, rew = roi[3] * scale, reh = roi[4] * scale;
  const float rw = fmaxf(rew - rsw, 1.f), rh = fmaxf(reh - rsh, 1.f);
  const float bh = rh / 7.f, bw = rw / 7.f;
  const int t = threadIdx.x;
  const int v = t & ((1 << lvecs) - 1);
  const int pw = t >> lvecs;
  if (pw >= 7) return;
  const f16* fb = ml.f[l] + (size_t)n * H * W * C;
  f16* ob = out + (size_t)r * 49 * C;
  Samp1 xs[2];
#pragma unroll
  for (int ix = 0; ix < 2; ++ix) {
    const float x = rsw + (float)pw * bw + ((float)ix + .5f) * bw / 2.f;
    xs[ix] = samp1_setup(x, W, (unsigned)C);
    xs[ix].lo += (unsigned)v * 8u;
    xs[ix].hi += (unsigned)v * 8u;
  }
  const unsigned rowpitch = (unsigned)W * (unsigned)C;
#pragma unroll 1
  for (int ph = 0; ph < 7; ++ph) {
    Samp1 ys[2];
#pragma unroll
    for (int iy = 0; iy < 2; ++iy) {
      const float y = rsh + (float)ph * bh + ((float)iy + .5f) * bh / 2.f;
      ys[iy] = samp1_setup(y, H, rowpitch);
    }
    f16x8 t1[4], t2[4], t3[4], t4[4];
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
      const Samp1 a = ys[sidx >> 1], b = xs[sidx & 1];      // (clamped taps are always inside the map: loaded whether valid or not)
      t1[sidx] = *reinterpret_cast<const f16x8*>(fb + (a.lo + b.lo));
      t2[sidx] = *reinterpret_cast<const f16x8*>(fb + (a.lo + b.hi));
      t3[sidx] = *reinterpret_cast<const f16x8*>(fb + (a.hi + b.lo));
      t4[sidx] = *reinterpret_cast<const f16x8*>(fb + (a.hi + b.hi));
    }
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
      const Samp1 a = ys[sidx >> 1], b = xs[sidx & 1];
      if (!(a.valid && b.valid)) continue;
      const float w1 = a.h * b.h, w2 = a.h * b.l, w3 = a.l * b.h, w4 = a.l * b.l;
#define HD_ROI_MAC(k) acc[k] = mac_tap<k>(w4, t4[sidx], mac_tap<k>(w3, t3[sidx], mac_tap<k>(w2, t2[sidx], mac_tap<k>(w1, t1[sidx], acc[k]))));
      HD_ROI_MAC(0) HD_ROI_MAC(1) HD_ROI_MAC(2) HD_ROI_MAC(3) HD_ROI_MAC(4) HD_ROI_MAC(5) HD_ROI_MAC(6) HD_ROI_MAC(7)
#undef HD_ROI_MAC
    }
    f16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (f16)(acc[k] * 0.25f);        // == acc / 4 bit for bit
    *reinterpret_cast<f16x8*>(ob + ((unsigned)(ph * 7 + pw) * (unsigned)C + (unsigned)v * 8u)) = o;
  }
}

struct RoiGeom {
  int n, l, H, W, y0, x0, ph_, pw_, gh, gw;
  float rsw, rsh, bh, bw, count;
  bool any;
};

__device__ __forceinline__ RoiGeom roi_geom(const MLFeat& ml, const float* rois, const int* level, int r, int PH, int PW, int sr) {
  RoiGeom g;
  g.l = level[r];
  g.H = ml.H[g.l];
  g.W = ml.W[g.l];
  const float scale = ml.scale[g.l];
  const float* roi = rois + (size_t)r * 5;
  g.n = (int)roi[0];
  g.rsw = roi[1] * scale;
  g.rsh = roi[2] * scale;
  float rew = roi[3] * scale, reh = roi[4] * scale;
  float rw = fmaxf(rew - g.rsw, 1.f), rh = fmaxf(reh - g.rsh, 1.f);
  g.bh = rh / (float)PH;
  g.bw = rw / (float)PW;
  g.gh = sr > 0 ? sr : (int)ceilf(rh / (float)PH);
  g.gw = sr > 0 ? sr : (int)ceilf(rw / (float)PW);
  g.count = fmaxf((float)(g.gh * g.gw), 1.f);
  // extent of the bilinear footprints (same clamping as bilin_setup)
  float ymin = g.rsh + .5f * g.bh / (float)g.gh, ymax = g.rsh + (float)(PH - 1) * g.bh + ((float)g.gh - .5f) * g.bh / (float)g.gh;
  float xmin = g.rsw + .5f * g.bw / (float)g.gw, xmax = g.rsw + (float)(PW - 1) * g.bw + ((float)g.gw - .5f) * g.bw / (float)g.gw;
  g.any = !(ymax < -1.f || ymin > (float)g.H || xmax < -1.f || xmin > (float)g.W);
  int y0 = (int)floorf(fmaxf(ymin, 0.f)), y1 = (int)floorf(fmaxf(ymax, 0.f)) + 1;
  int x0 = (int)floorf(fmaxf(xmin, 0.f)), x1 = (int)floorf(fmaxf(xmax, 0.f)) + 1;
  y0 = min(max(y0, 0), g.H - 1); y1 = min(max(y1, 0), g.H - 1);
  x0 = min(max(x0, 0), g.W - 1); x1 = min(max(x1, 0), g.W - 1);
  g.y0 = y0; g.x0 = x0; g.ph_ = y1 - y0 + 1; g.pw_ = x1 - x0 + 1;
  return g;
}

__global__ void roi_align_ml_bwd_kernel(MLFeat ml, const f16* __restrict__ dout, const float* __restrict__ rois,
                                        const int* __restrict__ level, int R, int C, int PH, int PW, int sr, int px_lo) {
  const int64_t total = (int64_t)R * PH * PW * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    int64_t q = i / C;
    int pw = (int)(q % PW);
    int ph = (int)((q / PW) % PH);
    int r = (int)(q / ((int64_t)PW * PH));
    if (px_lo > 0) {   // RoIs whose patch fits on chip were handled by the patch kernels
      const RoiGeom gg = roi_geom(ml, rois, level, r, PH, PW, sr);
      if (gg.any && gg.ph_ * gg.pw_ <= px_lo) continue;
    }
    int l = level[r];
    const int H = ml.H[l], W = ml.W[l];
    const float scale = ml.scale[l];
    const float* roi = rois + (size_t)r * 5;
    int n = (int)roi[0];
    float rsw = roi[1] * scale, rsh = roi[2] * scale, rew = roi[3] * scale, reh = roi[4] * scale;
    float rw = fmaxf(rew - rsw, 1.f), rh = fmaxf(reh - rsh, 1.f);
    float bh = rh / (float)PH, bw = rw / (float)PW;
    int gh = sr > 0 ? sr : (int)ceilf(rh / (float)PH);
    int gw = sr > 0 ? sr : (int)ceilf(rw / (float)PW);
    float count = fmaxf((float)(gh * gw), 1.f);
    float g = (float)dout[i] / count;
    if (g == 0.f) continue;
    float* fb = ml.df[l] + (size_t)n * H * W * C + c;
    for (int iy = 0; iy < gh; ++iy) {
      float y = rsh + (float)ph * bh + ((float)iy + .5f) * bh / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        float x = rsw + (float)pw * bw + ((float)ix + .5f) * bw / (float)gw;
        Bilin b = bilin_setup(y, x, H, W);
        if (!b.valid) continue;
        atomicAdd(fb + ((size_t)b.yl * W + b.xl) * C, g * b.w1);
        atomicAdd(fb + ((size_t)b.yl * W + b.xh) * C, g * b.w2);
        atomicAdd(fb + ((size_t)b.yh * W + b.xl) * C, g * b.w3);
        atomicAdd(fb + ((size_t)b.yh * W + b.xh) * C, g * b.w4);
      }
    }
  }
}

// ---- RoIAlign backward with on-chip pre-accumulation ---------------------------------------------------------------
// Global float atomics run at one chip-wide rate (~1.3 TB/s of added bytes), and the direct kernel issues
// 7*7*4*4 = 784 of them per RoI and channel.  A RoI only touches the (h+2)x(w+2) feature pixels under it, so a block
// first accumulates one RoI x 64 channels into an LDS patch (ds_add_f32) and then flushes each touched pixel ONCE:
// npx atomics instead of 784 (3-10x fewer for the RoIs this path takes).  RoIs whose patch does not fit the class
// budget are left to the direct kernel (same arithmetic).
// grid (R, C/64), block = ONE wave: lane = channel.  The wave walks the RoI's sample points in order and accumulates into
// its private LDS patch with plain read-modify-write (LDS operations of one wave execute in order, so no atomics and
// no barriers are needed).  Handles RoIs with px_lo < patch pixels <= MAXPX.
template <int MAXPX>
__global__ __launch_bounds__(64) void roi_align_ml_bwd_patch_kernel(MLFeat ml, const f16* __restrict__ dout, const float* __restrict__ rois,
                                                                    const int* __restrict__ level, int R, int C, int PH, int PW,
                                                                    int sr, int px_lo) {
  __shared__ float patch[MAXPX * 64];
  const int r = blockIdx.x;
  const RoiGeom g = roi_geom(ml, rois, level, r, PH, PW, sr);
  const int npx = g.ph_ * g.pw_;
  if (!g.any || npx <= px_lo || npx > MAXPX) return;          // uniform
  const int lane = threadIdx.x;
  const int c = blockIdx.y * 64 + lane;
  for (int i = 0; i < npx; ++i) patch[i * 64 + lane] = 0.f;
  const f16* dr = dout + (size_t)r * PH * PW * C + c;
  for (int ph = 0; ph < PH; ++ph) {
    for (int pw = 0; pw < PW; ++pw) {
      const float gv = (float)dr[(size_t)(ph * PW + pw) * C] / g.count;
      for (int iy = 0; iy < g.gh; ++iy) {
        const float y = g.rsh + (float)ph * g.bh + ((float)iy + .5f) * g.bh / (float)g.gh;
        for (int ix = 0; ix < g.gw; ++ix) {
          const float x = g.rsw + (float)pw * g.bw + ((float)ix + .5f) * g.bw / (float)g.gw;
          const Bilin b = bilin_setup(y, x, g.H, g.W);
          if (!b.valid) continue;                                 // uniform
          float* p0 = patch + ((b.yl - g.y0) * g.pw_ + (b.xl - g.x0)) * 64 + lane;
          const int dy = (b.yh - b.yl) * g.pw_ * 64, dx = (b.xh - b.xl) * 64;
          p0[0] += gv * b.w1;
          p0[dx] += gv * b.w2;
          p0[dy] += gv * b.w3;
          p0[dy + dx] += gv * b.w4;
        }
      }
    }
  }
  float* fb = ml.df[g.l] + (size_t)g.n * g.H * g.W * C + c;
  for (int py = 0; py < g.ph_; ++py)
    for (int px = 0; px < g.pw_; ++px) {
      const float v = patch[(py * g.pw_ + px) * 64 + lane];
      if (v != 0.f) atomicAdd(fb + ((size_t)(g.y0 + py) * g.W + (g.x0 + px)) * C, v);
    }
}

// ---- RoIAlign backward, gather form -------------------------------------------------------------------------------
// One thread owns ONE feature pixel (and CCH channels) and pulls the gradient from every RoI bin that touches it: no
// atomics, no zero-initialised fp32 maps, every output element is written exactly once (fp32 accumulate, fp16 store)
// and the result is bit-reproducible run to run.  The bilinear weight of a sample point is separable, and so is the sum
// over the SRxSR sample points of a bin:  w(pixel <- bin (ph,pw)) = WY[ph] * WX[pw] / count  with
// WY[ph] = sum over the bin's sample rows of the row weight of pixel row py (same clamping rules as the forward).
// Block = 16x16 pixel tile of one (level, image) x CCH channels.  The block first scans the RoI list (any order) for
// RoIs of its image / level whose touched-pixel box meets the tile (ordered compaction -> deterministic summation
// order), then every thread walks that list.
struct GatherRoi {
  float rsw, rsh, bw, bh;
  int r;
  short y0, y1, x0, x1;
};

template <int SR>
__device__ __forceinline__ float axis_weight(float start, float bin, int p, int pix, int size) {
  // sum over the SR sample coordinates of bin `p` of the weight with which they hit pixel `pix` along one axis
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < SR; ++i) {
    float y = start + (float)p * bin + ((float)i + .5f) * bin / (float)SR;
    if (y < -1.0f || y > (float)size) continue;
    if (y <= 0.f) y = 0.f;
    int yl = (int)y, yh;
    if (yl >= size - 1) {
      yh = yl = size - 1;
      y = (float)yl;
    } else
      yh = yl + 1;
    const float ly = y - (float)yl, hy = 1.f - ly;
    acc += (yl == pix ? hy : 0.f) + (yh == pix ? ly : 0.f);
  }
  return acc;
}

template <int PH, int PW, int SR, int CCH>
__global__ __launch_bounds__(256) void roi_align_ml_bwd_gather_kernel(MLFeat ml, f16* const* __restrict__ dst_unused, const f16* __restrict__ dout,
                                                                      const float* __restrict__ rois, const int* __restrict__ level,
                                                                      int R, int C, int L, int4 tile_base, int n_images) {
  constexpr int CAP = 1024;
  __shared__ GatherRoi list[CAP];
  __shared__ int wcnt[4];
  // ---- which tile
  int b = blockIdx.x, l = 0;
  const int bases[4] = {tile_base.x, tile_base.y, tile_base.z, tile_base.w};
  while (l + 1 < L && b >= bases[l + 1]) ++l;
  b -= bases[l];
  const int H = ml.H[l], W = ml.W[l];
  const int tw = (W + 15) / 16, th = (H + 15) / 16;
  const int n = b / (tw * th);
  const int ty = (b / tw) % th, tx = b % tw;
  const int c0 = blockIdx.y * CCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int py = ty * 16 + (tid >> 4), px = tx * 16 + (tid & 15);
  const bool live = py < H && px < W;
  const int ty0 = ty * 16, ty1 = min(ty * 16 + 15, H - 1), tx0 = tx * 16, tx1 = min(tx * 16 + 15, W - 1);
  float acc[CCH];
#pragma unroll
  for (int k = 0; k < CCH; ++k) acc[k] = 0.f;

  for (int base = 0; base < R; base += CAP) {
    // ---- ordered compaction of the RoIs [base, base+CAP) that touch this tile
    int total = 0;
    for (int it = 0; it < CAP / 256; ++it) {
      const int r = base + it * 256 + tid;
      bool hit = false;
      GatherRoi e;
      if (r < R && level[r] == l && (int)rois[(size_t)r * 5] == n) {
        const RoiGeom g = roi_geom(ml, rois, level, r, PH, PW, SR);
        if (g.any) {
          const int y1 = g.y0 + g.ph_ - 1, x1 = g.x0 + g.pw_ - 1;
          hit = !(y1 < ty0 || g.y0 > ty1 || x1 < tx0 || g.x0 > tx1);
          e.rsw = g.rsw; e.rsh = g.rsh; e.bw = g.bw; e.bh = g.bh; e.r = r;
          e.y0 = (short)g.y0; e.y1 = (short)y1; e.x0 = (short)g.x0; e.x1 = (short)x1;
        }
      }
      const unsigned long long m = __ballot(hit);
      if (lane == 0) wcnt[wave] = __popcll(m);
      __syncthreads();
      int off = total;
      for (int w = 0; w < wave; ++w) off += wcnt[w];
      if (hit) list[off + __popcll(m & ((1ull << lane) - 1ull))] = e;
      total += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
      __syncthreads();
    }
    // ---- every pixel walks the list
    if (live) {
      for (int i = 0; i < total; ++i) {
        const GatherRoi e = list[i];
        if (py < e.y0 || py > e.y1 || px < e.x0 || px > e.x1) continue;
        float wy[PH], wx[PW];
        bool anyy = false, anyx = false;
#pragma unroll
        for (int p = 0; p < PH; ++p) {
          // bin p's sample rows lie in [rsh + p*bh, rsh + (p+1)*bh]; they reach pixel row py only within one pixel of it
          const float lo = e.rsh + (float)p * e.bh, hi = lo + e.bh;
          const bool near_ = !(hi < (float)py - 1.f || lo > (float)py + 1.f) || py == 0 || py == H - 1;
          wy[p] = near_ ? axis_weight<SR>(e.rsh, e.bh, p, py, H) : 0.f;
          anyy |= wy[p] != 0.f;
        }
        if (!anyy) continue;
#pragma unroll
        for (int p = 0; p < PW; ++p) {
          const float lo = e.rsw + (float)p * e.bw, hi = lo + e.bw;
          const bool near_ = !(hi < (float)px - 1.f || lo > (float)px + 1.f) || px == 0 || px == W - 1;
          wx[p] = near_ ? axis_weight<SR>(e.rsw, e.bw, p, px, W) : 0.f;
          anyx |= wx[p] != 0.f;
        }
        if (!anyx) continue;
        const f16* dr = dout + (size_t)e.r * PH * PW * C + c0;
#pragma unroll
        for (int ph = 0; ph < PH; ++ph) {
          if (wy[ph] == 0.f) continue;
#pragma unroll
          for (int pw = 0; pw < PW; ++pw) {
            const float w = wy[ph] * wx[pw] * (1.f / (float)(SR * SR));
            if (w == 0.f) continue;
            const f16* q = dr + (size_t)(ph * PW + pw) * C;
#pragma unroll
            for (int v = 0; v < CCH / 8; ++v) {
              const f16x8 d = *reinterpret_cast<const f16x8*>(q + v * 8);
              mac_vec(w, d, &acc[v * 8]);
            }
          }
        }
      }
    }
    __syncthreads();
  }
  if (live) {
    f16* o = const_cast<f16*>(ml.f[l]) + (((size_t)n * H + py) * W + px) * C + c0;
#pragma unroll
    for (int v = 0; v < CCH / 8; ++v) {
      f16x8 t;
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = (f16)acc[v * 8 + k];
      *reinterpret_cast<f16x8*>(o + v * 8) = t;
    }
  }
}

// C % 256 == 0 form of the gather backward (the detector's 256-channel pyramid): 8x8 pixel tile, thread = (pixel, 64-channel
// quarter).  The per-axis bin weights depend on (RoI, pixel ROW) and (RoI, pixel COLUMN) only, so a tile needs
// 16 x 7 of them per RoI instead of 64 x 14: they are computed cooperatively into LDS for 32 RoIs at a time and every
// pixel thread then reads its 14 numbers -- the weight arithmetic, which dominated the generic kernel above, drops ~8x
// and is no longer repeated per channel chunk.
template <int PH, int PW, int SR, bool FUSED>
__global__ __launch_bounds__(256) void roi_align_ml_bwd_gather256_kernel(MLFeat ml, const f16* __restrict__ dout, const float* __restrict__ rois,
                                                                         const int* __restrict__ level, int R, int C, int L, int4 tile_base) {
  constexpr int CAP = 1024, SB = 32, TS = 8, CQ = 64;
  static_assert(PH == PW, "square pooler");
  __shared__ GatherRoi list[CAP];
  __shared__ float wtab[SB][2][TS][PH];
  __shared__ int wcnt[4];
  int b = blockIdx.x, l = 0;
  const int bases[4] = {tile_base.x, tile_base.y, tile_base.z, tile_base.w};
  while (l + 1 < L && b >= bases[l + 1]) ++l;
  b -= bases[l];
  const int H = ml.H[l], W = ml.W[l];
  const int tw = (W + TS - 1) / TS, th = (H + TS - 1) / TS;
  const int n = b / (tw * th);
  const int ty = (b / tw) % th, tx = b % tw;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ly = lane >> 3, lx = lane & 7;
  const int py = ty * TS + ly, px = tx * TS + lx;
  const bool live = py < H && px < W;
  const int c0 = blockIdx.y * 256 + wave * CQ;
  const int ty0 = ty * TS, ty1 = min(ty * TS + TS - 1, H - 1), tx0 = tx * TS, tx1 = min(tx * TS + TS - 1, W - 1);
  float acc[CQ];
#pragma unroll
  for (int k = 0; k < CQ; ++k) acc[k] = 0.f;

  for (int base = 0; base < R; base += CAP) {
    int total = 0;
    for (int it = 0; it < CAP / 256; ++it) {
      const int r = base + it * 256 + tid;
      bool hit = false;
      GatherRoi e;
      if (r < R && level[r] == l && (int)rois[(size_t)r * 5] == n) {
        const RoiGeom g = roi_geom(ml, rois, level, r, PH, PW, SR);
        if (g.any) {
          const int y1 = g.y0 + g.ph_ - 1, x1 = g.x0 + g.pw_ - 1;
          hit = !(y1 < ty0 || g.y0 > ty1 || x1 < tx0 || g.x0 > tx1);
          e.rsw = g.rsw; e.rsh = g.rsh; e.bw = g.bw; e.bh = g.bh; e.r = r;
          e.y0 = (short)g.y0; e.y1 = (short)y1; e.x0 = (short)g.x0; e.x1 = (short)x1;
        }
      }
      const unsigned long long m = __ballot(hit);
      if (lane == 0) wcnt[wave] = __popcll(m);
      __syncthreads();
      int off = total;
      for (int w = 0; w < wave; ++w) off += wcnt[w];
      if (hit) list[off + __popcll(m & ((1ull << lane) - 1ull))] = e;
      total += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
      __syncthreads();
    }
    for (int sb = 0; sb < total; sb += SB) {
      const int nsb = min(SB, total - sb);
      // ---- phase A: bin weights of nsb RoIs for the tile's 8 rows and 8 columns
      for (int idx = tid; idx < nsb * 2 * TS * PH; idx += 256) {
        const int p = idx % PH;
        const int pos = (idx / PH) % TS;
        const int axis = (idx / (PH * TS)) & 1;
        const int j = idx / (2 * TS * PH);
        const GatherRoi e = list[sb + j];
        const float start = axis ? e.rsw : e.rsh, bin = axis ? e.bw : e.bh;
        const int pix = axis ? tx * TS + pos : ty * TS + pos, size = axis ? W : H;
        const float lo = start + (float)p * bin, hi = lo + bin;
        const bool near_ = pix < size && (!(hi < (float)pix - 1.f || lo > (float)pix + 1.f) || pix == 0 || pix == size - 1);
        wtab[j][axis][pos][p] = near_ ? axis_weight<SR>(start, bin, p, pix, size) : 0.f;
      }
      __syncthreads();
      // ---- phase B
      if (live) {
        for (int j = 0; j < nsb; ++j) {
          const GatherRoi e = list[sb + j];
          if (py < e.y0 || py > e.y1 || px < e.x0 || px > e.x1) continue;
          // the bins that reach a pixel are contiguous along each axis: find the two ranges, then a small 2-D loop
          // (kept rolled: 49 unrolled bin bodies of 64 FMAs each overflow the instruction cache)
          const float* WY = &wtab[j][0][ly][0];
          const float* WX = &wtab[j][1][lx][0];
          int ya = -1, yb = -1, xa = -1, xb = -1;
#pragma unroll
          for (int p = 0; p < PH; ++p) {
            if (WY[p] != 0.f) { if (ya < 0) ya = p; yb = p; }
            if (WX[p] != 0.f) { if (xa < 0) xa = p; xb = p; }
          }
          if (ya < 0 || xa < 0) continue;
          const f16* dr = dout + (size_t)e.r * PH * PW * C + c0;
          for (int ph = ya; ph <= yb; ++ph) {
            const float wyv = WY[ph] * (1.f / (float)(SR * SR));
            for (int pw = xa; pw <= xb; ++pw) {
              const float w = wyv * WX[pw];
              if (w == 0.f) continue;
              const f16* q = dr + (size_t)(ph * PW + pw) * C;
#pragma unroll
              for (int v = 0; v < CQ / 8; ++v) {
                const f16x8 d = *reinterpret_cast<const f16x8*>(q + v * 8);
                mac_vec<FUSED>(w, d, &acc[v * 8]);
              }
            }
          }
        }
      }
      __syncthreads();
    }
  }
  if (live) {
    f16* o = const_cast<f16*>(ml.f[l]) + (((size_t)n * H + py) * W + px) * C + c0;
#pragma unroll
    for (int v = 0; v < CQ / 8; ++v) {
      f16x8 t;
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = (f16)acc[v * 8 + k];
      *reinterpret_cast<f16x8*>(o + v * 8) = t;
    }
  }
}

}  // namespace

extern "C" int HD_API(hd_roi_align)(const void* feat, const float* rois, void* out, int R, int N, int H, int W, int C, int PH, int PW,
                            float spatial_scale, int sampling_ratio, void* stream) {
  HD_CHECK_ARG(feat && rois && out && R >= 0 && C % 8 == 0 && N > 0, "hd_roi_align: bad args");
  if (R == 0) return HD_OK;
  int64_t total = (int64_t)R * PH * PW * C / 8;
  int g = (int)((total + 255) / 256);
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(roi_align_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, (const f16*)feat, rois, (f16*)out, R, H, W, C, PH, PW,
                     spatial_scale, sampling_ratio);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_roi_align_bwd)(const void* dout, const float* rois, float* dfeat_f32, int R, int N, int H, int W, int C, int PH,
                                int PW, float spatial_scale, int sampling_ratio, void* stream) {
  HD_CHECK_ARG(dout && rois && dfeat_f32 && R >= 0 && N > 0, "hd_roi_align_bwd: bad args");
  if (R == 0) return HD_OK;
  int64_t total = (int64_t)R * PH * PW * C;
  int g = (int)((total + 255) / 256);
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(roi_align_bwd_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, (const f16*)dout, rois, dfeat_f32, R, H, W, C, PH,
                     PW, spatial_scale, sampling_ratio);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

// Round 4: the same gather with the lanes of a wave laid along the CHANNELS.  Above, the 64 lanes of a wave are 64 different pixels,
// so one 16-byte load instruction touches up to 64 cache lines (a lane walks its own 128-byte run over eight instructions) and the
// wave runs the union of 64 pixels' bin ranges.  Here a wave is one tile ROW: lane = (pixel column, 16-byte channel slot), a thread
// owns the four slots cv, cv + 8, cv + 16, cv + 24 of a 256-channel chunk, so every load instruction reads eight full 128-byte lines,
// the eight pixels of a wave share their row's bin range, and a thread carries 32 accumulators instead of 64.  512 threads per 8x8
// tile; list building, weight table and summation order per element are those of the kernel above (equal results).
template <int PH, int PW, int SR>
__global__ __launch_bounds__(512) void roi_align_ml_bwd_gather256_rows_kernel(MLFeat ml, const f16* __restrict__ dout, const float* __restrict__ rois,
                                                                              const int* __restrict__ level, int R, int C, int L, int4 tile_base) {
  constexpr int CAP = 1024, SB = 32, TS = 8, NT = 512;
  static_assert(PH == PW, "square pooler");
  __shared__ GatherRoi list[CAP];
  __shared__ float wtab[SB][2][TS][PH];
  __shared__ int wcnt[NT / 64];
  int b = blockIdx.x, l = 0;
  const int bases[4] = {tile_base.x, tile_base.y, tile_base.z, tile_base.w};
  while (l + 1 < L && b >= bases[l + 1]) ++l;
  b -= bases[l];
  const int H = ml.H[l], W = ml.W[l];
  const int tw = (W + TS - 1) / TS, th = (H + TS - 1) / TS;
  const int n = b / (tw * th);
  const int ty = (b / tw) % th, tx = b % tw;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ly = wave, lx = lane >> 3, cv = lane & 7;
  const int py = ty * TS + ly, px = tx * TS + lx;
  const bool live = py < H && px < W;
  const int c0 = blockIdx.y * 256 + cv * 8;
  const int ty0 = ty * TS, ty1 = min(ty * TS + TS - 1, H - 1), tx0 = tx * TS, tx1 = min(tx * TS + TS - 1, W - 1);
  float acc[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) acc[k] = 0.f;

  for (int base = 0; base < R; base += CAP) {
    int total = 0;
    for (int it = 0; it < CAP / NT; ++it) {
      const int r = base + it * NT + tid;
      bool hit = false;
      GatherRoi e;
      if (r < R && level[r] == l && (int)rois[(size_t)r * 5] == n) {
        const RoiGeom g = roi_geom(ml, rois, level, r, PH, PW, SR);
        if (g.any) {
          const int y1 = g.y0 + g.ph_ - 1, x1 = g.x0 + g.pw_ - 1;
          hit = !(y1 < ty0 || g.y0 > ty1 || x1 < tx0 || g.x0 > tx1);
          e.rsw = g.rsw; e.rsh = g.rsh; e.bw = g.bw; e.bh = g.bh; e.r = r;
          e.y0 = (short)g.y0; e.y1 = (short)y1; e.x0 = (short)g.x0; e.x1 = (short)x1;
        }
      }
      const unsigned long long m = __ballot(hit);
      if (lane == 0) wcnt[wave] = __popcll(m);
      __syncthreads();
      int off = total, all = 0;
#pragma unroll
      for (int w = 0; w < NT / 64; ++w) {
        if (w < wave) off += wcnt[w];
        all += wcnt[w];
      }
      if (hit) list[off + __popcll(m & ((1ull << lane) - 1ull))] = e;
      total += all;
      __syncthreads();
    }
    for (int sb = 0; sb < total; sb += SB) {
      const int nsb = min(SB, total - sb);
      for (int idx = tid; idx < nsb * 2 * TS * PH; idx += NT) {
        const int p = idx % PH;
        const int pos = (idx / PH) % TS;
        const int axis = (idx / (PH * TS)) & 1;
        const int j = idx / (2 * TS * PH);
        const GatherRoi e = list[sb + j];
        const float start = axis ? e.rsw : e.rsh, bin = axis ? e.bw : e.bh;
        const int pix = axis ? tx * TS + pos : ty * TS + pos, size = axis ? W : H;
        const float lo = start + (float)p * bin, hi = lo + bin;
        const bool near_ = pix < size && (!(hi < (float)pix - 1.f || lo > (float)pix + 1.f) || pix == 0 || pix == size - 1);
        wtab[j][axis][pos][p] = near_ ? axis_weight<SR>(start, bin, p, pix, size) : 0.f;
      }
      __syncthreads();
      if (live) {
        for (int j = 0; j < nsb; ++j) {
          const GatherRoi e = list[sb + j];
          if (py < e.y0 || py > e.y1 || px < e.x0 || px > e.x1) continue;
          const float* WY = &wtab[j][0][ly][0];
          const float* WX = &wtab[j][1][lx][0];
          int ya = -1, yb = -1, xa = -1, xb = -1;
#pragma unroll
          for (int p = 0; p < PH; ++p) {
            if (WY[p] != 0.f) { if (ya < 0) ya = p; yb = p; }
            if (WX[p] != 0.f) { if (xa < 0) xa = p; xb = p; }
          }
          if (ya < 0 || xa < 0) continue;
          const f16* dr = dout + (size_t)e.r * PH * PW * C + c0;
          for (int ph = ya; ph <= yb; ++ph) {
            const float wyv = WY[ph] * (1.f / (float)(SR * SR));
            for (int pw = xa; pw <= xb; ++pw) {
              const float w = wyv * WX[pw];
              if (w == 0.f) continue;
              const f16* q = dr + (size_t)(ph * PW + pw) * C;
              const f16x8 d0 = *reinterpret_cast<const f16x8*>(q);
              const f16x8 d1 = *reinterpret_cast<const f16x8*>(q + 64);
              const f16x8 d2 = *reinterpret_cast<const f16x8*>(q + 128);
              const f16x8 d3 = *reinterpret_cast<const f16x8*>(q + 192);
              mac_vec(w, d0, &acc[0]);
              mac_vec(w, d1, &acc[8]);
              mac_vec(w, d2, &acc[16]);
              mac_vec(w, d3, &acc[24]);
            }
          }
        }
      }
      __syncthreads();
    }
  }
  if (live) {
    f16* o = const_cast<f16*>(ml.f[l]) + (((size_t)n * H + py) * W + px) * C + c0;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      f16x8 t;
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = (f16)acc[v * 8 + k];
      *reinterpret_cast<f16x8*>(o + v * 64) = t;
    }
  }
}

extern "C" int HD_API(hd_roi_align_ml)(const void* const* feats, const int* H, const int* W, const float* scale, int L, const float* rois,
                               const int* level, void* out, int R, int C, int PH, int PW, int sampling_ratio, void* stream) {
  HD_CHECK_ARG(feats && H && W && scale && rois && level && out && L >= 1 && L <= 4 && C % 8 == 0 && R >= 0, "hd_roi_align_ml: bad args");
  if (R == 0) return HD_OK;
  MLFeat ml = {};
  for (int l = 0; l < L; ++l) {
    ml.f[l] = (const f16*)feats[l];
    ml.H[l] = H[l];
    ml.W[l] = W[l];
    ml.scale[l] = scale[l];
  }
  // the hot path's pooler: one block per RoI (HD_ROI_ROWS=0 keeps the one-thread-per-output form: A/B, and the fallback for other poolers)
  static const int rows_on = getenv("HD_ROI_ROWS") ? atoi(getenv("HD_ROI_ROWS")) : 1;
  const int vecs = C / 8;
  bool small_maps = true;
  for (int l = 0; l < L; ++l) small_maps = small_maps && (int64_t)H[l] * W[l] * C < (int64_t)1 << 31;
  if (rows_on && PH == 7 && PW == 7 && sampling_ratio == 2 && (vecs & (vecs - 1)) == 0 && vecs <= 32 && small_maps) {
    int lv = 0;
    while ((1 << lv) < vecs) ++lv;
    hipLaunchKernelGGL(roi_align_ml_roi7_kernel, dim3(R), dim3((7 * vecs + 63) / 64 * 64), 0, (hipStream_t)stream, ml, rois, level, (f16*)out, C, lv);
    HD_CHECK_LAUNCH();
    return HD_OK;
  }
  int64_t total = (int64_t)R * PH * PW * C / 8;
  int g = (int)((total + 255) / 256);
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(roi_align_ml_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, ml, rois, level, (f16*)out, R, C, PH, PW, sampling_ratio);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_roi_align_ml_bwd_gather)(const void* dout, const float* rois, const int* level, void* const* dfeat_f16, const int* H,
                                          const int* W, const float* scale, int L, int R, int n_images, int C, int PH, int PW,
                                          int sampling_ratio, void* stream) {
  HD_CHECK_ARG(dout && rois && level && dfeat_f16 && H && W && scale && L >= 1 && L <= 4 && R >= 0 && n_images >= 0, "hd_roi_align_ml_bwd_gather: bad args");
  HD_CHECK_ARG(PH == 7 && PW == 7 && sampling_ratio == 2 && C % 32 == 0,
               "hd_roi_align_ml_bwd_gather: built for the 7x7 / sampling_ratio 2 pooler of the hot path (got %dx%d sr %d C %d)", PH, PW, sampling_ratio, C);
  if (n_images == 0) return HD_OK;
  MLFeat ml = {};
  int base[5] = {0, 0, 0, 0, 0};
  for (int l = 0; l < L; ++l) {
    ml.f[l] = (const f16*)dfeat_f16[l];       // destination maps ride in the `f` slots (fp16), written once per element
    ml.H[l] = H[l];
    ml.W[l] = W[l];
    ml.scale[l] = scale[l];
    const int ts = (C % 256 == 0) ? 8 : 16;
    base[l + 1] = base[l] + n_images * ((H[l] + ts - 1) / ts) * ((W[l] + ts - 1) / ts);
  }
  for (int l = L; l < 4; ++l) base[l + 1] = base[L];
  if (C % 256 == 0) {
    static const int fused_on = getenv("HD_ROI_ROWS") ? atoi(getenv("HD_ROI_ROWS")) : 1;
    if (fused_on == 1)
      hipLaunchKernelGGL((roi_align_ml_bwd_gather256_rows_kernel<7, 7, 2>), dim3(base[L], C / 256), dim3(512), 0, (hipStream_t)stream, ml,
                         (const f16*)dout, rois, level, R, C, L, make_int4(base[0], base[1], base[2], base[3]));
    else if (fused_on)
      hipLaunchKernelGGL((roi_align_ml_bwd_gather256_kernel<7, 7, 2, true>), dim3(base[L], C / 256), dim3(256), 0, (hipStream_t)stream, ml,
                         (const f16*)dout, rois, level, R, C, L, make_int4(base[0], base[1], base[2], base[3]));
    else
      hipLaunchKernelGGL((roi_align_ml_bwd_gather256_kernel<7, 7, 2, false>), dim3(base[L], C / 256), dim3(256), 0, (hipStream_t)stream, ml,
                         (const f16*)dout, rois, level, R, C, L, make_int4(base[0], base[1], base[2], base[3]));
    HD_CHECK_LAUNCH();
    return HD_OK;
  }
  hipLaunchKernelGGL((roi_align_ml_bwd_gather_kernel<7, 7, 2, 32>), dim3(base[L], C / 32), dim3(256), 0, (hipStream_t)stream, ml,
                     (f16* const*)nullptr, (const f16*)dout, rois, level, R, C, L, make_int4(base[0], base[1], base[2], base[3]), n_images);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_roi_align_ml_bwd)(const void* dout, const float* rois, const int* level, float* const* dfeat_f32, const int* H,
                                   const int* W, const float* scale, int L, int R, int C, int PH, int PW, int sampling_ratio,
                                   void* stream) {
  HD_CHECK_ARG(dout && rois && level && dfeat_f32 && H && W && scale && L >= 1 && L <= 4 && R >= 0, "hd_roi_align_ml_bwd: bad args");
  if (R == 0) return HD_OK;
  MLFeat ml = {};
  for (int l = 0; l < L; ++l) {
    ml.df[l] = dfeat_f32[l];
    ml.H[l] = H[l];
    ml.W[l] = W[l];
    ml.scale[l] = scale[l];
  }
  int64_t total = (int64_t)R * PH * PW * C;
  int g = (int)((total + 255) / 256);
  if (g > 16384) g = 16384;
  int px_lo = 0;
  if (C % 64 == 0) {
    // RoIs that touch <= 64 feature pixels: one-wave LDS patch accumulation (16 KB, 10 waves per CU), one global atomic
    // per touched pixel instead of 784 per RoI.  (A 224-pixel class was measured too: at 56 KB of LDS per wave its
    // occupancy makes it slower than the direct atomics it replaces.)
    hipLaunchKernelGGL((roi_align_ml_bwd_patch_kernel<64>), dim3(R, C / 64), dim3(64), 0, (hipStream_t)stream, ml, (const f16*)dout, rois, level, R, C, PH, PW, sampling_ratio, 0);
    px_lo = 64;
  }
  hipLaunchKernelGGL(roi_align_ml_bwd_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, ml, (const f16*)dout, rois, level, R, C, PH, PW, sampling_ratio, px_lo);
  HD_CHECK_LAUNCH();
  return HD_OK;
}
