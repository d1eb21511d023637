// fp32-STORAGE convolution, data gradient, FC, weight gradient and weight re-pack: the arithmetic of `--precision 32`, the reference's
// default (src/config/config.py:149, passed to pl.Trainer at train_hallucidet.py:507, train_detector.py:387, eval_hallucidet.py:230).
//
// NOT a fast path.  One untuned instance per operation, written for parity: the fp16 kernels of this library round every stored
// activation to 11 bits, so their agreement with the reference's fp32 evaluation can only be shown up to decisions (ReLU on / off,
// pool winners, NMS keeps) that flip inside that noise.  With activations, weights and gradients stored in fp32 and fp32 FMA
// accumulation the whole training step can be held against the CPU oracle WITHOUT shared decisions and without a rounding schedule
// (tests/test_fp32_mode_gpu.py: losses 1e-4, gradients 1e-3, identical proposal / sampler / NMS index sets).
//
// Same argument blocks as the fp16 entry points (hd_conv_args / hd_wgrad_args, include/hallucidet_hip.h) with every `f16` tensor
// read as fp32: x, x2, w, res, mask, y (out_mode 0 and 2 coincide: NHWC fp32; 1 = NCHW fp32).  Same semantics: logical input =
// concat(up1 ? nearest2x(x) : x, x2), zero-dilated by in_dil; v = acc + res + bias; v = 0 where mask <= 0; statistics of v per
// M tile ([rows][2][Cout]); y = act(v); consumer-side BatchNorm of the x source (in_scale / in_shift / in_relu).
//
// Kernel: implicit GEMM on the vector ALU, 64 x 64 output tile per 256-thread block, 16-deep K steps through LDS, 4 x 4 outputs per
// thread; K order k = tap * Cin + ci (the layout of the weights); one thread gathers four consecutive channels of one pixel at one tap
// (channel counts are multiples of 8, so a quad never straddles a tap or the concat boundary).
#include "hd_common.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 16;

struct CP {
  const float* x;
  const float* x2;
  const float* w;
  const float* bias;
  const float* res;
  const float* mask;
  float* y;
  float* stats;
  const float* in_scale;
  const float* in_shift;
  int N, Hsrc, Wsrc, Hin, Win, C1, C2, Cin, Ho, Wo, Cout, KH, KW, stride, pad, up1, in_dil, act, out_mode, in_relu;
  int M, Ktot;
};

// value quad (4 channels from ci) of the logical input at output pixel (n, ho, wo), tap (kh, kw)
__device__ __forceinline__ f32x4 gather4(const CP& p, int n, int ho, int wo, int kh, int kw, int ci, bool live) {
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  if (!live) return z;
  int hi = ho * p.stride - p.pad + kh, wi = wo * p.stride - p.pad + kw;
  if (p.in_dil > 1) {
    if (hi < 0 || wi < 0 || (hi % p.in_dil) != 0 || (wi % p.in_dil) != 0) return z;
    hi /= p.in_dil;
    wi /= p.in_dil;
    if (hi >= p.Hsrc || wi >= p.Wsrc) return z;
    return *reinterpret_cast<const f32x4*>(p.x + ((size_t)(n * p.Hsrc + hi) * p.Wsrc + wi) * p.C1 + ci);
  }
  if ((unsigned)hi >= (unsigned)p.Hin || (unsigned)wi >= (unsigned)p.Win) return z;
  if (ci < p.C1) {
    const int hs = p.up1 ? (hi >> 1) : hi, ws = p.up1 ? (wi >> 1) : wi;
    f32x4 v = *reinterpret_cast<const f32x4*>(p.x + ((size_t)(n * p.Hsrc + hs) * p.Wsrc + ws) * p.C1 + ci);
    if (p.in_scale) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float f = hd_bn_affine(v[k], p.in_scale[ci + k], p.in_shift[ci + k]);
        v[k] = p.in_relu ? fmaxf(f, 0.f) : f;
      }
    }
    return v;
  }
  return *reinterpret_cast<const f32x4*>(p.x2 + ((size_t)(n * p.Hin + hi) * p.Win + wi) * p.C2 + (ci - p.C1));
}

__global__ __launch_bounds__(256) void conv_f32_kernel(CP p) {
  __shared__ float sa[BK][BM + 4];      // [k][pixel]
  __shared__ float sb[BK][BN + 4];      // [k][cout]
  __shared__ float sred[4][BN][2];
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  // loader role: row lr (pixel / cout) and channel quad lq of the 16-deep K step
  const int lr = tid >> 2, lq = tid & 3;
  const int HoWo = p.Ho * p.Wo;
  const int pm = m0 + lr;
  const bool plive = pm < p.M;
  const int pn = plive ? pm / HoWo : 0, prem = plive ? pm - pn * HoWo : 0, pho = prem / p.Wo, pwo = prem - pho * p.Wo;
  const int wco = n0 + lr;
  // compute role: 4 pixels x 4 couts
  const int tx = tid & 15, ty = tid >> 4;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int k0 = 0; k0 < p.Ktot; k0 += BK) {
    const int k = k0 + lq * 4;
    f32x4 av = {0.f, 0.f, 0.f, 0.f}, bv = av;
    if (k < p.Ktot) {
      const int tap = k / p.Cin, ci = k - tap * p.Cin, kh = tap / p.KW, kw = tap - kh * p.KW;
      av = gather4(p, pn, pho, pwo, kh, kw, ci, plive);
      if (wco < p.Cout) bv = *reinterpret_cast<const f32x4*>(p.w + (size_t)wco * p.Ktot + k);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      sa[lq * 4 + q][lr] = av[q];
      sb[lq * 4 + q][lr] = bv[q];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; ++kk) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(&sa[kk][ty * 4]);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(&sb[kk][tx * 4]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(a4[i], b4[j], acc[i][j]);
    }
  }

  // ---- epilogue: rows ty*4 + i (pixels), columns tx*4 + j (couts)
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= p.M) continue;
    const int n = m / HoWo, rem = m - n * HoWo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = n0 + tx * 4 + j;
      if (co >= p.Cout) continue;
      float v = acc[i][j];
      if (p.res) v += p.res[(size_t)m * p.Cout + co];
      if (p.bias) v += p.bias[co];
      if (p.mask && !(p.mask[(size_t)m * p.Cout + co] > 0.f)) v = 0.f;
      s1[j] += v;
      s2[j] += v * v;
      if (p.act == HD_ACT_RELU) v = fmaxf(v, 0.f);
      else if (p.act == HD_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
      if (p.out_mode == HD_OUT_NCHW_F32) p.y[((size_t)n * p.Cout + co) * HoWo + rem] = v;
      else p.y[(size_t)m * p.Cout + co] = v;
    }
  }
  if (p.stats) {
    // fold the 16 pixel groups (ty) in a fixed order: 16 -> 4 by lane exchange inside a 64-lane wave (ty = wave*4 + (lane>>4)), then LDS
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s1[j] += __shfl_xor(s1[j], 16);
      s1[j] += __shfl_xor(s1[j], 32);
      s2[j] += __shfl_xor(s2[j], 16);
      s2[j] += __shfl_xor(s2[j], 32);
    }
    if ((tid & 63) < 16) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sred[tid >> 6][tx * 4 + j][0] = s1[j];
        sred[tid >> 6][tx * 4 + j][1] = s2[j];
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int c = tid >> 1, which = tid & 1;
      const float s = (sred[0][c][which] + sred[1][c][which]) + (sred[2][c][which] + sred[3][c][which]);
      if (n0 + c < p.Cout) p.stats[((size_t)blockIdx.x * 2 + which) * p.Cout + n0 + c] = s;
    }
  }
}

int fill(const hd_conv_args* a, CP& p) {
  HD_CHECK_ARG(a && a->x && a->w && a->y, "hd_conv2d_f32: null pointer");
  HD_CHECK_ARG(a->C1 > 0 && a->C1 % 4 == 0 && a->C2 >= 0 && a->C2 % 4 == 0, "hd_conv2d_f32: channel counts must be multiples of 4 (C1=%d C2=%d)", a->C1, a->C2);
  HD_CHECK_ARG((a->C2 == 0) == (a->x2 == nullptr), "hd_conv2d_f32: x2/C2 mismatch");
  HD_CHECK_ARG(a->N > 0 && a->Ho > 0 && a->Wo > 0 && a->Cout > 0 && a->KH > 0 && a->KW > 0 && a->stride > 0, "hd_conv2d_f32: bad extent");
  HD_CHECK_ARG(!(a->in_dil > 1 && (a->up1 || a->C2)), "hd_conv2d_f32: in_dil excludes up1/x2");
  HD_CHECK_ARG(!a->up1 || (a->Hin == 2 * a->Hsrc && a->Win == 2 * a->Wsrc), "hd_conv2d_f32: up1 needs Hin=2*Hsrc");
  HD_CHECK_ARG((int64_t)a->N * a->Ho * a->Wo < (1ll << 31), "hd_conv2d_f32: too many pixels");
  HD_CHECK_ARG((a->in_scale == nullptr) == (a->in_shift == nullptr), "hd_conv2d_f32: in_scale / in_shift must be given together");
  HD_CHECK_ARG(a->out_mode >= HD_OUT_NHWC_F16 && a->out_mode <= HD_OUT_NHWC_F32, "hd_conv2d_f32: out_mode %d", a->out_mode);
  p.x = (const float*)a->x; p.x2 = (const float*)a->x2; p.w = (const float*)a->w; p.bias = a->bias;
  p.res = (const float*)a->res; p.mask = (const float*)a->mask; p.y = (float*)a->y; p.stats = a->stats;
  p.in_scale = a->in_scale; p.in_shift = a->in_shift; p.in_relu = a->in_relu;
  p.N = a->N; p.Hsrc = a->Hsrc; p.Wsrc = a->Wsrc; p.Hin = a->Hin; p.Win = a->Win; p.C1 = a->C1; p.C2 = a->C2; p.Cin = a->C1 + a->C2;
  p.Ho = a->Ho; p.Wo = a->Wo; p.Cout = a->Cout; p.KH = a->KH; p.KW = a->KW; p.stride = a->stride; p.pad = a->pad; p.up1 = a->up1;
  p.in_dil = a->in_dil < 1 ? 1 : a->in_dil; p.act = a->act; p.out_mode = a->out_mode;
  p.M = a->N * a->Ho * a->Wo;
  p.Ktot = a->KH * a->KW * p.Cin;
  return HD_OK;
}

// ---------------------------------------------------------------- weight gradient
struct WP {
  const float* x;
  const float* x2;
  const float* dy;
  float* slab;
  const float* in_scale;
  const float* in_shift;
  int N, Hsrc, Wsrc, Hin, Win, C1, C2, Cin, Ho, Wo, Cout, KH, KW, stride, pad, up1, nsplit, in_relu;
  int M, Ktot, per;
};

// slab[s][co][k] = sum over the pixels of slice s of dY[pix][co] * X[pix @ tap(k)][ci(k)]; tile 64 (co) x 64 (k), 16 pixels per step
__global__ __launch_bounds__(256) void wgrad_f32_kernel(WP p) {
  __shared__ float sd[BK][BM + 4];      // [pixel][cout]
  __shared__ float sx[BK][BN + 4];      // [pixel][k]
  const int tid = threadIdx.x;
  const int co0 = blockIdx.x * BM, k0 = blockIdx.y * BN, s = blockIdx.z;
  const int p0 = s * p.per, p1 = min(p.M, p0 + p.per);
  const int HoWo = p.Ho * p.Wo;
  // loader role: pixel lp of the step, quad lq (16 quads = 64 columns)
  const int lp = tid >> 4, lq = tid & 15;
  const int kx = k0 + lq * 4;
  const bool klive = kx < p.Ktot;
  const int tap = klive ? kx / p.Cin : 0, ci = kx - tap * p.Cin, kh = tap / p.KW, kw = tap - kh * p.KW;
  CP g;                                  // the gather of the forward convolution (no dilation on this path)
  g.x = p.x; g.x2 = p.x2; g.in_scale = p.in_scale; g.in_shift = p.in_shift; g.in_relu = p.in_relu;
  g.Hsrc = p.Hsrc; g.Wsrc = p.Wsrc; g.Hin = p.Hin; g.Win = p.Win; g.C1 = p.C1; g.C2 = p.C2; g.stride = p.stride; g.pad = p.pad;
  g.up1 = p.up1; g.in_dil = 1;
  const int tx = tid & 15, ty = tid >> 4;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int pb = p0; pb < p1; pb += BK) {
    const int pix = pb + lp;
    const bool live = pix < p1;
    f32x4 dv = {0.f, 0.f, 0.f, 0.f}, xv = dv;
    if (live) {
      const int co = co0 + lq * 4;
      if (co + 3 < p.Cout) dv = *reinterpret_cast<const f32x4*>(p.dy + (size_t)pix * p.Cout + co);
      else
        for (int q = 0; q < 4; ++q)
          if (co + q < p.Cout) dv[q] = p.dy[(size_t)pix * p.Cout + co + q];
      if (klive) {
        const int n = pix / HoWo, rem = pix - n * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
        xv = gather4(g, n, ho, wo, kh, kw, ci, true);
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      sd[lp][lq * 4 + q] = dv[q];
      sx[lp][lq * 4 + q] = xv[q];
    }
    __syncthreads();
#pragma unroll
    for (int pp = 0; pp < BK; ++pp) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(&sd[pp][ty * 4]);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(&sx[pp][tx * 4]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(a4[i], b4[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int co = co0 + ty * 4 + i;
    if (co >= p.Cout) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + tx * 4 + j;
      if (k < p.Ktot) p.slab[((size_t)s * p.Cout + co) * p.Ktot + k] = acc[i][j];
    }
  }
}

__global__ void weight_prep_f32_kernel(const float* __restrict__ w, const float* __restrict__ oscale, float* __restrict__ wf,
                                       float* __restrict__ wd, int Cout, int Cin, int KH, int KW, int Cin_pad, int Cout_pad) {
  const int taps = KH * KW;
  if (wf) {      // forward layout [Cout][tap][Cin_pad]
    const int64_t total = (int64_t)Cout * taps * Cin_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      const int ci = (int)(i % Cin_pad), t = (int)((i / Cin_pad) % taps), co = (int)(i / ((int64_t)Cin_pad * taps));
      float v = 0.f;
      if (ci < Cin) {
        v = w[((size_t)co * Cin + ci) * taps + t];
        if (oscale) v *= oscale[co];
      }
      wf[i] = v;
    }
  }
  if (wd) {      // data-gradient layout [Cin_pad][flipped tap][Cout_pad]
    const int64_t total = (int64_t)Cin_pad * taps * Cout_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      const int co = (int)(i % Cout_pad), t = (int)((i / Cout_pad) % taps), ci = (int)(i / ((int64_t)Cout_pad * taps));
      float v = 0.f;
      if (ci < Cin && co < Cout) {
        v = w[((size_t)co * Cin + ci) * taps + (taps - 1 - t)];
        if (oscale) v *= oscale[co];
      }
      wd[i] = v;
    }
  }
}

}  // namespace

extern "C" int hd_conv2d_stats_rows_f32(const hd_conv_args* a) {
  if (!a) return HD_E_ARG;
  CP p;
  int rc = fill(a, p);
  if (rc) return rc;
  return hd_cdiv(p.M, BM);
}

extern "C" int hd_conv2d_f32(const hd_conv_args* a, void* stream) {
  CP p;
  int rc = fill(a, p);
  if (rc) return rc;
  dim3 grid(hd_cdiv(p.M, BM), hd_cdiv(p.Cout, BN));
  HD_CHECK_ARG(grid.y <= 65535, "hd_conv2d_f32: Cout %d too large", p.Cout);
  hipLaunchKernelGGL(conv_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_wgrad_f32(const hd_wgrad_args* a, void* stream) {
  HD_CHECK_ARG(a && a->x && a->dy && a->slab && a->nsplit >= 1, "hd_wgrad_f32: bad args");
  HD_CHECK_ARG(a->C1 > 0 && a->C1 % 4 == 0 && a->C2 >= 0 && a->C2 % 4 == 0 && (a->C2 == 0) == (a->x2 == nullptr), "hd_wgrad_f32: channel counts");
  HD_CHECK_ARG((a->in_scale == nullptr) == (a->in_shift == nullptr), "hd_wgrad_f32: in_scale / in_shift must be given together");
  WP p;
  p.x = (const float*)a->x; p.x2 = (const float*)a->x2; p.dy = (const float*)a->dy; p.slab = a->slab;
  p.in_scale = a->in_scale; p.in_shift = a->in_shift; p.in_relu = a->in_relu;
  p.N = a->N; p.Hsrc = a->Hsrc; p.Wsrc = a->Wsrc; p.Hin = a->Hin; p.Win = a->Win; p.C1 = a->C1; p.C2 = a->C2; p.Cin = a->C1 + a->C2;
  p.Ho = a->Ho; p.Wo = a->Wo; p.Cout = a->Cout; p.KH = a->KH; p.KW = a->KW; p.stride = a->stride; p.pad = a->pad; p.up1 = a->up1;
  p.nsplit = a->nsplit;
  p.M = a->N * a->Ho * a->Wo;
  p.Ktot = a->KH * a->KW * p.Cin;
  p.per = hd_cdiv(p.M, a->nsplit);
  dim3 grid(hd_cdiv(p.Cout, BM), hd_cdiv(p.Ktot, BN), a->nsplit);
  HD_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "hd_wgrad_f32: grid too large (K %d, nsplit %d)", p.Ktot, a->nsplit);
  hipLaunchKernelGGL(wgrad_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_weight_prep_f32(const float* w_oihw, const float* out_scale, void* w_fwd, void* w_dgrad, int Cout, int Cin, int KH,
                                  int KW, int Cin_pad, int Cout_pad, void* stream) {
  HD_CHECK_ARG(w_oihw && (w_fwd || w_dgrad) && Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && Cin_pad >= Cin && Cout_pad >= Cout,
               "hd_weight_prep_f32: bad args");
  const int64_t total = (int64_t)(Cout_pad > Cout ? Cout_pad : Cout) * KH * KW * Cin_pad;
  int g = (int)((total + 255) / 256);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(weight_prep_f32_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, w_oihw, out_scale, (float*)w_fwd, (float*)w_dgrad,
                     Cout, Cin, KH, KW, Cin_pad, Cout_pad);
  HD_CHECK_LAUNCH();
  return HD_OK;
}
