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
// Kernel: implicit GEMM on the MATRIX cores in fp32 (round 5): v_mfma_f32_32x32x2_f32 -- f32 operands, f32 accumulate, and by the
// ISA's definition bit for bit the k-ordered fmaf chain D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)) that the vector-ALU form of rounds 3-4
// evaluated, so the convolution outputs of this mode did not change by a bit when the inner loop moved (the reference's default
// precision is an fp32 one: src/config/config.py:149).  64 x 64 output tile per 256-thread block = four waves of 32 x 32, 16-deep K
// steps through LDS ([k][row] images: a lane's operand is one conflict-free dword), eight MFMAs per wave and step; K order
// k = tap * Cin + ci (the layout of the weights); one thread gathers four consecutive channels of one pixel at one tap (channel
// counts are multiples of 8, so a quad never straddles a tap or the concat boundary).  Peak of this path: 157 TFLOP/s (the f32 MFMA
// rate equals the f32 vector rate; what the matrix form buys is one instruction per 2 048 FMAs instead of 32, the accumulators out
// of the VALU's way, and a free vector pipe for the gather).
#include "hd_common.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 16;

// one 16-deep step of a wave's 32 x 32 sub-tile: rows from sa[k][r0 + (lane & 31)], columns from sb[k][c0 + (lane & 31)], k pairs in order
__device__ __forceinline__ void mfma_step16(const float (*sa)[BM + 4], const float (*sb)[BN + 4], int r0, int c0, int lane, f32x16& acc) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < BK / 2; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[2 * kk + h][r0 + l31], sb[2 * kk + h][c0 + l31], acc, 0, 0, 0);
}

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

// TM x TN output tile, four waves as 2 x 2, a wave owns (TM / 2) x (TN / 2) = up to 2 x 2 MFMA tiles (pick_tile_f32 below: 64 x 64 is
// the instance that is built).
template <int TM, int TN>
__global__ __launch_bounds__(256) void conv_f32_kernel(CP p) {
  constexpr int RA = TM / 64, RB = TN / 64;     // operand rows per loader thread
  constexpr int MA = TM / 64, MB = TN / 64;     // 32 x 32 MFMA tiles per wave along rows / columns
  __shared__ float sa[2][BK][TM + 4];   // [stage][k][pixel]
  __shared__ float sb[2][BK][TN + 4];   // [stage][k][cout]
  __shared__ float sred[2][TN][2];
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;
  // loader role: rows lr + 64 i (pixel / cout) and channel quad lq of the 16-deep K step
  const int lr = tid >> 2, lq = tid & 3;
  const int HoWo = p.Ho * p.Wo;
  int pn[RA], pho[RA], pwo[RA];
  bool plive[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int pm = m0 + lr + 64 * i;
    plive[i] = pm < p.M;
    pn[i] = plive[i] ? pm / HoWo : 0;
    const int prem = plive[i] ? pm - pn[i] * HoWo : 0;
    pho[i] = prem / p.Wo;
    pwo[i] = prem - pho[i] * p.Wo;
  }
  const float* wrow[RB];
  bool wlive[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int wco = n0 + lr + 64 * i;
    wlive[i] = wco < p.Cout;
    wrow[i] = p.w + (size_t)(wlive[i] ? wco : 0) * p.Ktot;
  }
  // compute role: wave (wm, wn) owns rows wm * TM/2 .. (pixels) x columns wn * TN/2 .. (couts) of the tile
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
  f32x16 acc[MA][MB];
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // this thread's K position (tap, channel quad) is stepped, not re-divided: the two integer divisions per 16-deep step were
  // ~80 of the ~150 vector instructions of a step, and with the products on the matrix cores the vector ALU is the busier pipe
  int ci = lq * 4, kh = 0, kw = 0;
  while (ci >= p.Cin) {
    ci -= p.Cin;
    if (++kw == p.KW) { kw = 0; ++kh; }
  }
  // Software pipeline: the operands of step i + 1 are requested into registers BEFORE the MFMAs of step i are issued and go to the
  // other LDS stage behind them -- one barrier per step, the global round trip of a step under the matrix work of the previous one.
  f32x4 av[RA], bv[RB];
  auto fetch = [&](int k0) {
    const int k = k0 + lq * 4;
    const bool klive = k < p.Ktot;
#pragma unroll
    for (int i = 0; i < RA; ++i) av[i] = gather4(p, pn[i], pho[i], pwo[i], kh, kw, ci, plive[i] && klive);
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      bv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (klive && wlive[i]) bv[i] = *reinterpret_cast<const f32x4*>(wrow[i] + k);
    }
    ci += BK;
    while (ci >= p.Cin) {
      ci -= p.Cin;
      if (++kw == p.KW) { kw = 0; ++kh; }
    }
  };
  auto stash = [&](int st) {
#pragma unroll
    for (int i = 0; i < RA; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) sa[st][lq * 4 + q][lr + 64 * i] = av[i][q];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) sb[st][lq * 4 + q][lr + 64 * i] = bv[i][q];
  };
  auto mfma_step = [&](int st) {
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float fa[MA], fb[MB];
#pragma unroll
      for (int a = 0; a < MA; ++a) fa[a] = sa[st][2 * kk + lh][wm * (TM / 2) + a * 32 + l31];
#pragma unroll
      for (int b = 0; b < MB; ++b) fb[b] = sb[st][2 * kk + lh][wn * (TN / 2) + b * 32 + l31];
#pragma unroll
      for (int a = 0; a < MA; ++a)
#pragma unroll
        for (int b = 0; b < MB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
  };
  fetch(0);
  stash(0);
  __syncthreads();
  int st = 0;
  for (int k0 = 0; k0 < p.Ktot; k0 += BK) {
    const bool more = k0 + BK < p.Ktot;
    if (more) fetch(k0 + BK);
    mfma_step(st);
    if (more) stash(st ^ 1);
    __syncthreads();
    st ^= 1;
  }

  // ---- epilogue: acc[a][b][r] = (pixel row wm * TM/2 + 32 a + (r & 3) + 8 (r >> 2) + 4 lh, cout column wn * TN/2 + 32 b + l31)
  float s1[MB], s2[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b) {
    const int co = n0 + wn * (TN / 2) + b * 32 + l31;
    const bool cok = co < p.Cout;
    const float bias = (p.bias && cok) ? p.bias[co] : 0.f;
    s1[b] = s2[b] = 0.f;
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (TM / 2) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= p.M || !cok) continue;
        float v = acc[a][b][r];
        if (p.res) v += p.res[(size_t)m * p.Cout + co];
        v += bias;
        if (p.mask && !(p.mask[(size_t)m * p.Cout + co] > 0.f)) v = 0.f;
        s1[b] += v;
        s2[b] += v * v;
        if (p.act == HD_ACT_RELU) v = fmaxf(v, 0.f);
        else if (p.act == HD_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
        if (p.out_mode == HD_OUT_NCHW_F32) {
          const int n = m / HoWo, rem = m - n * HoWo;
          p.y[((size_t)n * p.Cout + co) * HoWo + rem] = v;
        } else {
          p.y[(size_t)m * p.Cout + co] = v;
        }
      }
  }
  if (p.stats) {
    // per cout: the two lane halves of a wave, then the two waves that share the column range -- a fixed order
#pragma unroll
    for (int b = 0; b < MB; ++b) {
      const float t1 = s1[b] + __shfl_xor(s1[b], 32), t2 = s2[b] + __shfl_xor(s2[b], 32);
      if (lh == 0) {
        sred[wm][wn * (TN / 2) + b * 32 + l31][0] = t1;
        sred[wm][wn * (TN / 2) + b * 32 + l31][1] = t2;
      }
    }
    __syncthreads();
    if (tid < 2 * TN) {
      const int c = tid >> 1, which = tid & 1;
      const float t = sred[0][c][which] + sred[1][c][which];
      if (n0 + c < p.Cout) p.stats[((size_t)blockIdx.x * 2 + which) * p.Cout + n0 + c] = t;
    }
  }
}

// Tile of a problem.  The kernel is written for 64 / 128 rows x 64 / 128 columns; measured on the training step (bench.py --precision 32,
// same box): 64 x 64 everywhere 60 TFLOP/s over the step's convolutions, 128-wide tiles where the grid allows 53 -- the larger tiles
// take a CU from eight co-resident blocks to two or four and this kernel hides its LDS round trips with blocks, not with a pipeline.
// One instance is built.
void pick_tile_f32(int M, int Cout, int& tm, int& tn) {
  (void)M; (void)Cout;
  tm = 64;
  tn = 64;
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
  __shared__ float sd[2][BK][BM + 4];   // [stage][pixel][cout]
  __shared__ float sx[2][BK][BN + 4];   // [stage][pixel][k]
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
  // compute role: wave (wm, wn) owns couts wm * 32 .. + 31 x K columns wn * 32 .. + 31 of the tile; the reduction index is the pixel
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // the loader's pixel (n, ho, wo) is stepped by 16 per iteration instead of being divided out of the flat index every time
  int pn, pho, pwo;
  {
    const int pix0 = p0 + lp;
    pn = pix0 / HoWo;
    const int rem = pix0 - pn * HoWo;
    pho = rem / p.Wo;
    pwo = rem - pho * p.Wo;
  }
  f32x4 dv, xv;
  auto fetch = [&](int pb) {
    const int pix = pb + lp;
    const bool live = pix < p1;
    dv = f32x4{0.f, 0.f, 0.f, 0.f};
    xv = dv;
    if (live) {
      const int co = co0 + lq * 4;
      if (co + 3 < p.Cout) dv = *reinterpret_cast<const f32x4*>(p.dy + (size_t)pix * p.Cout + co);
      else
        for (int q = 0; q < 4; ++q)
          if (co + q < p.Cout) dv[q] = p.dy[(size_t)pix * p.Cout + co + q];
      if (klive) xv = gather4(g, pn, pho, pwo, kh, kw, ci, true);
    }
    pwo += BK;
    while (pwo >= p.Wo) {
      pwo -= p.Wo;
      if (++pho == p.Ho) { pho = 0; ++pn; }
    }
  };
  auto stash = [&](int st) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      sd[st][lp][lq * 4 + q] = dv[q];
      sx[st][lp][lq * 4 + q] = xv[q];
    }
  };
  if (p0 < p1) {      // (same pipeline as the convolution: next step's operands in flight under this step's MFMAs)
    fetch(p0);
    stash(0);
  }
  __syncthreads();
  int st = 0;
  for (int pb = p0; pb < p1; pb += BK) {
    const bool more = pb + BK < p1;
    if (more) fetch(pb + BK);
    mfma_step16(sd[st], sx[st], wm * 32, wn * 32, lane, acc);
    if (more) stash(st ^ 1);
    __syncthreads();
    st ^= 1;
  }
  const int k = k0 + wn * 32 + l31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (co < p.Cout && k < p.Ktot) p.slab[((size_t)s * p.Cout + co) * p.Ktot + k] = acc[r];
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
  int tm, tn;
  pick_tile_f32(p.M, p.Cout, tm, tn);
  return hd_cdiv(p.M, tm);
}

extern "C" int hd_conv2d_f32(const hd_conv_args* a, void* stream) {
  CP p;
  int rc = fill(a, p);
  if (rc) return rc;
  int tm, tn;
  pick_tile_f32(p.M, p.Cout, tm, tn);
  dim3 grid(hd_cdiv(p.M, tm), hd_cdiv(p.Cout, tn));
  HD_CHECK_ARG(grid.y <= 65535, "hd_conv2d_f32: Cout %d too large", p.Cout);
  hipLaunchKernelGGL((conv_f32_kernel<64, 64>), grid, dim3(256), 0, (hipStream_t)stream, p);
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
