// Bandwidth-bound kernels of the hot path: BatchNorm pieces, pooling, resampling,
// layout conversion.  All activations NHWC f16, 8 halves (16 B) per lane.
#include "hd_common.h"

namespace {

constexpr int TB = 256;

__device__ __forceinline__ f16x8 ld8(const f16* p) { return *reinterpret_cast<const f16x8*>(p); }
__device__ __forceinline__ void st8(f16* p, f16x8 v) { *reinterpret_cast<f16x8*>(p) = v; }

inline int grid_for(int64_t work, int per_block = TB, int cap = 256 * 16) {
  int64_t g = (work + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// ---------------------------------------------------------------- column sum
// in [rows][W] fp32 -> out[r2][W]; grid (ceil(W/64), R2), block 256 = 4 row lanes x 64 cols
__global__ void colsum_stage(const float* __restrict__ in, int rows, int W, float* __restrict__ out, int R2) {
  __shared__ double red[4][64];
  int c = blockIdx.x * 64 + (threadIdx.x & 63);
  int rl = threadIdx.x >> 6;
  int per = (rows + R2 - 1) / R2;
  int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  double s = 0.0;
  if (c < W)
    for (int r = r0 + rl; r < r1; r += 4) s += (double)in[(size_t)r * W + c];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < W) out[(size_t)blockIdx.y * W + c] = (float)(red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---------------------------------------------------------------- BN finalize
// part [rows][2][C] partial (sum, sumsq) rows of the conv epilogue are added here, ANY number of rows: block = 64 row lanes x
// 4 channels (fixed order: deterministic), so a 1 280-tile layer costs each thread 20 independent loads instead of a separate
// row-sum launch in front of this one (46 launches per training step).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int rows, int C, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* running_mean, float* running_var, float momentum, float eps,
                                                          float* mean, float* invstd, float* scale, float* shift) {
  __shared__ double red[2][4][4];
  const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int c = blockIdx.x * 4 + cl;
  double s1 = 0.0, s2 = 0.0;
  // the four threads that finish a channel ask for its parameters NOW, beside the row loads: read behind the reduction they were a second
  // dependent memory round trip (cold in step order: last touched by the previous step's optimizer) in a launch that is two round trips long
  float g = 1.f, b = 0.f, rm_old = 0.f, rv_old = 0.f;
  if (rl == 0 && c < C) {
    if (gamma) g = gamma[c];
    if (beta) b = beta[c];
    if (running_mean) {
      rm_old = running_mean[c];
      rv_old = running_var[c];
    }
  }
  if (c < C) {
    int r = rl;
    for (; r + 192 < rows; r += 256) {           // four loads of each kind in flight
      const float a0 = part[(size_t)r * 2 * C + c], a1 = part[(size_t)(r + 64) * 2 * C + c];
      const float a2 = part[(size_t)(r + 128) * 2 * C + c], a3 = part[(size_t)(r + 192) * 2 * C + c];
      const float b0 = part[(size_t)r * 2 * C + C + c], b1 = part[(size_t)(r + 64) * 2 * C + C + c];
      const float b2 = part[(size_t)(r + 128) * 2 * C + C + c], b3 = part[(size_t)(r + 192) * 2 * C + C + c];
      s1 += (double)a0 + (double)a1 + (double)a2 + (double)a3;
      s2 += (double)b0 + (double)b1 + (double)b2 + (double)b3;
    }
    for (; r < rows; r += 64) {
      s1 += (double)part[(size_t)r * 2 * C + c];
      s2 += (double)part[(size_t)r * 2 * C + C + c];
    }
  }
  // 64 row lanes -> one value per channel: inside a wave (16 row lanes x 4 channels) by four xor-shuffles, then the four waves' values through
  // LDS -- a fixed order (deterministic); the 64-step serial loop over red[][64][4] this replaces was ~0.5 us of a ~4 us dependent launch
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if ((threadIdx.x & 63) < 4) {
    red[0][threadIdx.x >> 6][cl] = s1;
    red[1][threadIdx.x >> 6][cl] = s2;
  }
  __syncthreads();
  if (rl != 0 || c >= C) return;
  s1 = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
  s2 = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
  double m = s1 / count;
  double var = s2 / count - m * m;
  if (var < 0.0) var = 0.0;
  float is = (float)(1.0 / sqrt(var + (double)eps));
  if (mean) mean[c] = (float)m;
  if (invstd) invstd[c] = is;
  scale[c] = g * is;
  shift[c] = b - (float)m * g * is;
  if (running_mean) {
    double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * rm_old + momentum * (float)m;
    running_var[c] = (1.f - momentum) * rv_old + momentum * (float)unbiased;
  }
}

__global__ void bn_eval_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C,
                               float* scale, float* shift) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float is = 1.f / sqrtf(rv[c] + eps);
  float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  scale[c] = g * is;
  shift[c] = b - rm[c] * g * is;
}

// eight consecutive floats of a per-channel table as two 16-byte loads (the element-wise form -- eight dword loads at a 32-byte lane
// stride -- costs 16 cache-line requests per wave-instruction, 32 instructions per thread in the BN backward prologue)
__device__ __forceinline__ void ld8f(const float* __restrict__ p, float (&o)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o[k] = a[k];
    o[4 + k] = b[k];
  }
}

// ---------------------------------------------------------------- BN apply (+res) (+relu)
// grid stride is a multiple of C (C = 8 * 2^k <= 2048): a thread always sees the same 8 channels -> coefficients in registers
// RES / RELU are template parameters: as run-time flags the compiler tests them with a scalar branch PER ELEMENT (8 per vector).
template <bool RES, bool RELU>
__global__ void bn_apply_kernel(const f16* __restrict__ y, const f16* __restrict__ res, const float* __restrict__ scale,
                                const float* __restrict__ shift, f16* __restrict__ z, int64_t nvec, int C) {
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c0 = (int)((i0 * 8) % C);
  float sc[8], sh[8];
  ld8f(scale + c0, sc);
  ld8f(shift + c0, sh);
  for (int64_t i = i0; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    f16x8 v = ld8(y + i * 8);
    f16x8 r;
    if (RES) r = ld8(res + i * 8);
    f16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float f = hd_bn_affine((float)v[k], sc[k], sh[k]);
      if (RES) f += (float)r[k];
      if (RELU) f = fmaxf(f, 0.f);
      o[k] = (f16)f;
    }
    st8(z + i * 8, o);
  }
}

// ---------------------------------------------------------------- BN backward
// ReLU mask: from the saved activation z when given (residual units), else recomputed from y exactly as the forward
// computed it ((f16)(y*scale+shift) > 0) -- one input stream less for every non-residual unit.
// thread tid = pl*vecs + v ; v = channel vector (8 ch), pl = pixel lane
// RELU / USEZ (mask from the saved activation z instead of recomputing it from y) are template parameters: as run-time flags
// the compiler emitted a scalar branch per element -- ~300 branches per trip, 9-10 us for a 5 MB tensor that bn_apply moves in 3.
template <bool RELU, bool USEZ>
__global__ void bn_bwd_reduce_kernel(const f16* __restrict__ dz, const f16* __restrict__ z, const f16* __restrict__ y,
                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                     float* __restrict__ part, int64_t npix, int C) {
  extern __shared__ float sm[];  // [256][16]
  const int vecs = C / 8;
  const int plan = TB / vecs;
  const int v = threadIdx.x % vecs, pl = threadIdx.x / vecs;
  const int64_t per = (npix + gridDim.x - 1) / gridDim.x;
  const int64_t p0 = (int64_t)blockIdx.x * per, p1 = min(npix, p0 + per);
  float sg[8], sgx[8], mu[8], is[8], sc[8], sh[8], ga[8], be[8];
  ld8f(mean + v * 8, mu);
  ld8f(invstd + v * 8, is);
  if (gamma) ld8f(gamma + v * 8, ga);
  if (beta) ld8f(beta + v * 8, be);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sg[k] = sgx[k] = 0.f;
    const float g = gamma ? ga[k] : 1.f, b = beta ? be[k] : 0.f;
    sc[k] = g * is[k];
    sh[k] = b - mu[k] * g * is[k];
  }
  if (pl < plan) {
    // 4 pixels per trip with all loads issued up front: the loop is otherwise one dependent HBM round trip per pixel
    constexpr int U = 4;
    for (int64_t p = p0 + pl; p < p1; p += (int64_t)plan * U) {
      f16x8 g[U], yy[U], zz[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t pu = p + (int64_t)u * plan;
        const size_t off = (size_t)(pu < p1 ? pu : p) * C + v * 8;
        g[u] = ld8(dz + off);
        yy[u] = ld8(y + off);
        if (USEZ) zz[u] = ld8(z + off);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float live = (p + (int64_t)u * plan < p1) ? 1.f : 0.f;      // a select, not a branch: the trip stays straight-line code
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float gk = (float)g[u][k] * live;
          if (RELU) {
            const bool on = USEZ ? ((float)zz[u][k] > 0.f) : ((float)(f16)hd_bn_affine((float)yy[u][k], sc[k], sh[k]) > 0.f);
            gk = on ? gk : 0.f;
          }
          const float xh = ((float)yy[u][k] - mu[k]) * is[k];
          sg[k] += gk;
          sgx[k] += gk * xh;
        }
      }
    }
  }
  // [16 values][256 threads]: consecutive lanes on consecutive banks (the [thread][16] form was a 16-way conflict on every write)
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sm[k * TB + threadIdx.x] = sg[k];
    sm[(8 + k) * TB + threadIdx.x] = sgx[k];
  }
  __syncthreads();
  for (int o = threadIdx.x; o < 2 * C; o += TB) {
    const int which = o >= C ? 1 : 0, c = o - which * C;
    const int vv = c >> 3, k = c & 7;
    float s = 0.f;
    for (int q = 0; q < plan; ++q) s += sm[(which * 8 + k) * TB + q * vecs + vv];
    part[(size_t)blockIdx.x * 2 * C + o] = s;
  }
}

// column c of the partial rows; the common 16-row case is unrolled so that all 32 loads are in flight at once
__device__ __forceinline__ void sum_part_rows(const float* __restrict__ part, int rows, int C, int c, float& sg, float& sgx) {
  if (rows == 16) {
    float a[16], b[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      a[r] = part[(size_t)r * 2 * C + c];
      b[r] = part[(size_t)r * 2 * C + C + c];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sg += a[r];
      sgx += b[r];
    }
    return;
  }
  // any other row count (<= 64): eight rows per trip, all sixteen loads issued before the first add (a plain loop is one
  // dependent L2 round trip per row in EVERY block's prologue: measured +0.4 ms per training step)
  int r = 0;
  for (; r + 8 <= rows; r += 8) {
    float a[8], b[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      a[q] = part[(size_t)(r + q) * 2 * C + c];
      b[q] = part[(size_t)(r + q) * 2 * C + C + c];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      sg += a[q];
      sgx += b[q];
    }
  }
  for (; r < rows; ++r) {
    sg += part[(size_t)r * 2 * C + c];
    sgx += part[(size_t)r * 2 * C + C + c];
  }
}

// Per-channel coefficients of the BN backward, ONCE per unit: sums any number of partial rows (64 row lanes x 4 channels per
// block, fixed order), writes coef [5][C] = A, B, D (below), forward scale, forward shift, and dgamma / dbeta.  Before, every
// block of the apply kernel summed the partial rows itself: up to 64 x 2C floats per block, ten times its payload on the small
// layers (bn_bwd_apply 0.68 -> 1.11 ms per step when the separate row-sum launch was dropped naively).
__global__ __launch_bounds__(256) void bn_bwd_coef_kernel(const float* __restrict__ part, int rows, int C, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float invM, float gscale, int accumulate,
                                                          float* dgamma, float* dbeta, float* __restrict__ coef) {
  __shared__ float red[2][4][4];
  const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int c = blockIdx.x * 4 + cl;
  float s1 = 0.f, s2 = 0.f;
  // (as in bn_finalize_kernel: the finishing threads' per-channel loads are issued beside the row loads)
  float ga = 1.f, be = 0.f, is = 0.f, mu = 0.f, dg_old = 0.f, db_old = 0.f;
  if (rl == 0 && c < C) {
    if (gamma) ga = gamma[c];
    if (beta) be = beta[c];
    is = invstd[c];
    mu = mean[c];
    if (accumulate) {
      if (dgamma) dg_old = dgamma[c];
      if (dbeta) db_old = dbeta[c];
    }
  }
  if (c < C) {
    int r = rl;
    for (; r + 192 < rows; r += 256) {
      const float a0 = part[(size_t)r * 2 * C + c], a1 = part[(size_t)(r + 64) * 2 * C + c];
      const float a2 = part[(size_t)(r + 128) * 2 * C + c], a3 = part[(size_t)(r + 192) * 2 * C + c];
      const float b0 = part[(size_t)r * 2 * C + C + c], b1 = part[(size_t)(r + 64) * 2 * C + C + c];
      const float b2 = part[(size_t)(r + 128) * 2 * C + C + c], b3 = part[(size_t)(r + 192) * 2 * C + C + c];
      s1 += (a0 + a1) + (a2 + a3);
      s2 += (b0 + b1) + (b2 + b3);
    }
    for (; r < rows; r += 64) {
      s1 += part[(size_t)r * 2 * C + c];
      s2 += part[(size_t)r * 2 * C + C + c];
    }
  }
#pragma unroll
  for (int o = 4; o < 64; o <<= 1) {          // (as in bn_finalize_kernel: wave-level xor-shuffles, then the four waves' values)
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if ((threadIdx.x & 63) < 4) {
    red[0][threadIdx.x >> 6][cl] = s1;
    red[1][threadIdx.x >> 6][cl] = s2;
  }
  __syncthreads();
  if (rl != 0 || c >= C) return;
  const float sg = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
  const float sgx = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
  const float dg = sgx * gscale, db = sg * gscale;
  if (dgamma) dgamma[c] = accumulate ? dg_old + dg : dg;
  if (dbeta) dbeta[c] = accumulate ? db_old + db : db;
  const float a_ = ga * is;
  const float b_ = -a_ * is * sgx * invM;
  coef[c] = a_;
  coef[C + c] = b_;
  coef[2 * C + c] = -a_ * sg * invM - b_ * mu;
  coef[3 * C + c] = ga * is;
  coef[4 * C + c] = be - mu * ga * is;
}

// dy = A[c]*g + B[c]*y + D[c]  with  A = gamma*invstd, B = -A*invstd*sum_gx/M, D = -A*sum_g/M - B*mean
template <bool RELU, bool USEZ, bool DRES>
__global__ void bn_bwd_apply_kernel(const f16* __restrict__ dz, const f16* __restrict__ z, const f16* __restrict__ y,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ coef, f16* __restrict__ dy,
                                    f16* __restrict__ dres, int64_t npix, int C) {
  const int64_t nvec = npix * C / 8;
  const float* __restrict__ cf = coef;        // [5][C] from bn_bwd_coef_kernel: A, B, D, sc, sh
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c0 = (int)((i0 * 8) % C);
  float A[8], B[8], D[8], sc[8], sh[8];
  ld8f(cf + c0, A);
  ld8f(cf + C + c0, B);
  ld8f(cf + 2 * C + c0, D);
  ld8f(cf + 3 * C + c0, sc);
  ld8f(cf + 4 * C + c0, sh);
  for (int64_t i = i0; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    f16x8 g = ld8(dz + i * 8), yy = ld8(y + i * 8), zz;
    if (USEZ) zz = ld8(z + i * 8);
    f16x8 o, gr;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float gk = (float)g[k];
      const float yk = (float)yy[k];
      if (RELU) {
        const bool on = USEZ ? ((float)zz[k] > 0.f) : ((float)(f16)hd_bn_affine(yk, sc[k], sh[k]) > 0.f);
        gk = on ? gk : 0.f;
      }
      o[k] = (f16)(A[k] * gk + B[k] * yk + D[k]);
      gr[k] = (f16)gk;
    }
    st8(dy + i * 8, o);
    if (DRES) st8(dres + i * 8, gr);
  }
}

// ---------------------------------------------------------------- maxpool 3x3 s2 p1
__global__ void maxpool_kernel(const f16* __restrict__ x, f16* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo) {
  const int vecs = C / 8;
  const int64_t total = (int64_t)N * Ho * Wo * vecs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int v = (int)(i % vecs);
    int64_t p = i / vecs;
    int wo = (int)(p % Wo);
    int ho = (int)((p / Wo) % Ho);
    int n = (int)(p / ((int64_t)Wo * Ho));
    float m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = -INFINITY;
    for (int kh = 0; kh < 3; ++kh) {
      int h = ho * 2 - 1 + kh;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        int w = wo * 2 - 1 + kw;
        if ((unsigned)w >= (unsigned)W) continue;
        f16x8 t = ld8(x + ((size_t)(n * H + h) * W + w) * C + v * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], (float)t[k]);
      }
    }
    f16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (f16)m[k];
    st8(y + (size_t)p * C + v * 8, o);
  }
}

// gather-form backward: first maximal element in (kh,kw) scan order receives the gradient (ATen rule: strict >)
__global__ void maxpool_bwd_kernel(const f16* __restrict__ x, const f16* __restrict__ dy, f16* __restrict__ dx, int N, int H,
                                   int W, int C, int Ho, int Wo) {
  const int vecs = C / 8;
  const int64_t total = (int64_t)N * H * W * vecs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int v = (int)(i % vecs);
    int64_t p = i / vecs;
    int w = (int)(p % W);
    int h = (int)((p / W) % H);
    int n = (int)(p / ((int64_t)W * H));
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    int ho0 = max(0, h / 2), ho1 = min(Ho - 1, (h + 1) / 2);
    int wo0 = max(0, w / 2), wo1 = min(Wo - 1, (w + 1) / 2);
    for (int ho = ho0; ho <= ho1; ++ho)
      for (int wo = wo0; wo <= wo1; ++wo) {
        // my position inside this window
        int mykh = h - (ho * 2 - 1), mykw = w - (wo * 2 - 1);
        if (mykh < 0 || mykh > 2 || mykw < 0 || mykw > 2) continue;
        int mypos = mykh * 3 + mykw;
        float best[8];
        int bidx[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          best[k] = -INFINITY;
          bidx[k] = -1;
        }
        for (int kh = 0; kh < 3; ++kh) {
          int hh = ho * 2 - 1 + kh;
          if ((unsigned)hh >= (unsigned)H) continue;
          for (int kw = 0; kw < 3; ++kw) {
            int ww = wo * 2 - 1 + kw;
            if ((unsigned)ww >= (unsigned)W) continue;
            f16x8 t = ld8(x + ((size_t)(n * H + hh) * W + ww) * C + v * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              float f = (float)t[k];
              if (f > best[k] || bidx[k] < 0) {
                best[k] = f;
                bidx[k] = kh * 3 + kw;
              }
            }
          }
        }
        f16x8 g = ld8(dy + ((size_t)(n * Ho + ho) * Wo + wo) * C + v * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (bidx[k] == mypos) acc[k] += (float)g[k];
      }
    f16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (f16)acc[k];
    st8(dx + (size_t)p * C + v * 8, o);
  }
}

// forward that also records, per output element, WHICH of the 9 window positions won (first maximum in scan order): the
// backward then reads 1 byte + the gradient per window instead of re-deriving the argmax from 9 inputs per window
__global__ void maxpool_idx_kernel(const f16* __restrict__ x, f16* __restrict__ y, unsigned char* __restrict__ idx, int N, int H, int W,
                                   int C, int Ho, int Wo) {
  const int vecs = C / 8;
  const int64_t total = (int64_t)N * Ho * Wo * vecs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int v = (int)(i % vecs);
    int64_t p = i / vecs;
    int wo = (int)(p % Wo);
    int ho = (int)((p / Wo) % Ho);
    int n = (int)(p / ((int64_t)Wo * Ho));
    float best[8];
    int bidx[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      best[k] = -INFINITY;
      bidx[k] = -1;
    }
    for (int kh = 0; kh < 3; ++kh) {
      int h = ho * 2 - 1 + kh;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        int w = wo * 2 - 1 + kw;
        if ((unsigned)w >= (unsigned)W) continue;
        f16x8 t = ld8(x + ((size_t)(n * H + h) * W + w) * C + v * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float f = (float)t[k];
          if (f > best[k] || bidx[k] < 0) {
            best[k] = f;
            bidx[k] = kh * 3 + kw;
          }
        }
      }
    }
    f16x8 o;
    unsigned long long pk = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      o[k] = (f16)best[k];
      pk |= (unsigned long long)(bidx[k] & 0xff) << (8 * k);
    }
    st8(y + (size_t)p * C + v * 8, o);
    *reinterpret_cast<unsigned long long*>(idx + (size_t)p * C + v * 8) = pk;
  }
}

// ADD: dx = f16(f16(routed gradient) + add) -- the sum of the routed gradient with a second gradient of the pooled tensor's INPUT
// (the U-Net's f1 also feeds a decoder skip), rounded exactly as the separate hd_add_f16 pass it replaces
template <bool ADD>
__global__ void maxpool_bwd_idx_kernel(const unsigned char* __restrict__ idx, const f16* __restrict__ dy, const f16* __restrict__ add,
                                       f16* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo) {
  const int vecs = C / 8;
  const int64_t total = (int64_t)N * H * W * vecs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int v = (int)(i % vecs);
    int64_t p = i / vecs;
    int w = (int)(p % W);
    int h = (int)((p / W) % H);
    int n = (int)(p / ((int64_t)W * H));
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    int ho0 = max(0, h / 2), ho1 = min(Ho - 1, (h + 1) / 2);
    int wo0 = max(0, w / 2), wo1 = min(Wo - 1, (w + 1) / 2);
    for (int ho = ho0; ho <= ho1; ++ho)
      for (int wo = wo0; wo <= wo1; ++wo) {
        int mykh = h - (ho * 2 - 1), mykw = w - (wo * 2 - 1);
        if (mykh < 0 || mykh > 2 || mykw < 0 || mykw > 2) continue;
        const unsigned long long mypos = (unsigned long long)(mykh * 3 + mykw);
        const size_t off = ((size_t)(n * Ho + ho) * Wo + wo) * C + v * 8;
        const unsigned long long pk = *reinterpret_cast<const unsigned long long*>(idx + off);
        f16x8 g = ld8(dy + off);
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (((pk >> (8 * k)) & 0xffull) == mypos) acc[k] += (float)g[k];
      }
    f16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (f16)acc[k];
    if (ADD) {
      const f16x8 e = ld8(add + (size_t)p * C + v * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (f16)((float)o[k] + (float)e[k]);
    }
    st8(dx + (size_t)p * C + v * 8, o);
  }
}

__global__ void subsample2_kernel(const f16* __restrict__ x, f16* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo) {
  const int vecs = C / 8;
  const int64_t total = (int64_t)N * Ho * Wo * vecs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int v = (int)(i % vecs);
    int64_t p = i / vecs;
    int wo = (int)(p % Wo);
    int ho = (int)((p / Wo) % Ho);
    int n = (int)(p / ((int64_t)Wo * Ho));
    st8(y + (size_t)p * C + v * 8, ld8(x + ((size_t)(n * H + ho * 2) * W + wo * 2) * C + v * 8));
  }
}

__global__ void subsample2_bwd_kernel(const f16* __restrict__ dy, f16* __restrict__ dx, int N, int H, int W, int C, int Ho,
                                      int Wo, int accumulate) {
  const int vecs = C / 8;
  const int64_t total = (int64_t)N * H * W * vecs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int v = (int)(i % vecs);
    int64_t p = i / vecs;
    int w = (int)(p % W);
    int h = (int)((p / W) % H);
    int n = (int)(p / ((int64_t)W * H));
    f16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (f16)0.f;
    if ((h & 1) == 0 && (w & 1) == 0 && h / 2 < Ho && w / 2 < Wo)
      o = ld8(dy + ((size_t)(n * Ho + h / 2) * Wo + w / 2) * C + v * 8);
    f16* d = dx + (size_t)p * C + v * 8;
    if (accumulate) {
      f16x8 e = ld8(d);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (f16)((float)o[k] + (float)e[k]);
    }
    st8(d, o);
  }
}

// ---------------------------------------------------------------- nearest index (ATen legacy 'nearest')
__device__ __forceinline__ int nn_src(int dst, float scale, int in_size) {
  int s = (int)floorf((float)dst * scale);
  return s < in_size - 1 ? s : in_size - 1;
}

// `cstride` / `nstride` (elements): channel / image stride of x.  cstride == 0 reads every channel from plane 0: the reference's
// `imgs_ir.repeat(1, 3, 1, 1)` (src/utils/utils.py:52-53) as a stride-0 view -- the three copies are never materialised.
__global__ void nchw_to_nhwc_resize_kernel(const float* __restrict__ x, f16* __restrict__ y, int N, int Cr, int H, int W,
                                           int Ho, int Wo, int Cp, float sh, float sw, int64_t nstride, int64_t cstride) {
  const int64_t total = (int64_t)N * Ho * Wo;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    int wo = (int)(p % Wo);
    int ho = (int)((p / Wo) % Ho);
    int n = (int)(p / ((int64_t)Wo * Ho));
    int hs = nn_src(ho, sh, H), ws = nn_src(wo, sw, W);
    for (int c0 = 0; c0 < Cp; c0 += 8) {
      f16x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        int c = c0 + k;
        o[k] = c < Cr ? (f16)x[(size_t)n * nstride + (size_t)c * cstride + (size_t)hs * W + ws] : (f16)0.f;
      }
      st8(y + (size_t)p * Cp + c0, o);
    }
  }
}

__global__ void nchw_to_nhwc_resize_bwd_kernel(const f16* __restrict__ dy, float* __restrict__ dx, int N, int Cr, int H, int W,
                                               int Ho, int Wo, int Cp, float sh, float sw, float gscale) {
  const int64_t total = (int64_t)N * H * W;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    int w = (int)(p % W);
    int h = (int)((p / W) % H);
    int n = (int)(p / ((int64_t)W * H));
    float acc[8];
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    int ho_lo = max(0, (int)floorf((float)h / sh) - 1), ho_hi = min(Ho - 1, (int)floorf((float)(h + 1) / sh) + 1);
    int wo_lo = max(0, (int)floorf((float)w / sw) - 1), wo_hi = min(Wo - 1, (int)floorf((float)(w + 1) / sw) + 1);
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
      if (nn_src(ho, sh, H) != h) continue;
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        if (nn_src(wo, sw, W) != w) continue;
        const f16* g = dy + ((size_t)(n * Ho + ho) * Wo + wo) * Cp;
        for (int c = 0; c < Cr && c < 8; ++c) acc[c] += (float)g[c];
      }
    }
    for (int c = 0; c < Cr && c < 8; ++c) dx[((size_t)(n * Cr + c) * H + h) * W + w] = acc[c] * gscale;
  }
}

__global__ void nhwc_to_nchw_kernel(const f16* __restrict__ x, float* __restrict__ y, int N, int Cr, int H, int W, int Cp) {
  const int64_t total = (int64_t)N * H * W;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    int64_t hw = p % ((int64_t)H * W);
    int n = (int)(p / ((int64_t)H * W));
    for (int c = 0; c < Cr; ++c) y[((size_t)n * Cr + c) * H * W + hw] = (float)x[(size_t)p * Cp + c];
  }
}

// y = a + nearest(b)
__global__ void upsample_add_kernel(const f16* __restrict__ a, const f16* __restrict__ b, f16* __restrict__ y, int N, int H,
                                    int W, int C, int Hb, int Wb, float sh, float sw) {
  const int vecs = C / 8;
  const int64_t total = (int64_t)N * H * W * vecs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int v = (int)(i % vecs);
    int64_t p = i / vecs;
    int w = (int)(p % W);
    int h = (int)((p / W) % H);
    int n = (int)(p / ((int64_t)W * H));
    int hs = nn_src(h, sh, Hb), ws = nn_src(w, sw, Wb);
    f16x8 va = ld8(a + (size_t)p * C + v * 8);
    f16x8 vb = ld8(b + ((size_t)(n * Hb + hs) * Wb + ws) * C + v * 8);
    f16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (f16)((float)va[k] + (float)vb[k]);
    st8(y + (size_t)p * C + v * 8, o);
  }
}

__global__ void upsample_add_bwd_kernel(const f16* __restrict__ dy, f16* __restrict__ db, int N, int H, int W, int C, int Hb,
                                        int Wb, float sh, float sw, int accumulate) {
  const int vecs = C / 8;
  const int64_t total = (int64_t)N * Hb * Wb * vecs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int v = (int)(i % vecs);
    int64_t p = i / vecs;
    int wb = (int)(p % Wb);
    int hb = (int)((p / Wb) % Hb);
    int n = (int)(p / ((int64_t)Wb * Hb));
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    int h_lo = max(0, (int)floorf((float)hb / sh) - 1), h_hi = min(H - 1, (int)floorf((float)(hb + 1) / sh) + 1);
    int w_lo = max(0, (int)floorf((float)wb / sw) - 1), w_hi = min(W - 1, (int)floorf((float)(wb + 1) / sw) + 1);
    for (int h = h_lo; h <= h_hi; ++h) {
      if (nn_src(h, sh, Hb) != hb) continue;
      for (int w = w_lo; w <= w_hi; ++w) {
        if (nn_src(w, sw, Wb) != wb) continue;
        f16x8 g = ld8(dy + ((size_t)(n * H + h) * W + w) * C + v * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += (float)g[k];
      }
    }
    f16* d = db + (size_t)p * C + v * 8;
    f16x8 o;
    if (accumulate) {
      f16x8 e = ld8(d);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += (float)e[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (f16)acc[k];
    st8(d, o);
  }
}

__global__ void upsample2_bwd_kernel(const f16* __restrict__ dy, f16* __restrict__ dx, int N, int Hl, int Wl, int C, int Ctot,
                                     int c_off, int accumulate) {
  const int vecs = C / 8;
  const int64_t total = (int64_t)N * Hl * Wl * vecs;
  const int Hu = Hl * 2, Wu = Wl * 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int v = (int)(i % vecs);
    int64_t p = i / vecs;
    int w = (int)(p % Wl);
    int h = (int)((p / Wl) % Hl);
    int n = (int)(p / ((int64_t)Wl * Hl));
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        f16x8 g = ld8(dy + ((size_t)(n * Hu + 2 * h + a) * Wu + 2 * w + b) * Ctot + c_off + v * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += (float)g[k];
      }
    f16* d = dx + (size_t)p * C + v * 8;
    if (accumulate) {
      f16x8 e = ld8(d);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += (float)e[k];
    }
    f16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (f16)acc[k];
    st8(d, o);
  }
}

__global__ void add_kernel(const f16* __restrict__ a, const f16* __restrict__ b, f16* __restrict__ o, int64_t nvec) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    f16x8 x = ld8(a + i * 8), y = ld8(b + i * 8), r;
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = (f16)((float)x[k] + (float)y[k]);
    st8(o + i * 8, r);
  }
}

__global__ void slice_channels_kernel(const f16* __restrict__ x, f16* __restrict__ y, int64_t npix, int Ctot, int c_off, int C,
                                      int accumulate) {
  const int vecs = C / 8;
  const int64_t total = npix * vecs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int v = (int)(i % vecs);
    int64_t p = i / vecs;
    f16x8 g = ld8(x + (size_t)p * Ctot + c_off + v * 8);
    f16* d = y + (size_t)p * C + v * 8;
    if (accumulate) {
      f16x8 e = ld8(d);
#pragma unroll
      for (int k = 0; k < 8; ++k) g[k] = (f16)((float)g[k] + (float)e[k]);
    }
    st8(d, g);
  }
}

// Gradient of cat([nearest_2x(a), skip], channel) in ONE launch: the first `n_low` vectors are the 2x2 sum-pool of channels [0, Cup)
// into the low-resolution tensor (hd_upsample2_bwd's arithmetic and order), the rest copy channels [Cup, Cup + Cskip) into the
// skip gradient (hd_slice_channels).  Reference: the autograd of decoder.py:37-41 (interpolate + torch.cat).
__global__ void concat_up_bwd_kernel(const f16* __restrict__ dcat, f16* __restrict__ dlow, f16* __restrict__ dskip, int N, int Hl, int Wl,
                                     int Cup, int Cskip) {
  const int Ctot = Cup + Cskip, Hu = Hl * 2, Wu = Wl * 2;
  const int vu = Cup / 8, vs = Cskip / 8;
  const int64_t n_low = (int64_t)N * Hl * Wl * vu, total = n_low + (int64_t)N * Hu * Wu * vs;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < n_low) {
      const int v = (int)(i % vu);
      const int64_t p = i / vu;
      const int w = (int)(p % Wl), h = (int)((p / Wl) % Hl), n = (int)(p / ((int64_t)Wl * Hl));
      float acc[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const f16x8 g = ld8(dcat + ((size_t)(n * Hu + 2 * h + a) * Wu + 2 * w + b) * Ctot + v * 8);
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[k] += (float)g[k];
        }
      f16x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (f16)acc[k];
      st8(dlow + (size_t)p * Cup + v * 8, o);
    } else {
      const int64_t j = i - n_low;
      const int v = (int)(j % vs);
      const int64_t p = j / vs;
      st8(dskip + (size_t)p * Cskip + v * 8, ld8(dcat + (size_t)p * Ctot + Cup + v * 8));
    }
  }
}

__global__ void sigmoid_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ s, f16* __restrict__ dl, int N,
                                   int Cr, int H, int W, int Cp, float gscale) {
  const int64_t total = (int64_t)N * H * W;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    int64_t hw = p % ((int64_t)H * W);
    int n = (int)(p / ((int64_t)H * W));
    for (int c0 = 0; c0 < Cp; c0 += 8) {
      f16x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        int c = c0 + k;
        float v = 0.f;
        if (c < Cr) {
          size_t idx = ((size_t)n * Cr + c) * H * W + hw;
          float sv = s[idx];
          v = dy[idx] * sv * (1.f - sv) * gscale;
        }
        o[k] = (f16)v;
      }
      st8(dl + (size_t)p * Cp + c0, o);
    }
  }
}

__global__ void relu_bwd_kernel(const f16* __restrict__ dy, const f16* __restrict__ z, f16* __restrict__ dx, int64_t nvec) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    f16x8 g = ld8(dy + i * 8), zz = ld8(z + i * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (!((float)zz[k] > 0.f)) g[k] = (f16)0.f;
    st8(dx + i * 8, g);
  }
}

__global__ void f32_to_f16_kernel(const float* __restrict__ x, f16* __restrict__ y, int64_t n, float scale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = (f16)(x[i] * scale);
}
__global__ void f16_to_f32_kernel(const f16* __restrict__ x, float* __restrict__ y, int64_t n, float scale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = (float)x[i] * scale;
}

// per-channel sum of an NHWC f16 tensor -> part[rows][C]  (bias gradients)
// [P][C] fp32 -> [P][Cp] fp16, channels >= C zero (gradients of the small fp32 head outputs on their way into a data-gradient conv).
// The source may be a slice of a larger buffer: row p lives at (p / rows_per_image) * image_stride + (p % rows_per_image) * C.
__global__ void pad_cast_kernel(const float* __restrict__ x, f16* __restrict__ y, int64_t P, int C, int Cp, int64_t rows_per_image,
                                int64_t image_stride) {
  const int64_t total = P * Cp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / Cp;
    const int c = (int)(i - p * Cp);
    const int64_t n = p / rows_per_image;
    y[i] = c < C ? (f16)x[n * image_stride + (p - n * rows_per_image) * C + c] : (f16)0.f;
  }
}

// several pad_cast problems in ONE launch (the ten head gradients of an RPN backward pass: five levels x {objectness, box deltas},
// each a 2-30 KB tensor that paid a launch of its own): the descriptor table travels in the kernel arguments, block ranges -> entries
constexpr int PAD_CAST_MAX = 16;
struct PadCastTab {
  const float* x[PAD_CAST_MAX];
  f16* y[PAD_CAST_MAX];
  int64_t P[PAD_CAST_MAX], rows_per_image[PAD_CAST_MAX], image_stride[PAD_CAST_MAX];
  int C[PAD_CAST_MAX], Cp[PAD_CAST_MAX], first[PAD_CAST_MAX + 1];
  int n;
};
__global__ void pad_cast_multi_kernel(PadCastTab t) {
  int e = 0;
  for (int i = 1; i < t.n; ++i)
    if ((int)blockIdx.x >= t.first[i]) e = i;
  const float* __restrict__ x = t.x[e];
  f16* __restrict__ y = t.y[e];
  const int C = t.C[e], Cp = t.Cp[e];
  const int64_t total = t.P[e] * Cp, rpi = t.rows_per_image[e], istr = t.image_stride[e];
  const int nb = t.first[e + 1] - t.first[e];
  for (int64_t i = (int64_t)((int)blockIdx.x - t.first[e]) * blockDim.x + threadIdx.x; i < total; i += (int64_t)nb * blockDim.x) {
    const int64_t p = i / Cp;
    const int c = (int)(i - p * Cp);
    const int64_t n = p / rpi;
    y[i] = c < C ? (f16)x[n * istr + (p - n * rpi) * C + c] : (f16)0.f;
  }
}

__global__ void channel_sum_kernel(const f16* __restrict__ x, int64_t npix, int C, float* __restrict__ part) {
  extern __shared__ float sm[];  // [256][8]
  const int vecs = C / 8;
  const int plan = TB / vecs;
  const int v = threadIdx.x % vecs, pl = threadIdx.x / vecs;
  const int64_t per = (npix + gridDim.x - 1) / gridDim.x;
  const int64_t p0 = (int64_t)blockIdx.x * per, p1 = min(npix, p0 + per);
  float s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = 0.f;
  if (pl < plan)
    for (int64_t p = p0 + pl; p < p1; p += plan) {
      f16x8 g = ld8(x + (size_t)p * C + v * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += (float)g[k];
    }
#pragma unroll
  for (int k = 0; k < 8; ++k) sm[threadIdx.x * 8 + k] = s[k];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += TB) {
    int vv = c / 8, k = c & 7;
    float t = 0.f;
    for (int q = 0; q < plan; ++q) t += sm[(q * vecs + vv) * 8 + k];
    part[(size_t)blockIdx.x * C + c] = t;
  }
}

__global__ void scale_store_kernel(const float* __restrict__ in, float* __restrict__ out, int n, float scale, int accumulate) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = accumulate ? out[i] + in[i] * scale : in[i] * scale;
}

bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

#define S_ ((hipStream_t)stream)

#ifndef HD_STORE_F32
extern "C" int hd_colsum(const float* in, int rows, int W, float* out, float* ws, void* stream) {
  HD_CHECK_ARG(in && out && rows > 0 && W > 0, "hd_colsum: bad args");
  int R2 = rows > 512 ? 128 : (rows > 32 ? 16 : 1);
  HD_CHECK_ARG(R2 == 1 || ws, "hd_colsum: workspace required for rows > 32");
  dim3 g(hd_cdiv(W, 64), R2);
  if (R2 == 1) {
    hipLaunchKernelGGL(colsum_stage, g, dim3(256), 0, S_, in, rows, W, out, 1);
  } else {
    hipLaunchKernelGGL(colsum_stage, g, dim3(256), 0, S_, in, rows, W, ws, R2);
    hipLaunchKernelGGL(colsum_stage, dim3(hd_cdiv(W, 64), 1), dim3(256), 0, S_, (const float*)ws, R2, W, out, 1);
  }
  HD_CHECK_LAUNCH();
  return HD_OK;
}
#endif

#ifndef HD_STORE_F32
extern "C" int hd_rowsum(const float* in, int rows, int W, float* out, int out_rows, void* stream) {
  HD_CHECK_ARG(in && out && rows > 0 && W > 0 && out_rows > 0 && out_rows <= rows, "hd_rowsum: bad args");
  hipLaunchKernelGGL(colsum_stage, dim3(hd_cdiv(W, 64), out_rows), dim3(256), 0, S_, in, rows, W, out, out_rows);
  HD_CHECK_LAUNCH();
  return HD_OK;
}
#endif

#ifndef HD_STORE_F32
extern "C" int hd_bn_finalize(const float* part, int rows, int C, double count, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                              float* scale, float* shift, void* stream) {
  HD_CHECK_ARG(part && rows > 0 && scale && shift && C > 0 && count > 0, "hd_bn_finalize: bad args");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(hd_cdiv(C, 4)), dim3(256), 0, S_, part, rows, C, count, gamma, beta, running_mean,
                     running_var, momentum, eps, mean, invstd, scale, shift);
  HD_CHECK_LAUNCH();
  return HD_OK;
}
#endif

#ifndef HD_STORE_F32
extern "C" int hd_bn_eval_scale_shift(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                      float eps, int C, float* scale, float* shift, void* stream) {
  HD_CHECK_ARG(running_mean && running_var && scale && shift && C > 0, "hd_bn_eval_scale_shift: bad args");
  hipLaunchKernelGGL(bn_eval_kernel, dim3(hd_cdiv(C, 64)), dim3(64), 0, S_, gamma, beta, running_mean, running_var, eps, C, scale, shift);
  HD_CHECK_LAUNCH();
  return HD_OK;
}
#endif

extern "C" int HD_API(hd_bn_apply)(const void* y, const void* res, const float* scale, const float* shift, void* z, int64_t n, int C,
                           int relu, void* stream) {
  HD_CHECK_ARG(y && z && scale && shift && n > 0 && C % 8 == 0 && n % 8 == 0, "hd_bn_apply: bad args");
  HD_CHECK_ARG(pow2(C / 8) && C <= 2048, "hd_bn_apply: C/8 must be a power of two (C=%d)", C);
#define HD_BNA(RS, RL) hipLaunchKernelGGL((bn_apply_kernel<RS, RL>), dim3(grid_for(n / 8)), dim3(TB), 0, S_, (const f16*)y, (const f16*)res, scale, shift, (f16*)z, n / 8, C)
  if (res) { if (relu) HD_BNA(true, true); else HD_BNA(true, false); }
  else { if (relu) HD_BNA(false, true); else HD_BNA(false, false); }
#undef HD_BNA
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_bn_bwd_reduce)(const void* dz, const void* z, const void* y, const float* mean, const float* invstd,
                                const float* gamma, const float* beta, float* part, int rows, int64_t npix, int C, int relu,
                                void* stream) {
  HD_CHECK_ARG(dz && y && mean && invstd && part && rows > 0 && npix > 0, "hd_bn_bwd_reduce: bad args");
  HD_CHECK_ARG(C % 8 == 0 && pow2(C / 8) && C / 8 <= TB, "hd_bn_bwd_reduce: C/8 must be a power of two <= 256 (C=%d)", C);
  const bool usez = relu && z != nullptr;
#define HD_RED(R, Z) hipLaunchKernelGGL((bn_bwd_reduce_kernel<R, Z>), dim3(rows), dim3(TB), TB * 16 * sizeof(float), S_, (const f16*)dz, \
                                        (const f16*)z, (const f16*)y, mean, invstd, gamma, beta, part, npix, C)
  if (!relu) HD_RED(false, false);
  else if (usez) HD_RED(true, true);
  else HD_RED(true, false);
#undef HD_RED
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_bn_bwd_apply)(const void* dz, const void* z, const void* y, const float* mean, const float* invstd,
                               const float* gamma, const float* beta, const float* part, int rows, float* coef_ws, void* dy, void* dres,
                               float* dgamma, float* dbeta, float gscale, int accumulate, int64_t npix, int C, int relu,
                               void* stream) {
  HD_CHECK_ARG(dz && y && mean && invstd && part && rows > 0 && coef_ws && dy && npix > 0, "hd_bn_bwd_apply: bad args");
  HD_CHECK_ARG(C % 8 == 0 && pow2(C / 8) && C <= 2048, "hd_bn_bwd_apply: C/8 must be a power of two (C=%d)", C);
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3(hd_cdiv(C, 4)), dim3(256), 0, S_, part, rows, C, mean, invstd, gamma, beta, 1.f / (float)npix, gscale,
                     accumulate, dgamma, dbeta, coef_ws);
  const bool usez = relu && z != nullptr;
#define HD_APP(R, Z, D)                                                                                                              \
  hipLaunchKernelGGL((bn_bwd_apply_kernel<R, Z, D>), dim3(grid_for(npix * C / 8, TB, 2048)), dim3(TB), 0, S_, (const f16*)dz, (const f16*)z, \
                     (const f16*)y, mean, invstd, gamma, beta, (const float*)coef_ws, (f16*)dy, (f16*)dres, npix, C)
  if (!relu) { if (dres) HD_APP(false, false, true); else HD_APP(false, false, false); }
  else if (usez) { if (dres) HD_APP(true, true, true); else HD_APP(true, true, false); }
  else { if (dres) HD_APP(true, false, true); else HD_APP(true, false, false); }
#undef HD_APP
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_maxpool3x3s2)(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, void* stream) {
  HD_CHECK_ARG(x && y && C % 8 == 0 && Ho == (H + 2 - 3) / 2 + 1 && Wo == (W + 2 - 3) / 2 + 1, "hd_maxpool3x3s2: bad args");
  hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for((int64_t)N * Ho * Wo * C / 8)), dim3(TB), 0, S_, (const f16*)x, (f16*)y, N, H, W, C, Ho, Wo);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_maxpool3x3s2_bwd)(const void* x, const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo, void* stream) {
  HD_CHECK_ARG(x && dy && dx && C % 8 == 0, "hd_maxpool3x3s2_bwd: bad args");
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((int64_t)N * H * W * C / 8)), dim3(TB), 0, S_, (const f16*)x, (const f16*)dy, (f16*)dx, N, H, W, C, Ho, Wo);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_maxpool3x3s2_idx)(const void* x, void* y, void* idx_u8, int N, int H, int W, int C, int Ho, int Wo, void* stream) {
  HD_CHECK_ARG(x && y && idx_u8 && C % 8 == 0 && Ho == (H + 2 - 3) / 2 + 1 && Wo == (W + 2 - 3) / 2 + 1, "hd_maxpool3x3s2_idx: bad args");
  hipLaunchKernelGGL(maxpool_idx_kernel, dim3(grid_for((int64_t)N * Ho * Wo * C / 8)), dim3(TB), 0, S_, (const f16*)x, (f16*)y, (unsigned char*)idx_u8, N, H, W, C, Ho, Wo);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_maxpool3x3s2_bwd_idx)(const void* idx_u8, const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo, void* stream) {
  HD_CHECK_ARG(idx_u8 && dy && dx && C % 8 == 0, "hd_maxpool3x3s2_bwd_idx: bad args");
  hipLaunchKernelGGL(maxpool_bwd_idx_kernel<false>, dim3(grid_for((int64_t)N * H * W * C / 8)), dim3(TB), 0, S_, (const unsigned char*)idx_u8, (const f16*)dy, (const f16*)nullptr, (f16*)dx, N, H, W, C, Ho, Wo);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_maxpool3x3s2_bwd_idx_add)(const void* idx_u8, const void* dy, const void* add, void* dx, int N, int H, int W, int C, int Ho,
                                                   int Wo, void* stream) {
  HD_CHECK_ARG(idx_u8 && dy && add && dx && C % 8 == 0, "hd_maxpool3x3s2_bwd_idx_add: bad args");
  hipLaunchKernelGGL(maxpool_bwd_idx_kernel<true>, dim3(grid_for((int64_t)N * H * W * C / 8)), dim3(TB), 0, S_, (const unsigned char*)idx_u8, (const f16*)dy, (const f16*)add, (f16*)dx, N, H, W, C, Ho, Wo);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_subsample2)(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, void* stream) {
  HD_CHECK_ARG(x && y && C % 8 == 0 && Ho == (H - 1) / 2 + 1 && Wo == (W - 1) / 2 + 1, "hd_subsample2: bad args");
  hipLaunchKernelGGL(subsample2_kernel, dim3(grid_for((int64_t)N * Ho * Wo * C / 8)), dim3(TB), 0, S_, (const f16*)x, (f16*)y, N, H, W, C, Ho, Wo);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_subsample2_bwd)(const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo, int accumulate, void* stream) {
  HD_CHECK_ARG(dy && dx && C % 8 == 0, "hd_subsample2_bwd: bad args");
  hipLaunchKernelGGL(subsample2_bwd_kernel, dim3(grid_for((int64_t)N * H * W * C / 8)), dim3(TB), 0, S_, (const f16*)dy, (f16*)dx, N, H, W, C, Ho, Wo, accumulate);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_nchw_to_nhwc_resize)(const float* x, void* y, int N, int Cr, int H, int W, int Ho, int Wo, int Cp, void* stream) {
  HD_CHECK_ARG(x && y && Cp % 8 == 0 && Cr <= Cp && Cr > 0, "hd_nchw_to_nhwc_resize: bad args");
  float sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
  hipLaunchKernelGGL(nchw_to_nhwc_resize_kernel, dim3(grid_for((int64_t)N * Ho * Wo)), dim3(TB), 0, S_, x, (f16*)y, N, Cr, H, W, Ho, Wo, Cp, sh, sw,
                     (int64_t)Cr * H * W, (int64_t)H * W);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_nchw_to_nhwc_resize_strided)(const float* x, int64_t nstride, int64_t cstride, void* y, int N, int Cr, int H, int W, int Ho,
                                              int Wo, int Cp, void* stream) {
  HD_CHECK_ARG(x && y && Cp % 8 == 0 && Cr <= Cp && Cr > 0 && nstride >= 0 && cstride >= 0, "hd_nchw_to_nhwc_resize_strided: bad args");
  float sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
  hipLaunchKernelGGL(nchw_to_nhwc_resize_kernel, dim3(grid_for((int64_t)N * Ho * Wo)), dim3(TB), 0, S_, x, (f16*)y, N, Cr, H, W, Ho, Wo, Cp, sh, sw,
                     nstride, cstride);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_nchw_to_nhwc_resize_bwd)(const void* dy, float* dx, int N, int Cr, int H, int W, int Ho, int Wo, int Cp,
                                          float gscale, void* stream) {
  HD_CHECK_ARG(dy && dx && Cp % 8 == 0 && Cr <= 8 && Cr > 0, "hd_nchw_to_nhwc_resize_bwd: bad args (Cr<=8)");
  float sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
  hipLaunchKernelGGL(nchw_to_nhwc_resize_bwd_kernel, dim3(grid_for((int64_t)N * H * W)), dim3(TB), 0, S_, (const f16*)dy, dx, N, Cr, H, W, Ho, Wo, Cp, sh, sw, gscale);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_nhwc_to_nchw)(const void* x, float* y, int N, int Cr, int H, int W, int Cp, void* stream) {
  HD_CHECK_ARG(x && y && Cr <= Cp, "hd_nhwc_to_nchw: bad args");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((int64_t)N * H * W)), dim3(TB), 0, S_, (const f16*)x, y, N, Cr, H, W, Cp);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_upsample_add)(const void* a, const void* b, void* y, int N, int H, int W, int C, int Hb, int Wb, void* stream) {
  HD_CHECK_ARG(a && b && y && C % 8 == 0, "hd_upsample_add: bad args");
  float sh = (float)Hb / (float)H, sw = (float)Wb / (float)W;
  hipLaunchKernelGGL(upsample_add_kernel, dim3(grid_for((int64_t)N * H * W * C / 8)), dim3(TB), 0, S_, (const f16*)a, (const f16*)b, (f16*)y, N, H, W, C, Hb, Wb, sh, sw);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_upsample_add_bwd)(const void* dy, void* db, int N, int H, int W, int C, int Hb, int Wb, int accumulate, void* stream) {
  HD_CHECK_ARG(dy && db && C % 8 == 0, "hd_upsample_add_bwd: bad args");
  float sh = (float)Hb / (float)H, sw = (float)Wb / (float)W;
  hipLaunchKernelGGL(upsample_add_bwd_kernel, dim3(grid_for((int64_t)N * Hb * Wb * C / 8)), dim3(TB), 0, S_, (const f16*)dy, (f16*)db, N, H, W, C, Hb, Wb, sh, sw, accumulate);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_upsample2_bwd)(const void* dy_up, void* dx_low, int N, int Hl, int Wl, int C, int Ctot, int c_off, int accumulate, void* stream) {
  HD_CHECK_ARG(dy_up && dx_low && C % 8 == 0 && Ctot % 8 == 0 && c_off % 8 == 0 && c_off + C <= Ctot, "hd_upsample2_bwd: bad args");
  hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(grid_for((int64_t)N * Hl * Wl * C / 8)), dim3(TB), 0, S_, (const f16*)dy_up, (f16*)dx_low, N, Hl, Wl, C, Ctot, c_off, accumulate);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_concat_up_bwd)(const void* dcat, void* dlow, void* dskip, int N, int Hl, int Wl, int Cup, int Cskip, void* stream) {
  HD_CHECK_ARG(dcat && dlow && (dskip || Cskip == 0) && Cup > 0 && Cup % 8 == 0 && Cskip >= 0 && Cskip % 8 == 0, "hd_concat_up_bwd: bad args");
  const int64_t vecs = (int64_t)N * Hl * Wl * (Cup / 8) + (int64_t)N * Hl * 2 * Wl * 2 * (Cskip / 8);
  hipLaunchKernelGGL(concat_up_bwd_kernel, dim3(grid_for(vecs)), dim3(TB), 0, S_, (const f16*)dcat, (f16*)dlow, (f16*)dskip, N, Hl, Wl, Cup, Cskip);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_add_f16)(const void* a, const void* b, void* out, int64_t n, void* stream) {
  HD_CHECK_ARG(a && b && out && n % 8 == 0, "hd_add_f16: bad args");
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 8)), dim3(TB), 0, S_, (const f16*)a, (const f16*)b, (f16*)out, n / 8);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_slice_channels)(const void* x, void* y, int64_t npix, int Ctot, int c_off, int C, int accumulate, void* stream) {
  HD_CHECK_ARG(x && y && C % 8 == 0 && Ctot % 8 == 0 && c_off % 8 == 0 && c_off + C <= Ctot, "hd_slice_channels: bad args");
  hipLaunchKernelGGL(slice_channels_kernel, dim3(grid_for(npix * C / 8)), dim3(TB), 0, S_, (const f16*)x, (f16*)y, npix, Ctot, c_off, C, accumulate);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_sigmoid_bwd_nchw_to_nhwc)(const float* dy, const float* s, void* dlogit, int N, int Cr, int H, int W, int Cp,
                                           float gscale, void* stream) {
  HD_CHECK_ARG(dy && s && dlogit && Cp % 8 == 0 && Cr <= Cp, "hd_sigmoid_bwd_nchw_to_nhwc: bad args");
  hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(grid_for((int64_t)N * H * W)), dim3(TB), 0, S_, dy, s, (f16*)dlogit, N, Cr, H, W, Cp, gscale);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_relu_bwd)(const void* dy, const void* z, void* dx, int64_t n, void* stream) {
  HD_CHECK_ARG(dy && z && dx && n % 8 == 0, "hd_relu_bwd: bad args");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n / 8)), dim3(TB), 0, S_, (const f16*)dy, (const f16*)z, (f16*)dx, n / 8);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_f32_to_f16)(const float* x, void* y, int64_t n, float scale, void* stream) {
  HD_CHECK_ARG(x && y && n > 0, "hd_f32_to_f16: bad args");
  hipLaunchKernelGGL(f32_to_f16_kernel, dim3(grid_for(n)), dim3(TB), 0, S_, x, (f16*)y, n, scale);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_f16_to_f32)(const void* x, float* y, int64_t n, float scale, void* stream) {
  HD_CHECK_ARG(x && y && n > 0, "hd_f16_to_f32: bad args");
  hipLaunchKernelGGL(f16_to_f32_kernel, dim3(grid_for(n)), dim3(TB), 0, S_, (const f16*)x, y, n, scale);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_pad_cast_f32_f16)(const float* x, void* y, int64_t P, int C, int Cp, int64_t rows_per_image, int64_t image_stride, void* stream) {
  HD_CHECK_ARG(x && y && P >= 0 && C > 0 && Cp >= C && rows_per_image > 0 && image_stride >= rows_per_image * C, "hd_pad_cast_f32_f16: bad args");
  if (P == 0) return HD_OK;
  hipLaunchKernelGGL(pad_cast_kernel, dim3(grid_for(P * Cp)), dim3(TB), 0, S_, x, (f16*)y, P, C, Cp, rows_per_image, image_stride);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_pad_cast_f32_f16_multi)(const float* const* x, void* const* y, const int64_t* P, const int* C, const int* Cp,
                                                 const int64_t* rows_per_image, const int64_t* image_stride, int n, void* stream) {
  HD_CHECK_ARG(x && y && P && C && Cp && rows_per_image && image_stride && n >= 1 && n <= PAD_CAST_MAX, "hd_pad_cast_f32_f16_multi: 1 .. %d tensors", PAD_CAST_MAX);
  PadCastTab t;
  int total = 0, m = 0;
  for (int i = 0; i < n; ++i) {
    HD_CHECK_ARG(P[i] >= 0 && C[i] > 0 && Cp[i] >= C[i] && rows_per_image[i] > 0 && image_stride[i] >= rows_per_image[i] * C[i],
                 "hd_pad_cast_f32_f16_multi: bad entry %d", i);
    if (P[i] == 0) continue;                       // (an empty tensor has no storage: its pointers may be null)
    HD_CHECK_ARG(x[i] && y[i], "hd_pad_cast_f32_f16_multi: null pointer in entry %d", i);
    t.x[m] = x[i]; t.y[m] = (f16*)y[i]; t.P[m] = P[i]; t.C[m] = C[i]; t.Cp[m] = Cp[i];
    t.rows_per_image[m] = rows_per_image[i]; t.image_stride[m] = image_stride[i];
    t.first[m] = total;
    total += grid_for(P[i] * Cp[i], TB, 256);
    ++m;
  }
  if (m == 0) return HD_OK;
  t.first[m] = total;
  t.n = m;
  hipLaunchKernelGGL(pad_cast_multi_kernel, dim3(total), dim3(TB), 0, S_, t);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int HD_API(hd_channel_sum_f16)(const void* x, int64_t npix, int C, float* part, int rows, void* stream) {
  HD_CHECK_ARG(x && part && rows > 0 && C % 8 == 0 && pow2(C / 8) && C / 8 <= TB, "hd_channel_sum_f16: bad args");
  hipLaunchKernelGGL(channel_sum_kernel, dim3(rows), dim3(TB), TB * 8 * sizeof(float), S_, (const f16*)x, npix, C, part);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

#ifndef HD_STORE_F32
extern "C" int hd_scale_store(const float* in, float* out, int n, float scale, int accumulate, void* stream) {
  HD_CHECK_ARG(in && out && n > 0, "hd_scale_store: bad args");
  hipLaunchKernelGGL(scale_store_kernel, dim3(hd_cdiv(n, 256)), dim3(256), 0, S_, in, out, n, scale, accumulate);
  HD_CHECK_LAUNCH();
  return HD_OK;
}
#endif
