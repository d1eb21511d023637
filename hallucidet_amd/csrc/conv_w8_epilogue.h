// Epilogue of the 8-wave patch-staged 3x3 families (conv3x3_w8.hip: TH x 8-pixel tiles; conv3x3_m160.hip: 4 x 40-pixel tiles), from the
// point where the WK partial accumulator tiles lie in LDS as fp32 [WK][BM][BN + 4] (row r of a tile = output pixel
// (ty0 + r / TWID, tx0 + r % TWID) of image n_img, column = output channel n0 + c):
//   y = act( mask( sum over partials (+ residual) + bias ) ), optional per-tile BatchNorm sums of the f16-rounded result, the BatchNorm
//   backward sums of hd_conv_args.bs_*, and the pooled / split output of hd_conv_args.out_pool2.
// Moved out of conv3x3_w8.hip verbatim in round 6 (the only changes: the tile width is the template parameter TWID where the code said
// 8, and a tile whose row count is not a multiple of the 512 / (BN / 8) rows a pass covers masks the rows beyond it).
#pragma once
#include "hd_common.h"
#include "conv_params.h"

template <int BM, int BN, int WK, int TWID>
__device__ __forceinline__ void hd_w8_epilogue(ConvP& p, f16* lds, int n_img, int ty0, int tx0, int n0, int tile_m) {
  constexpr int CP = BN + 4;
  const int tid = threadIdx.x;
  float* ct = reinterpret_cast<float*>(lds);
  constexpr int CPR = BN / 8;
  constexpr int RPI = 512 / CPR;
  constexpr int ITER = (BM + RPI - 1) / RPI;          // (the 160-pixel tile: 2.5 passes of 64 rows, the last half masked)
  constexpr bool RAGGED = (BM % RPI) != 0;
  const int cch = tid % CPR, r0 = tid / CPR;
  const int co = n0 + cch * 8;
  const bool cvalid = co < p.Cout;
  const int Cout = p.Cout;
  bool ok[ITER];
  unsigned off[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int row = r0 + it * RPI;
    const int oy = ty0 + row / TWID, ox = tx0 + row % TWID;
    ok[it] = cvalid && oy < p.Ho && ox < p.Wo && (!RAGGED || row < BM);
    off[it] = (unsigned)(((n_img * p.Ho + oy) * p.Wo + ox) * Cout + co);
  }
  const f16* __restrict__ resp = p.res;
  const f16* __restrict__ maskp = p.mask;
  float* __restrict__ statsp = p.stats;
  const int act = p.act;
  f16x8 rv[ITER], mv[ITER];
  if (resp) {
#pragma unroll
    for (int it = 0; it < ITER; ++it)
      if (ok[it]) rv[it] = *reinterpret_cast<const f16x8*>(resp + off[it]);
  }
  if (maskp) {
#pragma unroll
    for (int it = 0; it < ITER; ++it)
      if (ok[it]) mv[it] = *reinterpret_cast<const f16x8*>(maskp + off[it]);
  }
  float bias8[8];
  {
    f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0;
    if (p.bias && cvalid) {
      q0 = *reinterpret_cast<const f32x4*>(p.bias + co);
      q1 = *reinterpret_cast<const f32x4*>(p.bias + co + 4);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) bias8[k] = k < 4 ? q0[k] : q1[k - 4];
  }
  __syncthreads();
  HD_TRACE(12, clock64());

  if (p.pool2) {
    // hd_conv_args.out_pool2 (round 5): the data gradient of a decoder block's first convolution over cat([nearest_2x(a), skip]).  A
    // channel tile lies wholly inside the upsampled half (n0 < pool2: pool2 is a multiple of BN) or wholly inside the skip's:
    //   * upsampled half: the gradient of `a` is the 2 x 2 SUM of this tile's pixels -- a thread adds the four rows (r, r + 1, r + 8,
    //     r + 9: the 8-wide tile holds whole 2 x 2 blocks) of all WK partial tiles in fp32 and stores ONE pooled vector to
    //     y [N, Ho/2, Wo/2, pool2];
    //   * skip half: the unpooled vector goes to y2 [N, Ho, Wo, Cout - pool2].
    // No residual / mask / bias / activation / sums on this path (the dispatcher checks); hd_concat_up_bwd is not launched.
    const int c_up = p.pool2;
    if (n0 < c_up) {
      f16* __restrict__ yq = reinterpret_cast<f16*>(p.y);
      const int Hq = p.Ho >> 1, Wq = p.Wo >> 1;
      for (int pr = r0; pr < BM / 4; pr += RPI) {
        const int py = pr / (TWID / 2), px = pr % (TWID / 2);
        const int row = py * 2 * TWID + px * 2;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
#pragma unroll
        for (int g = 0; g < WK; ++g)
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const int rr = row + (d & 1) + (d >> 1) * TWID;
            s0 += *reinterpret_cast<const f32x4*>(ct + (g * BM + rr) * CP + cch * 8);
            s1 += *reinterpret_cast<const f32x4*>(ct + (g * BM + rr) * CP + cch * 8 + 4);
          }
        const int oy = ty0 + 2 * py, ox = tx0 + 2 * px;
        if (cvalid && oy < p.Ho && ox < p.Wo) {
          f16x8 o;
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] = (f16)(k < 4 ? s0[k] : s1[k - 4]);
          *reinterpret_cast<f16x8*>(yq + (size_t)((n_img * Hq + (oy >> 1)) * Wq + (ox >> 1)) * c_up + co) = o;
        }
      }
    } else {
      f16* __restrict__ y2 = reinterpret_cast<f16*>(p.y2);
      const int cs = Cout - c_up;
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int row = (RAGGED && r0 + it * RPI >= BM) ? r0 : r0 + it * RPI;
        f32x4 s0 = *reinterpret_cast<const f32x4*>(ct + row * CP + cch * 8), s1 = *reinterpret_cast<const f32x4*>(ct + row * CP + cch * 8 + 4);
#pragma unroll
        for (int g = 1; g < WK; ++g) {
          s0 += *reinterpret_cast<const f32x4*>(ct + (g * BM + row) * CP + cch * 8);
          s1 += *reinterpret_cast<const f32x4*>(ct + (g * BM + row) * CP + cch * 8 + 4);
        }
        if (ok[it]) {
          const int oy = ty0 + row / TWID, ox = tx0 + row % TWID;
          f16x8 o;
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] = (f16)(k < 4 ? s0[k] : s1[k - 4]);
          *reinterpret_cast<f16x8*>(y2 + (size_t)((n_img * p.Ho + oy) * p.Wo + ox) * cs + (co - c_up)) = o;
        }
      }
    }
    return;
  }

  // Row phase, written as whole-tile passes (one option test per pass, not per row): every thread first pulls ALL of its ITER rows
  // (x WK partial tiles) out of LDS -- 2*ITER*WK independent 16-byte reads in flight, the accumulator registers are free by now --
  // and then runs straight-line fp32 code over ITER x 8 values.  The per-row form (`if (ok[it]) { read; ...; store; }`) exposed
  // one LDS round trip per row and kept the compiler from scheduling across rows: 5 700 (+2 500 of wave skew at the next barrier)
  // clocks per 256 x 128 tile with BN sums, measured with tools/w8_trace.py.
  float ssum8[8], ssq8[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) ssum8[k] = ssq8[k] = 0.f;
  f16* __restrict__ yp = reinterpret_cast<f16*>(p.y);
  float v[ITER][8];
  {
    f32x4 c0[ITER], c1[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int row = (RAGGED && r0 + it * RPI >= BM) ? r0 : r0 + it * RPI;
      c0[it] = *reinterpret_cast<const f32x4*>(ct + row * CP + cch * 8);
      c1[it] = *reinterpret_cast<const f32x4*>(ct + row * CP + cch * 8 + 4);
    }
#pragma unroll
    for (int g = 1; g < WK; ++g)
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int row = (RAGGED && r0 + it * RPI >= BM) ? r0 : r0 + it * RPI;
        c0[it] += *reinterpret_cast<const f32x4*>(ct + (g * BM + row) * CP + cch * 8);
        c1[it] += *reinterpret_cast<const f32x4*>(ct + (g * BM + row) * CP + cch * 8 + 4);
      }
#pragma unroll
    for (int it = 0; it < ITER; ++it)
#pragma unroll
      for (int k = 0; k < 8; ++k) v[it][k] = k < 4 ? c0[it][k] : c1[it][k - 4];
  }
  if (resp) {
#pragma unroll
    for (int it = 0; it < ITER; ++it)
      if (ok[it]) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[it][k] += (float)rv[it][k];
      }
  }
#pragma unroll
  for (int it = 0; it < ITER; ++it)
#pragma unroll
    for (int k = 0; k < 8; ++k) v[it][k] += bias8[k];
  if (maskp) {
#pragma unroll
    for (int it = 0; it < ITER; ++it)
      if (ok[it]) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[it][k] = ((float)mv[it][k] > 0.f) ? v[it][k] : 0.f;
      }
  }
  if (statsp && p.bs_y) {
    // hd_conv_args.bs_*: this tile of y IS the gradient dz of a BatchNorm unit -- its backward sums (sum dz*m, sum dz*m*xhat; the
    // expressions of bn_bwd_reduce_kernel on the fp16-rounded dz) leave through the statistics rows, and the separate reduction
    // pass (one more read of dz, one launch) is not run.  The unit's y (and z) vectors are pulled row by row: the registers of the
    // tile pass above are still live.
    const f16* __restrict__ byp = p.bs_y;
    const f16* __restrict__ bzp = p.bs_z;
    const bool relu = p.bs_relu != 0;
    float mu[8], is[8], sc[8], sh[8];
    {
      f32x4 m0 = {0.f, 0.f, 0.f, 0.f}, m1 = m0, i0 = m0, i1 = m0, g0 = {1.f, 1.f, 1.f, 1.f}, g1 = g0, b0 = m0, b1 = m0;
      if (cvalid) {
        m0 = *reinterpret_cast<const f32x4*>(p.bs_mean + co); m1 = *reinterpret_cast<const f32x4*>(p.bs_mean + co + 4);
        i0 = *reinterpret_cast<const f32x4*>(p.bs_invstd + co); i1 = *reinterpret_cast<const f32x4*>(p.bs_invstd + co + 4);
        if (p.bs_gamma) { g0 = *reinterpret_cast<const f32x4*>(p.bs_gamma + co); g1 = *reinterpret_cast<const f32x4*>(p.bs_gamma + co + 4); }
        if (p.bs_beta) { b0 = *reinterpret_cast<const f32x4*>(p.bs_beta + co); b1 = *reinterpret_cast<const f32x4*>(p.bs_beta + co + 4); }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        mu[k] = k < 4 ? m0[k] : m1[k - 4];
        is[k] = k < 4 ? i0[k] : i1[k - 4];
        const float g = k < 4 ? g0[k] : g1[k - 4], b = k < 4 ? b0[k] : b1[k - 4];
        sc[k] = g * is[k];
        sh[k] = b - mu[k] * g * is[k];
      }
    }
    // rows in groups of HB: all of a group's y (and z) vectors are requested before the first is used (one exposed round trip per
    // group instead of one per row; the group size is what the register budget of the 2-blocks-per-CU kernels leaves)
    constexpr int HB = (ITER % 4 == 0) ? 4 : (ITER % 3 == 0 ? 3 : (ITER % 5 == 0 ? 5 : (ITER % 2 == 0 ? 2 : 1)));   // a divisor of ITER (1, 2, 3, 4, 5, 8 here)
    static_assert(ITER % HB == 0, "row groups");
#pragma unroll
    for (int h0 = 0; h0 < ITER; h0 += HB) {
      f16x8 yy[HB], zz[HB];
#pragma unroll
      for (int j = 0; j < HB; ++j) {
        // (a clamped, valid address: the value is discarded below.  Per ROW: with 40-pixel-wide tiles a thread's rows sit in different
        //  columns, row 0 may lie outside a ragged map while a later row lies inside)
        const unsigned o = ok[h0 + j] ? off[h0 + j] : 0u;
        yy[j] = *reinterpret_cast<const f16x8*>(byp + o);
        if (bzp) zz[j] = *reinterpret_cast<const f16x8*>(bzp + o);
      }
#pragma unroll
      for (int j = 0; j < HB; ++j) {
        const float keep = ok[h0 + j] ? 1.f : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float gk = (float)(f16)v[h0 + j][k] * keep;
          if (relu) {
            const bool on = bzp ? ((float)zz[j][k] > 0.f) : ((float)(f16)hd_bn_affine((float)yy[j][k], sc[k], sh[k]) > 0.f);
            gk = on ? gk : 0.f;
          }
          const float xh = ((float)yy[j][k] - mu[k]) * is[k];
          ssum8[k] += gk;
          ssq8[k] += gk * xh;
        }
      }
    }
  } else if (statsp) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const float keep = ok[it] ? 1.f : 0.f;        // rows outside the image / channels outside Cout do not count
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float vr = (float)(f16)v[it][k] * keep;
        ssum8[k] += vr;
        ssq8[k] += vr * vr;
      }
    }
  }
  if (act == HD_ACT_RELU) {
#pragma unroll
    for (int it = 0; it < ITER; ++it)
#pragma unroll
      for (int k = 0; k < 8; ++k) v[it][k] = fmaxf(v[it][k], 0.f);
  } else if (act == HD_ACT_SIGMOID) {
#pragma unroll
    for (int it = 0; it < ITER; ++it)
#pragma unroll
      for (int k = 0; k < 8; ++k) v[it][k] = 1.f / (1.f + __expf(-v[it][k]));
  }
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    f16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (f16)v[it][k];
    if (ok[it]) *reinterpret_cast<f16x8*>(yp + off[it]) = o;
  }
  HD_TRACE(13, clock64());
  if (statsp) {
    // per-thread partial sums -> LDS [thread row r0][BN][2] -> 2*BN threads add the RPI rows in a fixed order (deterministic;
    // 48 cross-lane shuffles per thread took 3 600 clocks here, this takes a few hundred)
    // raw barriers: __syncthreads() would first drain this wave's output stores (s_waitcnt vmcnt(0): ~3 000 clocks with every
    // CU storing); only LDS traffic has to be ordered here
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();         // everyone is done reading the C tile
    HD_TRACE(14, clock64());
    float* red = reinterpret_cast<float*>(lds);
    {
      float* d = red + (r0 * BN + cch * 8) * 2;
      f32x4 w0 = {ssum8[0], ssq8[0], ssum8[1], ssq8[1]}, w1 = {ssum8[2], ssq8[2], ssum8[3], ssq8[3]};
      f32x4 w2 = {ssum8[4], ssq8[4], ssum8[5], ssq8[5]}, w3 = {ssum8[6], ssq8[6], ssum8[7], ssq8[7]};
      *reinterpret_cast<f32x4*>(d) = w0;
      *reinterpret_cast<f32x4*>(d + 4) = w1;
      *reinterpret_cast<f32x4*>(d + 8) = w2;
      *reinterpret_cast<f32x4*>(d + 12) = w3;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    HD_TRACE(15, clock64());
    if constexpr (RPI >= 64 && 2 * BN <= 128) {
      // 64 partial rows (the 64-channel-wide tiles): all 512 threads add 16 rows each, 128 threads add the four quarter sums -- a fixed
      // order either way (deterministic); the 64-deep dependent chain of LDS reads below took 1 350 of the tile's 31 000 clocks
      constexpr int G = 512 / (2 * BN), RQ = RPI / G;
      const int c2 = tid % (2 * BN), part = tid / (2 * BN);
      float s = 0.f;
#pragma unroll
      for (int m = 0; m < RQ; ++m) s += red[(part * RQ + m) * BN * 2 + c2];
      float* red2 = red + RPI * BN * 2;
      red2[part * 2 * BN + c2] = s;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (tid < 2 * BN) {
        float t = red2[tid];
#pragma unroll
        for (int g = 1; g < G; ++g) t += red2[g * 2 * BN + tid];
        const int c = tid >> 1;
        if (n0 + c < Cout) statsp[((size_t)tile_m * 2 + (tid & 1)) * Cout + n0 + c] = t;
      }
    } else
    if (tid < 2 * BN) {
      float s = 0.f;
#pragma unroll 8
      for (int m = 0; m < RPI; ++m) s += red[m * BN * 2 + tid];
      const int c = tid >> 1;
      if (n0 + c < Cout) statsp[((size_t)tile_m * 2 + (tid & 1)) * Cout + n0 + c] = s;
    }
  }
}
