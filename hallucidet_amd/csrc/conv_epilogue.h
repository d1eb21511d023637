// Epilogue shared by the implicit-GEMM convolution kernels (conv_igemm_bk32.hip / conv_igemm_bk64.hip):
//   y = act( mask( acc (+ residual) + bias ) ),  optional per-tile BatchNorm partial sums of the f16-rounded pre-activation.
//
// Vector path (NHWC f16 output, Cout % 8 == 0): accumulators -> LDS fp32 tile -> each thread owns 8 consecutive output
// channels of BM/RPI pixels: 16-byte residual / mask loads and 16-byte coalesced stores (per-lane 2-byte stores cost
// 4-5x the HBM time on the 16/32-channel decoder layers).  Measured with per-block stamps (tools/conv_trace.py) the
// first version of this path took 5 700 - 13 500 cycles per block, 11-33 % of a block's life: every row iteration
// waited for its own residual / mask load and the per-element flag tests compiled to branches.  Now every residual /
// mask row of the thread is requested BEFORE the LDS transpose (they land while the tile is written and read back),
// the option tests are hoisted to one uniform branch per 8-vector, and the BatchNorm partial sums are reduced with
// lane shuffles + one 4-row LDS step instead of a 16-64-row serial LDS walk.
// Direct path (NCHW fp32 output or ragged channel counts: head / RPN / predictor outputs, all tiny): per-lane stores.
#pragma once
#include "conv_params.h"

template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, f32x16 (&acc)[BM / (WM * 32)][BN / (WN * 32)], f16* lds,
                                              int m0, int n0, int tile_m, int HoWo) {
  constexpr int MT = BM / (WM * 32);
  constexpr int NT = BN / (WN * 32);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  if (p.out_mode == HD_OUT_NHWC_F16 && (p.Cout & 7) == 0) {
    constexpr int CPR = BN / 8;            // 8-channel chunks per tile row
    constexpr int RPI = 256 / CPR;         // rows covered per iteration
    constexpr int ITER = BM / RPI;         // rows per thread
    const int cch = tid % CPR, r0 = tid / CPR;
    const int co = n0 + cch * 8;
    const bool cvalid = co < p.Cout;       // Cout % 8 == 0 => whole chunk valid
    const f16* __restrict__ resp = p.res;
    const f16* __restrict__ maskp = p.mask;
    float* __restrict__ statsp = p.stats;
    const int act = p.act;
    const int Cout = p.Cout;
    bool ok[ITER];
    size_t offs[ITER];     // element offset of (output pixel of row r0 + it*RPI, channel co)
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int m = m0 + r0 + it * RPI;
      ok[it] = cvalid && (m < p.M);
      offs[it] = (size_t)(p.par ? hd_par_pixel(p, ok[it] ? m : 0) : m) * Cout + co;
    }

    // residual / mask rows: all requested up front
    f16x8 rv[ITER], mv[ITER];
    if (resp) {
#pragma unroll
      for (int it = 0; it < ITER; ++it)
        if (ok[it]) rv[it] = *reinterpret_cast<const f16x8*>(resp + offs[it]);
    }
    if (maskp) {
#pragma unroll
      for (int it = 0; it < ITER; ++it)
        if (ok[it]) mv[it] = *reinterpret_cast<const f16x8*>(maskp + offs[it]);
    }
    float bias8[8];
    {   // two 16-byte loads: a VMEM instruction costs the issuing wave ~150 cycles whatever its width
      f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0;
      if (p.bias && cvalid) {
        q0 = *reinterpret_cast<const f32x4*>(p.bias + co);
        q1 = *reinterpret_cast<const f32x4*>(p.bias + co + 4);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) bias8[k] = k < 4 ? q0[k] : q1[k - 4];
    }

    float* ct = reinterpret_cast<float*>(lds);   // [BM][BN] fp32
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * MT * 32 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int col = wn * NT * 32 + b * 32 + (lane & 31);
          ct[row * BN + col] = acc[a][b][r];
        }
    __syncthreads();
    HD_TRACE(8, clock64());

    float ssum8[8], ssq8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) ssum8[k] = ssq8[k] = 0.f;
    f16* __restrict__ yp = reinterpret_cast<f16*>(p.y);
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      if (ok[it]) {
        const int row = r0 + it * RPI;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(ct + row * BN + cch * 8);
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(ct + row * BN + cch * 8 + 4);
        float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
        if (resp) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] += (float)rv[it][k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += bias8[k];
        if (maskp) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = ((float)mv[it][k] > 0.f) ? v[k] : 0.f;
        }
        if (statsp) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float vr = (float)(f16)v[k];
            ssum8[k] += vr;
            ssq8[k] += vr * vr;
          }
        }
        if (act == HD_ACT_RELU) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
        } else if (act == HD_ACT_SIGMOID) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = 1.f / (1.f + __expf(-v[k]));
        }
        f16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (f16)v[k];
        *reinterpret_cast<f16x8*>(yp + offs[it]) = o;
      }
    }
    HD_TRACE(9, clock64());
    if (statsp) {
      // per-thread partial sums -> LDS [thread row r0][BN][2] -> 2*BN threads add the RPI rows in a fixed order (deterministic).
      // (The first version folded rows with 32-48 cross-lane shuffles per thread: ~3 000 clocks per block, measured with
      // tools/w8_trace.py on the 8-wave kernels' copy of this code.)  Raw barriers: only LDS traffic is ordered here, a
      // __syncthreads() would also wait for this wave's output stores.
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                 // everyone is done reading the C tile
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
      if (tid < 2 * BN) {
        float s = 0.f;
#pragma unroll 8
        for (int m = 0; m < RPI; ++m) s += red[m * BN * 2 + tid];
        const int c = tid >> 1;
        if (n0 + c < Cout) statsp[((size_t)tile_m * 2 + (tid & 1)) * Cout + n0 + c] = s;
      }
    }
    return;
  }

  // Direct path
  float ssum[NT], ssq[NT];
#pragma unroll
  for (int b = 0; b < NT; ++b) ssum[b] = ssq[b] = 0.f;

#pragma unroll
  for (int b = 0; b < NT; ++b) {
    const int col = wn * NT * 32 + b * 32 + (lane & 31);
    const int co = n0 + col;
    const bool cvalid = co < p.Cout;
    const float bias = (p.bias && cvalid) ? p.bias[co] : 0.f;
#pragma unroll
    for (int a = 0; a < MT; ++a) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * MT * 32 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int mrow = m0 + row;
        const int pix = (p.par && mrow < p.M) ? hd_par_pixel(p, mrow) : mrow;
        if (mrow < p.M && cvalid) {
          float v = acc[a][b][r];
          if (p.res) v += (float)p.res[(size_t)pix * p.Cout + co];
          v += bias;
          if (p.mask && !((float)p.mask[(size_t)pix * p.Cout + co] > 0.f)) v = 0.f;
          if (p.stats) {
            float vr = (float)(f16)v;
            ssum[b] += vr;
            ssq[b] += vr * vr;
          }
          if (p.act == HD_ACT_RELU) v = fmaxf(v, 0.f);
          else if (p.act == HD_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
          if (p.out_mode == HD_OUT_NHWC_F16) {
            reinterpret_cast<f16*>(p.y)[(size_t)pix * p.Cout + co] = (f16)v;
          } else if (p.out_mode == HD_OUT_NHWC_F32) {
            reinterpret_cast<float*>(p.y)[(size_t)pix * p.Cout + co] = v;
          } else {
            int n = pix / HoWo;
            int rem = pix - n * HoWo;
            reinterpret_cast<float*>(p.y)[((size_t)n * p.Cout + co) * HoWo + rem] = v;
          }
        }
      }
    }
  }

  if (p.stats) {
    // reduce the two lane halves, then across the WM waves that share columns
    float* red = reinterpret_cast<float*>(lds);  // [WM][BN][2]
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      float s = ssum[b] + __shfl_xor(ssum[b], 32);
      float s2 = ssq[b] + __shfl_xor(ssq[b], 32);
      if (lane < 32) {
        int col = wn * NT * 32 + b * 32 + lane;
        red[(wm * BN + col) * 2 + 0] = s;
        red[(wm * BN + col) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    if (tid < BN) {
      int co = n0 + tid;
      if (co < p.Cout) {
        float s = 0.f, s2 = 0.f;
#pragma unroll
        for (int m = 0; m < WM; ++m) {
          s += red[(m * BN + tid) * 2 + 0];
          s2 += red[(m * BN + tid) * 2 + 1];
        }
        p.stats[((size_t)tile_m * 2 + 0) * p.Cout + co] = s;
        p.stats[((size_t)tile_m * 2 + 1) * p.Cout + co] = s2;
      }
    }
  }
}
