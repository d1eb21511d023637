// 3x3 / stride 1 / pad 1 convolution with LDS-STAGED INPUT PATCHES (v_mfma_f32_32x32x16_f16), Cin % 64 == 0.
//
// Why: the implicit-GEMM kernels (conv_igemm_bk*.hip) fill LDS with an im2col row per output pixel and tap, so every input
// element crosses the global->LDS path nine times; ablation shows that fill, ~21 B/clk/CU, is what bounds
// them.  Here a block owns a TH x TW = 8 x 16 patch of output pixels of ONE image and, per 64-channel chunk, stages the
// (TH+2) x (TW+2) input patch ONCE; the nine taps then read shifted windows of it:
//     per 64 channels and 9 taps:  patch 180 px x 128 B = 23 KB  +  weights 9 x BN x 128 B
// instead of 9 x (128 x 128 B + BN x 128 B): 1.7x (BN = 128) to 2.3x (BN = 64) fewer bytes filled per FLOP.
//
// Layout / schedule
//   * patch stage: pixel-major, 8 slots of 16 B (= 64 channels) per pixel, filled by LDS-DMA (lane-linear destination:
//     unit u = pixel*8 + slot lands at byte u*16); bank swizzle on the SOURCE side: slot s of pixel pp holds logical
//     channel group s ^ ((pp >> 1) & 7), so the 16 lanes one ds_read_b128 phase serves -- 16 consecutive patch pixels,
//     same k group -- hit 16 distinct 16-byte bank groups ((pp & 1) * 8 + slot is a bijection of pp mod 16; keying on
//     pp & 7 instead costs a 2-way conflict on every A read: measured);
//   * zero padding and ragged image edges are out-of-range buffer offsets (hardware zero fill), as in the igemm kernels;
//   * K order: channel chunk (outer) x tap (inner); one barrier per (chunk, tap) step; weights of step s+1 and -- at
//     tap 0 -- the patch of the next chunk are in flight behind the MFMAs of step s (counted vmcnt);
//   * MFMA tile rows are pixels (py, px) = (wave row + r/16, r%16): the A fragment address is the patch pixel
//     (py+ky)*(TW+2) + (px+kx); B fragments as in the igemm kernels;
//   * epilogue identical in function to the igemm vector path (bias, residual, ReLU mask, activation, BN partial sums
//     per tile, fp32 tile in LDS, 16-byte stores).
#include "hd_common.h"
#include "conv_params.h"

namespace {

constexpr int TH = 8, TW = 16, BM = TH * TW;        // 128 output pixels per block
constexpr int PW_ = TW + 2, PH_ = TH + 2;            // patch extent
constexpr int PPX = PW_ * PH_;                       // 180 patch pixels
constexpr int PUNITS = PPX * 8;                      // 16-byte units per patch chunk (1440)
constexpr int PPIECES = (PUNITS + 255) / 256;        // DMA pieces per wave per chunk (6)
constexpr int PATCH_HALVES = PPIECES * 256 * 8;      // stage size in halves (12288 = 24 KiB)
constexpr int LDS_ROW = 64;
constexpr unsigned OOB = 0xFFFFFFF0u;

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, f16* lds_dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds_dst, 16, voff, 0, 0, 0);
}

template <int BN>
__global__ __launch_bounds__(256) void conv3x3_patch_kernel(ConvP p) {
  constexpr int WM = 2, WN = 2;
  constexpr int MT = 2;                         // 2 x 32 pixels per wave = 4 patch rows
  constexpr int NT = BN / (WN * 32);            // 2 (BN=128) or 1 (BN=64)
  constexpr int B_LOADS = BN * 8 / 256;         // 4 or 2 DMA pieces per wave per step
  constexpr int BSTAGE = BN * LDS_ROW;          // halves
  constexpr int PIPE_HALVES = 2 * PATCH_HALVES + 2 * BSTAGE;
  constexpr int LDS_HALVES = PIPE_HALVES > BM * BN * 2 ? PIPE_HALVES : BM * BN * 2;
  __shared__ __attribute__((aligned(1024))) f16 lds[LDS_HALVES];
  f16* const patch0 = lds;
  f16* const bst0 = lds + 2 * PATCH_HALVES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
    bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  }
  const int tile_m = bid / p.gn, tile_n = bid - tile_m * p.gn;
  const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
  const int n_img = tile_m / (tiles_x * tiles_y);
  const int trem = tile_m - n_img * tiles_x * tiles_y;
  const int ty0 = (trem / tiles_x) * TH, tx0 = (trem % tiles_x) * TW;
  const int n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.x), 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.w), 0, p.wbytes, 0x00020000);

  // ---- patch fill: this lane's PPIECES units (fixed pixel / slot for the whole K loop; only the channel chunk moves)
  unsigned pbase[PPIECES];      // byte offset of (pixel, logical channel group) at channel chunk 0, or OOB
#pragma unroll
  for (int i = 0; i < PPIECES; ++i) {
    const int u = (i * 4 + wave) * 64 + lane;
    const int pp = u >> 3, slot = u & 7;
    const int iy = ty0 - 1 + pp / PW_, ix = tx0 - 1 + pp % PW_;
    const bool v = (u < PUNITS) && ((unsigned)iy < (unsigned)p.Hin) && ((unsigned)ix < (unsigned)p.Win);
    const int cg = slot ^ ((pp >> 1) & 7);
    pbase[i] = v ? (unsigned)((((size_t)n_img * p.Hin + iy) * p.Win + ix) * p.C1 + cg * 8) * 2u : OOB;
  }
  // ---- weight fill: B_LOADS rows per lane; slot swizzle by row as in the igemm kernels
  const int j = (tid & 7) ^ ((tid >> 4) & 7);
  unsigned wbase[B_LOADS];
#pragma unroll
  for (int i = 0; i < B_LOADS; ++i) {
    const int co = n0 + (tid >> 3) + i * 32;
    wbase[i] = co < p.Cout ? (unsigned)co * (unsigned)p.Ktot * 2u + (unsigned)j * 16u : OOB;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int ncc = p.C1 >> 6;                    // 64-channel chunks
  const int nsteps = ncc * 9;
  auto load_patch = [&](int cc, int stage) {
    f16* dst = patch0 + stage * PATCH_HALVES + wave * 512;       // piece i of this wave: units (i*4+wave)*64 ..
#pragma unroll
    for (int i = 0; i < PPIECES; ++i)
      dma16(rx, dst + i * 2048, (pbase[i] != OOB && cc < ncc) ? pbase[i] + (unsigned)cc * 128u : OOB);
  };
  auto load_b = [&](int cc, int tap, int stage) {
    f16* dst = bst0 + stage * BSTAGE + wave * 512;
    const bool v = cc < ncc;
    const unsigned koff = (unsigned)(tap * p.C1 + cc * 64) * 2u;
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) dma16(rw, dst + i * 2048, (wbase[i] != OOB && v) ? wbase[i] + koff : OOB);
  };

  // ---- fragment addressing
  const int frow = lane & 31, fh = lane >> 5;
  int apix[MT];                                 // patch pixel of this lane's row at tap (0,0)
#pragma unroll
  for (int a = 0; a < MT; ++a) apix[a] = (wm * 4 + a * 2 + (frow >> 4)) * PW_ + (frow & 15);
  const int bswz = (frow >> 1) & 7;

  auto compute = [&](int pstage, int bstage, int tap) {
    const f16* sp = patch0 + pstage * PATCH_HALVES;
    const f16* sb = bst0 + bstage * BSTAGE;
    const int toff = (tap / 3) * PW_ + (tap % 3);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      f16x8 af[MT], bf[NT];
#pragma unroll
      for (int a = 0; a < MT; ++a) {
        const int pp = apix[a] + toff;
        af[a] = *reinterpret_cast<const f16x8*>(sp + pp * 64 + (((ks * 2 + fh) ^ ((pp >> 1) & 7)) * 8));
      }
#pragma unroll
      for (int b = 0; b < NT; ++b)
        bf[b] = *reinterpret_cast<const f16x8*>(sb + (wn * NT * 32 + b * 32 + frow) * LDS_ROW + (((ks * 2 + fh) ^ bswz) * 8));
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
  };

  // ---- pipeline: patch(0), B(0) up front; step s issues B(s+1) and, at tap 0, patch(cc+1)
  load_patch(0, 0);
  load_b(0, 0, 0);
  int cc = 0, tap = 0;
  for (int s = 0; s < nsteps; ++s) {
    // in-order completion: everything up to B(s) must have landed; a patch issued at the previous step (after B(s)) may fly
    if (tap == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    int ncc_ = cc, ntap = tap + 1;
    if (ntap == 9) {
      ntap = 0;
      ++ncc_;
    }
    load_b(ncc_, ntap, (s + 1) & 1);
    if (tap == 0) load_patch(cc + 1, (cc + 1) & 1);
    compute(cc & 1, s & 1, tap);
    cc = ncc_;
    tap = ntap;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---------------- epilogue (NHWC f16, Cout % 8 == 0) ----------------
  float* ct = reinterpret_cast<float*>(lds);   // [BM][BN] fp32, row = py*16 + px
#pragma unroll
  for (int b = 0; b < NT; ++b)
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = wn * NT * 32 + b * 32 + (lane & 31);
        ct[row * BN + col] = acc[a][b][r];
      }
  __syncthreads();
  constexpr int CPR = BN / 8;
  constexpr int RPI = 256 / CPR;
  const int cch = tid % CPR, r0 = tid / CPR;
  const int co = n0 + cch * 8;
  const bool cvalid = co < p.Cout;
  float bias8[8], ssum8[8], ssq8[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    bias8[k] = (p.bias && cvalid) ? p.bias[co + k] : 0.f;
    ssum8[k] = ssq8[k] = 0.f;
  }
  if (cvalid) {
#pragma unroll 4
    for (int row = r0; row < BM; row += RPI) {
      const int oy = ty0 + (row >> 4), ox = tx0 + (row & 15);
      if (oy >= p.Ho || ox >= p.Wo) continue;
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(ct + row * BN + cch * 8);
      const f32x4 c1 = *reinterpret_cast<const f32x4*>(ct + row * BN + cch * 8 + 4);
      float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
      const size_t off = (((size_t)n_img * p.Ho + oy) * p.Wo + ox) * p.Cout + co;
      if (p.res) {
        const f16x8 rv = *reinterpret_cast<const f16x8*>(p.res + off);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += (float)rv[k];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += bias8[k];
      if (p.mask) {
        const f16x8 mv = *reinterpret_cast<const f16x8*>(p.mask + off);
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (!((float)mv[k] > 0.f)) v[k] = 0.f;
      }
      f16x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (p.stats) {
          float vr = (float)(f16)v[k];
          ssum8[k] += vr;
          ssq8[k] += vr * vr;
        }
        float w = v[k];
        if (p.act == HD_ACT_RELU) w = fmaxf(w, 0.f);
        else if (p.act == HD_ACT_SIGMOID) w = 1.f / (1.f + __expf(-w));
        o[k] = (f16)w;
      }
      *reinterpret_cast<f16x8*>(reinterpret_cast<f16*>(p.y) + off) = o;
    }
  }
  if (p.stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);   // [RPI][BN][2]
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      red[(r0 * BN + cch * 8 + k) * 2 + 0] = ssum8[k];
      red[(r0 * BN + cch * 8 + k) * 2 + 1] = ssq8[k];
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.Cout) {
      float s = 0.f, s2 = 0.f;
      for (int m = 0; m < RPI; ++m) {
        s += red[(m * BN + tid) * 2 + 0];
        s2 += red[(m * BN + tid) * 2 + 1];
      }
      p.stats[((size_t)tile_m * 2 + 0) * p.Cout + n0 + tid] = s;
      p.stats[((size_t)tile_m * 2 + 1) * p.Cout + n0 + tid] = s2;
    }
  }
}

}  // namespace

int hd_conv_patch_tiles(const ConvP& p) { return p.N * hd_cdiv(p.Ho, TH) * hd_cdiv(p.Wo, TW); }

void hd_conv_launch_patch(ConvP& p, hipStream_t s) {
  p.gm = hd_conv_patch_tiles(p);
  if (p.Cout > 64) {
    p.gn = hd_cdiv(p.Cout, 128);
    hipLaunchKernelGGL((conv3x3_patch_kernel<128>), dim3(p.gm * p.gn), dim3(256), 0, s, p);
  } else {
    p.gn = 1;
    hipLaunchKernelGGL((conv3x3_patch_kernel<64>), dim3(p.gm * p.gn), dim3(256), 0, s, p);
  }
}
