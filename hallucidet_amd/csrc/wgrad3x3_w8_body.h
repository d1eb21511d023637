// (body of wgrad3x3_w8.hip, shared with the fused launch of conv3x3_w8.hip)
// Weight gradient of the 3x3 / stride 1 / pad 1 layers with >= 64 channels on both sides, 8-wave family with LDS-STAGED INPUT
// PATCHES:   dW[co][tap][ci] = sum_pixels dY[pixel][co] * X[pixel + tap][ci]
//
// The general kernel (wgrad.hip) stages 32-pixel reduction tiles through registers (4 global loads + 4 LDS stores per thread
// for 8 MFMAs) and re-gathers the input once per tap: 13-17 % MFMA-pipe utilisation (profiles/r02_conv_mfma_util.json).  Here a
// block owns one 64 (co) x 64 (ci) weight tile for ALL nine taps and walks a range of 16 x 8-pixel tiles:
//   * per pixel tile it stages, by LDS-DMA in their natural [pixel][channel] layout, the 128 x 64 dY tile (16 KiB) and the
//     (16+2) x (8+2) input patch of the 64-channel chunk (23 KiB) ONCE -- the nine taps read shifted windows of the patch;
//     decoder layers gather the patch from the nearest-2x upsampled tensor or the skip tensor in place (decoder.py:38-41);
//   * MFMA fragments need 8 consecutive PIXELS per lane (the reduction index is the slow axis of both operands): produced by the
//     transposing LDS read ds_read_b64_tr_b16 (four pixel rows x 16 channels per 16-lane group); bank swizzle on the DMA source
//     side: 16-byte slot s of pixel p holds channel group s ^ (4 * ((p >> 1) & 1)), which puts the four pixel rows of a 32-lane
//     half into the four 64-byte quarters of the 256-byte bank row (conflict-free);
//   * rounds 4-5: 8 waves = 2 (ci halves) x 2 (tap groups: taps 0-4 / 5-8) x 2 (pixel halves of the tile); a wave keeps BOTH co halves of
//     its taps -- ten (eight) 32 x 32 fp32 accumulators, 160 registers -- across the whole tile range, the two pixel halves met once, in
//     LDS, at the end.  Round 6: four such waves (no pixel halves) are the CONSUMERS, four PRODUCER waves issue the LDS-DMA (see the body).  (Rounds 2-4: 2 co x 2 ci x 2 pixel halves, nine taps per wave: every MFMA needed its own transposed patch fragment, 80
//     ds_read_b64_tr_b16 per wave and tile, and the kernel ran at the rate of those reads -- 2.3 us per tile, MFMA utilisation 0.26.  With
//     both co halves in one wave a patch fragment feeds TWO MFMAs: 56 / 48 reads per wave and tile.  Every output element still sums its
//     pixels in the same order: results are bit-identical to the old decomposition.)
//   * one fp32 partial per block -> slab[slice][co][tap*Cin + ci], summed by hd_wgrad_reduce (deterministic, as before).
#pragma once
#include "hd_common.h"
#include <type_traits>

namespace hd_wg8 {

constexpr int TH = 16, TW = 8, PW = 10, PH = TH + 2;
constexpr int PPX = PH * PW;                    // 180 patch pixels
constexpr int XPIECES = (PPX * 8 + 63) / 64;    // 23 one-KiB pieces
constexpr int YPIECES = TH * TW * 8 / 64;       // 16
constexpr int XSTAGE = XPIECES * 512;           // halves
constexpr int YSTAGE = YPIECES * 512;
constexpr int STAGE = XSTAGE + YSTAGE;
constexpr int NPIECES = XPIECES + YPIECES;      // 39 per tile
constexpr int PPWV = (NPIECES + 7) / 8;         // 5 per wave
constexpr unsigned OOBB = 0x80000000u;

struct Wg8P {
  const f16* x;
  const f16* x2;
  const f16* dy;
  float* slab;
  float* dw;        // direct OIHW output (nsplit == 1) or nullptr
  float dw_scale;
  int N, Hsrc, Wsrc, H, W, C1, C2, Cin, Cout, Ktot;
  int tiles_x, tiles_y, ntiles, per_split, dual;
  unsigned xbytes, x2bytes, dybytes;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, f16* lds_dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds_dst, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ f16x8 tr_pair(const char* p0, const char* p1) {
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
  f16x4 fa = __builtin_bit_cast(f16x4, a), fb = __builtin_bit_cast(f16x4, b);
  f16x8 r = {fa[0], fa[1], fa[2], fa[3], fb[0], fb[1], fb[2], fb[3]};
  return r;
}
__device__ __forceinline__ int swz4(int p) { return ((p >> 1) & 1) << 2; }

constexpr int NSTAGE = 3;
constexpr int LDS_HALVES = NSTAGE * STAGE;      // 3 x 39 KiB tile stages (117 KiB): one block per CU

// One block of the grid (bx = pixel split, by = (co chunk, ci chunk)); `lds`: LDS_HALVES halves, 1 KiB aligned.  A device function so
// that the fused data-gradient + weight-gradient launch (conv3x3_w8.hip) can run it in the blocks behind its convolution tiles.
//
// Round 6: PRODUCER / CONSUMER wave roles (as conv3x3_m160.hip, and for the same measured reason: every 1-KiB LDS-DMA piece holds its
// issuing wave 100 - 185 clocks, five pieces per wave and tile that the eight-way split of the MFMA work could not hide -- the multi-layer
// grids ran at 0.34 MFMA utilisation, ~4 600 clocks per 128-pixel tile for 2 304 of matrix work).
//   * waves 4-7 (producers): the 39 pieces of tile t + 2 (ten each) into stage (t + 2) % 3, then a counted wait for tile t + 1, one barrier
//     per tile;
//   * waves 0-3 (consumers, one per SIMD) = 2 (ci halves) x 2 (tap groups: taps 0-4 / 5-8), both co halves, ALL 128 pixels of the tile
//     (eight 16-pixel K steps): ten (eight) 32 x 32 fp32 accumulators, 160 registers, exactly as before -- but a consumer now owns its
//     weights outright: the two pixel halves no longer meet in LDS at the end (a 147 KB exchange and two barriers per block gone), and it
//     issues no vector-memory instruction inside the loop.
// Every output element sums its pixels tile by tile, row pair by row pair in one chain (rounds 4-5: two half-tile chains added at the
// end) -- same products, another fp32 order; hd_wgrad_multi / the fused grid still run THIS body, so they stay bit-identical to separate
// hd_wgrad launches.
__device__ __forceinline__ void wgrad3x3_w8_body(const Wg8P& p, f16* lds, int bx, int by) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool consumer = wave < 4;
  const int ci_chunks = p.Cin >> 6;
  const int cchunk = by % ci_chunks, ochunk = by / ci_chunks;
  const int ci0 = cchunk * 64, co0 = ochunk * 64;
  const bool second = p.dual && ci0 >= p.C1;           // this ci chunk lives in the skip tensor
  const int t_begin = bx * p.per_split;
  int t_end = t_begin + p.per_split;
  if (t_end > p.ntiles) t_end = p.ntiles;
  const int nt = t_end > t_begin ? t_end - t_begin : 0;

  if (!consumer) {
    // =====================================================  PRODUCER  =====================================================
    const int pw = wave - 4;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(second ? p.x2 : p.x), 0, second ? p.x2bytes : p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.dy), 0, p.dybytes, 0x00020000);
    // ---- this wave's DMA pieces of a tile: piece q = k*4 + pw; q < XPIECES: input patch, else dY tile
    constexpr int PPWV4 = (NPIECES + 3) / 4;      // 10 per producer (the last producer: 9)
    int upix[PPWV4];              // patch / tile pixel of this lane's unit
    unsigned ucol[PPWV4];         // byte offset of its channel group within the pixel's source row
#pragma unroll
    for (int k = 0; k < PPWV4; ++k) {
      const int q = k * 4 + pw;
      const int u = (q < XPIECES ? q : q - XPIECES) * 64 + lane;
      const int px = u >> 3, slot = u & 7;
      upix[k] = px;
      const int cg = (slot ^ swz4(px)) & 7;
      ucol[k] = q < XPIECES ? (unsigned)((second ? ci0 - p.C1 : ci0) + cg * 8) * 2u : (unsigned)(co0 + cg * 8) * 2u;
    }
    const int srcC = second ? p.C2 : p.C1;
    auto issue_tile = [&](int t, int stage) {
      const bool live = t < t_end;
      const int tt = live ? t : 0;
      const int n = tt / (p.tiles_x * p.tiles_y);
      const int rem = tt - n * p.tiles_x * p.tiles_y;
      const int tyi = rem / p.tiles_x;
      const int ty0 = tyi * TH, tx0 = (rem - tyi * p.tiles_x) * TW;
      f16* sx = lds + stage * STAGE;
      f16* sy = sx + XSTAGE;
#pragma unroll
      for (int k = 0; k < PPWV4; ++k) {
        const int q = k * 4 + pw;
        if (q >= NPIECES) break;
        if (q < XPIECES) {
          const int pp = upix[k];
          const int y = (pp * 6554) >> 16, x = pp - y * PW;
          const int iy = ty0 - 1 + y, ix = tx0 - 1 + x;
          const bool v = live && pp < PPX && ((unsigned)iy < (unsigned)p.H) && ((unsigned)ix < (unsigned)p.W);
          unsigned off;
          if (p.dual && !second) off = (unsigned)(((n * p.Hsrc + (iy >> 1)) * p.Wsrc + (ix >> 1)) * srcC) * 2u;
          else off = (unsigned)(((n * p.H + iy) * p.W + ix) * srcC) * 2u;
          dma16(rx, sx + q * 512, v ? off + ucol[k] : OOBB);
        } else {
          const int pix = upix[k];
          const int oy = ty0 + (pix >> 3), ox = tx0 + (pix & 7);
          const bool v = live && oy < p.H && ox < p.W;
          dma16(rdy, sy + (q - XPIECES) * 512, v ? (unsigned)(((n * p.H + oy) * p.W + ox) * p.Cout) * 2u + ucol[k] : OOBB);
        }
      }
    };
    // counted wait: everything but the pieces of the tile just issued has landed (10 pieces; the last producer has 9)
    auto wait_older = [&]() {
      if ((NPIECES & 3) != 0 && pw >= (NPIECES & 3)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPWV4 - 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPWV4) : "memory");
    };
    issue_tile(t_begin, 0);
    issue_tile(t_begin + 1, 1);
    wait_older();                               // tile t_begin has landed
    __builtin_amdgcn_s_barrier();
    for (int i = 0; i < nt; ++i) {
      // stage (i + 2) % 3 held tile i - 1: every consumer finished with it before the barrier that ended iteration i - 1
      issue_tile(t_begin + i + 2, (i + 2) % NSTAGE);
      wait_older();                             // tile i + 1 has landed
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // =======================================================  CONSUMER  =======================================================
  const int wci = wave & 1, wtap = wave >> 1;           // ci half, tap group
  // ---- transposed-read geometry: 16-lane group g = lane >> 4 -> k half th = g >> 1 (pixel row of the 2 x 8 block), column half
  //      g & 1; within the group lane 4q+p addresses pixel q (x = q, second read x = q + 4), columns 4p .. 4p+3
  const int li = lane & 15, g = lane >> 4;
  const int tq = li >> 2, tp = li & 3, th = g >> 1, thalf = g & 1;
  const char* lb = reinterpret_cast<const char*>(lds);

  // The tap group is wave-uniform: two instantiations of the same code (T0 = first tap, NT = taps of this wave).
  auto run = [&](auto t0c, auto ntc) {
    constexpr int T0 = decltype(t0c)::value, NT = decltype(ntc)::value;
    // byte offsets (within a stage) for K step 0 (tile rows 0-1); K step s adds s * 2 rows
    unsigned ya[2][2], xa[NT][2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int y = th, x = tq + 4 * r;
      const int pix = y * 8 + x;
#pragma unroll
      for (int coh = 0; coh < 2; ++coh) {
        const int cb = coh * 64 + thalf * 32 + tp * 8;               // byte column within the 128-byte dY row (co half, 16-col half, 4-col group)
        ya[coh][r] = (unsigned)(XSTAGE * 2 + pix * 128 + (((cb >> 4) ^ swz4(pix)) << 4) + (cb & 15));
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int tap = T0 + t;
        const int pp = (y + tap / 3) * PW + x + tap % 3;
        const int cx = wci * 64 + thalf * 32 + tp * 8;
        xa[t][r] = (unsigned)(pp * 128 + (((cx >> 4) ^ swz4(pp)) << 4) + (cx & 15));
      }
    }

    f32x16 acc[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int coh = 0; coh < 2; ++coh)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][coh][r] = 0.f;

    __builtin_amdgcn_s_barrier();                 // the first tile has landed
    __builtin_amdgcn_s_setprio(1);
    for (int i = 0; i < nt; ++i) {
      // K steps: 2 pixel rows (16 pixels) each, eight per tile, run as two halves of four (one software-pipelined stream of 2 * NT * 4
      // MFMAs per half, as in rounds 4-5 -- all eight in one stream spilled 130 registers); row step = 8 dY pixels (1 KiB) / 10 patch
      // pixels (1 280 B).  MFMA pair j = (K step j / NT, tap j % NT) uses patch fragment j for both co halves; the patch fragment of
      // pair j + RINGW is requested right after pair j is issued, the two dY fragments of the next K step while the current one runs
      // (written as "read; mfma" the compiler put s_waitcnt lgkmcnt(0) in front of every MFMA).
#pragma unroll 1
      for (int hh = 0; hh < 2; ++hh) {
        const unsigned sbx = (unsigned)((i % NSTAGE) * STAGE * 2) + (unsigned)(hh * 4 * 2560), sby = (unsigned)((i % NSTAGE) * STAGE * 2) + (unsigned)(hh * 4 * 2048);
        constexpr int RINGW = 4, NPAIR = (TH / 4) * NT;
        f16x8 bq[RINGW], aq[2][2];
#define HD_WG8_B(J) tr_pair(lb + sbx + xa[(J) % NT][0] + ((J) / NT) * 2560, lb + sbx + xa[(J) % NT][1] + ((J) / NT) * 2560)
#define HD_WG8_A(S, COH) tr_pair(lb + sby + ya[COH][0] + (S) * 2048, lb + sby + ya[COH][1] + (S) * 2048)
        aq[0][0] = HD_WG8_A(0, 0);
        aq[0][1] = HD_WG8_A(0, 1);
#pragma unroll
        for (int j = 0; j < RINGW; ++j) bq[j] = HD_WG8_B(j);
#pragma unroll
        for (int j = 0; j < NPAIR; ++j) {
          const int s_ = j / NT, tp9 = j % NT;
          if (tp9 == 0 && s_ + 1 < TH / 4) {
            aq[(s_ + 1) & 1][0] = HD_WG8_A(s_ + 1, 0);
            aq[(s_ + 1) & 1][1] = HD_WG8_A(s_ + 1, 1);
          }
          __builtin_amdgcn_sched_barrier(0);
          acc[tp9][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[s_ & 1][0], bq[j % RINGW], acc[tp9][0], 0, 0, 0);
          acc[tp9][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[s_ & 1][1], bq[j % RINGW], acc[tp9][1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (j + RINGW < NPAIR) bq[j % RINGW] = HD_WG8_B(j + RINGW);
        }
#undef HD_WG8_A
#undef HD_WG8_B
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();               // this stage is free; the next tile has landed (the producers' wait)
    }
    __builtin_amdgcn_s_setprio(0);

    // ---- this consumer's sums ARE the block's result for its (ci half, taps): one coalesced write
    if (p.dw) {
      // hd_wgrad_args.dw_oihw (one pixel split): this block's sums ARE the gradient of its 64 x 64 x 9 weights.  A lane holds NT
      // consecutive taps of its (co, ci) pairs -- consecutive floats of the OIHW tensor, the 32 lanes of a half-wave 32 consecutive ci.
      // scale * sum as hd_wgrad_reduce forms it for one split (0.f + sum: the reduction's accumulator start, -0 -> +0; same bits).
#pragma unroll
      for (int coh = 0; coh < 2; ++coh)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co0 + coh * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int ci = ci0 + wci * 32 + (lane & 31);
          float* o = p.dw + ((size_t)co * p.Cin + ci) * 9 + T0;
#pragma unroll
          for (int t = 0; t < NT; ++t) o[t] = (0.f + acc[t][coh][r]) * p.dw_scale;
        }
    } else {
      float* out = p.slab + (size_t)bx * p.Cout * p.Ktot;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int coh = 0; coh < 2; ++coh)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = co0 + coh * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int ci = ci0 + wci * 32 + (lane & 31);
            out[(size_t)co * p.Ktot + (T0 + t) * p.Cin + ci] = acc[t][coh][r];
          }
    }
  };
  if (wtap == 0) run(std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{});
  else run(std::integral_constant<int, 5>{}, std::integral_constant<int, 4>{});
}


// host side: kernel parameters and grid of a launch (grid = (nsplit, ci chunks x co chunks))
inline void fill_params(const hd_wgrad_args* a, Wg8P& p, int* gx, int* gy) {
  p.x = (const f16*)a->x; p.x2 = (const f16*)a->x2; p.dy = (const f16*)a->dy; p.slab = a->slab;
  p.dw = a->nsplit == 1 ? a->dw_oihw : nullptr; p.dw_scale = a->dw_scale;
  p.N = a->N; p.Hsrc = a->Hsrc; p.Wsrc = a->Wsrc; p.H = a->Hin; p.W = a->Win; p.C1 = a->C1; p.C2 = a->C2;
  p.Cin = a->C1 + a->C2; p.Cout = a->Cout; p.Ktot = 9 * p.Cin;
  p.tiles_x = hd_cdiv(p.W, TW); p.tiles_y = hd_cdiv(p.H, TH);
  p.ntiles = p.N * p.tiles_x * p.tiles_y;
  p.per_split = hd_cdiv(p.ntiles, a->nsplit);
  p.dual = a->x2 != nullptr;
  p.xbytes = (unsigned)((int64_t)a->N * a->Hsrc * a->Wsrc * a->C1 * 2);
  p.x2bytes = a->x2 ? (unsigned)((int64_t)a->N * a->Hin * a->Win * a->C2 * 2) : 0u;
  p.dybytes = (unsigned)((int64_t)a->N * a->Ho * a->Wo * a->Cout * 2);
  *gx = a->nsplit;
  *gy = (p.Cin / 64) * (p.Cout / 64);
}

}  // namespace hd_wg8
