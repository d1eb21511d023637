// 3x3 / stride 1 / pad 1 convolution, 32 -> 128 channels, with the WEIGHTS RESIDENT IN REGISTERS (gfx950): the data gradient of decoder
// block 3's first convolution of the hallucination network (src/segmentation_models/decoders/unet/decoder.py:38-46; its forward takes
// 64 upsampled + 64 skip channels to 32, so the gradient w.r.t. the concatenated input is a 32 -> 128 channel convolution at 256 x 320:
// 48 GFLOP, 111 us in the 4-wave implicit-GEMM family with a 32-wide N tile).  The 128 x 288 weight matrix is 72 KiB, exactly the size
// conv3x3_c64.hip keeps in registers, so the same design applies:
//   * 4 waves per block, one wave per SIMD, one persistent block per CU over 8 x 16-pixel tiles; a wave owns 32 pixels;
//   * 4 cout blocks x 18 K steps = 72 A fragments per lane (64 in AGPRs, 8 in VGPRs), loaded once per block straight from memory;
//   * the 10 x 18 patch of 64-byte pixels (12 one-KiB DMA pieces) of the NEXT tile arrives during the K loop; chunk slot ^= (x >> 2) & 3 on
//     the source side keeps the sixteen lanes of a fragment read on distinct banks;
//   * the 128 output channels are produced in TWO passes of 64 (two accumulators per pass, as in the 64-channel kernel; the pixel
//     fragments are read again for the second pass -- the LDS port has the room), each pass followed by its register epilogue.
// K order: tap-major, 16 channels at a time -- the order the implicit-GEMM kernels accumulate in.
#include "hd_common.h"
#include "conv_params.h"
#include <stdlib.h>

namespace {

constexpr int TH = 8, TW = 16, PH = TH + 2, PW = TW + 2;
constexpr int NPIX = PH * PW;                        // 180 patch pixels of 64 bytes
constexpr int NPIECE = (NPIX * 4 + 63) / 64;         // 12
constexpr int PPW = (NPIECE + 3) / 4;                // 3 pieces per wave
constexpr int STAGE_BYTES = 4 * PPW * 1024;          // 12 KiB
constexpr int KSTEPS = 18, KROW = 288;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;
constexpr unsigned OOBB = 0x80000000u;

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds_dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds_dst, 16, voff, 0, 0, 0);
}

#define HD_C32_MFMA0_A(ACC, WF, BF) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(ACC) : "a"(WF), "v"(BF))
#define HD_C32_MFMA_A(ACC, WF, BF) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "a"(WF), "v"(BF))
#define HD_C32_MFMA_V(ACC, WF, BF) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(WF), "v"(BF))
#define HD_C32_DRAIN(A0, A1) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(A0), "+v"(A1))

constexpr int RING = 6;
constexpr int NA = 16;         // K steps whose four weight fragments live in AGPRs (4 x 16 x 4 = 256 registers)

__global__ __launch_bounds__(256) void conv3x3_c32to128_kernel(ConvP p, int tiles_total) {
  __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, pl = lane & 31;

  const int G = gridDim.x;
  int L;
  {
    const int b = blockIdx.x, xcd = b & 7, qq = G >> 3, rr = G & 7;
    L = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (b >> 3);
  }
  const int t_begin = (int)((long long)L * tiles_total / G), t_end = (int)((long long)(L + 1) * tiles_total / G);
  const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
  const int H = p.Hin, W = p.Win;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.x), 0, p.xbytes, 0x00020000);

  // ---- patch fill tables: unit u = (k*4 + wave)*64 + lane is 16-byte slot (u & 3) of patch pixel u >> 2
  int rel[PPW], pyx[PPW];
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    const int u = (k * 4 + wave) * 64 + lane;
    const int pp = u >> 2, slot = u & 3;
    const int py = (pp * 3641) >> 16, px = pp - py * PW;          // pp / 18 (exact for pp < 2 000)
    const int cg = slot ^ ((px >> 2) & 3);
    rel[k] = ((py - 1) * W + (px - 1)) * 64 + cg * 16;
    pyx[k] = pp < NPIX ? (py | (px << 8)) : 0x4000;                // bit 14: not a patch pixel
  }
  auto tile_pos = [&](int t, int& n, int& ty, int& tx) {
    const int r1 = t / tiles_x;
    tx = t - r1 * tiles_x;
    n = r1 / tiles_y;
    ty = r1 - n * tiles_y;
  };
  auto issue_patch = [&](int n, int ty, int tx, int stage, int k) {
    const int base = ((n * H + ty * TH) * W + tx * TW) * 64;
    const int py = pyx[k] & 0xff, px = (pyx[k] >> 8) & 0x3f;
    const int iy = ty * TH - 1 + py, ix = tx * TW - 1 + px;
    const bool ok = !(pyx[k] & 0x4000) && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    dma16(rx, lds + stage * STAGE_BYTES + (k * 4 + wave) * 1024, ok ? (unsigned)(base + rel[k]) : OOBB);
  };

  int cn = 0, cty = 0, ctx = 0;
  if (t_begin < t_end) {
    tile_pos(t_begin, cn, cty, ctx);
#pragma unroll
    for (int k = 0; k < PPW; ++k) issue_patch(cn, cty, ctx, 0, k);
  }
  // ---- weights: cout block b (0..3), row 32 b + pl, K step s, half h: K values 16 s + 8 h .. + 7 = 16 contiguous bytes of the row
  f16x8 wr[4][KSTEPS];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
    for (int b = 0; b < 4; ++b) wr[b][s] = *reinterpret_cast<const f16x8*>(p.w + (size_t)(32 * b + pl) * KROW + s * 16 + h * 8);

  // ---- B fragment addresses: this lane's pixel (two tile rows per wave) at tap (kh, kw), 16-channel block c, half h
  const int y0l = 2 * wave + ((lane >> 4) & 1), x0l = lane & 15;
  int ab[3][2];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int xk = x0l + kw;
#pragma unroll
    for (int c = 0; c < 2; ++c) ab[kw][c] = (y0l * PW + xk) * 64 + (((2 * c + h) ^ ((xk >> 2) & 3)) & 3) * 16;
  }
  f16* __restrict__ yp = reinterpret_cast<f16*>(p.y);
  // out_pool2 = 64 (round 5): this launch is the data gradient of decoder block 3's first convolution over cat([nearest_2x(a), skip]):
  // channels 0..63 belong to the upsampled half -- their 2 x 2 SUM is the gradient of `a` and goes, pooled in fp32 in the epilogue, to
  // y [N, Ho/2, Wo/2, 64]; channels 64..127 are the skip's gradient and go straight to y2 [N, Ho, Wo, 64].  The 168 MB concatenated
  // gradient is never written and hd_concat_up_bwd (273 MB of traffic) is not launched.
  const bool pool = p.pool2 != 0;
  f16* __restrict__ y2p = reinterpret_cast<f16*>(p.y2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  for (int t = t_begin; t < t_end; ++t) {
    const int stage = (t - t_begin) & 1;
    __builtin_amdgcn_s_barrier();          // every wave has retired its pieces of tile t and is done reading the other stage
    const bool more = t + 1 < t_end;
    int nn = 0, nty = 0, ntx = 0;
    if (more) tile_pos(t + 1, nn, nty, ntx);
    const char* sb = lds + stage * STAGE_BYTES;
    const int oy = cty * TH + y0l, ox = ctx * TW + x0l;
    const bool okp = oy < p.Ho && ox < p.Wo;
    const unsigned eoff = (unsigned)(((cn * p.Ho + oy) * p.Wo + ox) * 128);
    // K step s: tap s >> 1 = (kh, kw), channel block s & 1
#define HD_C32_B(S) (*reinterpret_cast<const f16x8*>(sb + ab[((S) >> 1) % 3][(S) & 1] + ((S) / 6) * (PW * 64)))
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      f32x16 acc0, acc1;
      f16x8 bf[RING];
#pragma unroll
      for (int s = 0; s < RING; ++s) bf[s] = HD_C32_B(s);
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        const f16x8 cur = bf[s % RING];
        if (s == 0) {
          HD_C32_MFMA0_A(acc0, wr[2 * pass][s], cur);
          HD_C32_MFMA0_A(acc1, wr[2 * pass + 1][s], cur);
        } else if (s < NA) {
          HD_C32_MFMA_A(acc0, wr[2 * pass][s], cur);
          HD_C32_MFMA_A(acc1, wr[2 * pass + 1][s], cur);
        } else {
          HD_C32_MFMA_V(acc0, wr[2 * pass][s], cur);
          HD_C32_MFMA_V(acc1, wr[2 * pass + 1][s], cur);
        }
        if (s + RING < KSTEPS) bf[s % RING] = HD_C32_B(s + RING);
        // the next tile's patch: three pieces per wave, spread over the first pass
        if (pass == 0 && s % 6 == 1 && more) issue_patch(nn, nty, ntx, stage ^ 1, s / 6);
      }
      HD_C32_DRAIN(acc0, acc1);
      __builtin_amdgcn_sched_barrier(0);
      // ---- epilogue in registers: acc{b}[4g + i] = channel 64 pass + 32 b + 8 g + 4 h + i of this lane's pixel
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int gp = 0; gp < 4; gp += 2) {
          unsigned pk[2][2];
#pragma unroll
          for (int gg = 0; gg < 2; ++gg) {
            const int g = gp + gg;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = b ? acc1[4 * g + i] : acc0[4 * g + i];
            if (pool && pass == 0) {          // this lane's pixel + its column neighbour (lane ^ 1) + the row below / above (lane ^ 16)
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                v[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[i]), 0xB1, 0xF, 0xF, true));
                v[i] += __shfl_xor(v[i], 16);
              }
            }
            const f16x2 o01 = {(f16)v[0], (f16)v[1]}, o23 = {(f16)v[2], (f16)v[3]};
            pk[gg][0] = __builtin_bit_cast(unsigned, o01);
            pk[gg][1] = __builtin_bit_cast(unsigned, o23);
          }
          const auto q0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
          const auto q1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
          const u32x4 o = {q0[0], q1[0], q0[1], q1[1]};
          if (!pool) {
            if (okp) *reinterpret_cast<u32x4*>(yp + eoff + 64 * pass + 32 * b + 8 * (gp + h)) = o;
          } else if (pass == 1) {
            if (okp) *reinterpret_cast<u32x4*>(y2p + (eoff >> 1) + 32 * b + 8 * (gp + h)) = o;        // [pixel][64]
          } else if (okp && !(lane & 17)) {   // even column, even row: the 2 x 2 block's owner (Ho, Wo even: the block is complete)
            *reinterpret_cast<u32x4*>(yp + (unsigned)(((cn * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1)) * 64) + 32 * b + 8 * (gp + h)) = o;
          }
        }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef HD_C32_B
    // the next tile's pieces were issued during the first pass: nothing younger than them (but this tile's stores) is outstanding
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    cn = nn; cty = nty; ctx = ntx;
  }
}

}  // namespace

// 3x3 / s1 / p1, one 32-channel source, 128 output channels, plain NHWC f16 output (a data gradient: no bias / residual / mask / act / sums)
bool hd_conv_c32to128_eligible(const ConvP& p) {
  if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.in_dil != 1 || p.up1) return false;
  if (p.C1 != 32 || p.C2 != 0 || p.x2 || p.Cout != 128 || p.out_mode != HD_OUT_NHWC_F16) return false;
  if (p.Ho != p.Hin || p.Wo != p.Win || p.Hsrc != p.Hin || p.Wsrc != p.Win || p.in_scale || p.bs_y) return false;
  if (p.act != HD_ACT_NONE || p.bias || p.res || p.mask) return false;
  if (p.xbytes & 0xC0000000u) return false;
  if ((int64_t)p.N * p.Ho * p.Wo * 128 >= (int64_t)1 << 31) return false;
  if (p.pool2 && !(p.pool2 == 64 && p.y2 && (p.Ho % 2) == 0 && (p.Wo % 2) == 0)) return false;
  return true;
}

void hd_conv_launch_c32to128(ConvP& p, hipStream_t s) {
  const int tiles = p.N * hd_cdiv(p.Ho, TH) * hd_cdiv(p.Wo, TW);
  hipLaunchKernelGGL(conv3x3_c32to128_kernel, dim3(tiles < 256 ? tiles : 256), dim3(256), 0, s, p, tiles);
}
