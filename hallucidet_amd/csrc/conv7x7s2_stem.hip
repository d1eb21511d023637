// 7x7 / stride 2 / pad 3 convolution, 8 (3 real) -> 64 channels, with the WEIGHTS RESIDENT IN REGISTERS (gfx950): the ResNet stem of
// the hallucination network (src/segmentation_models/encoders/resnet.py:47-52 via torchvision ResNet.conv1 [EXT]; 8 x 512 x 640 with
// BatchNorm partial sums) and of the frozen detector (24 x 300 x 300, folded FrozenBN bias + ReLU).  In the implicit-GEMM family these two
// launches gather a 392-deep K from 16-byte pixels tap by tap (per-lane tap recomputation, 72 and 55 us against a 16 us HBM floor).  Here,
// as in conv3x3_c64.hip:
//   * 4 waves per block, one wave per SIMD, one persistent block per CU walking a contiguous run of 8 x 16-pixel output tiles;
//   * the 64 x 392 weight matrix is the MFMA A operand and lives in the accumulator file: 2 x 25 fragments = 200 AGPRs per lane, loaded
//     once per block straight from memory (a lane's 8 K values are one tap's 8 channels: 16 contiguous bytes of a weight row);
//   * the (2*8 + 5) x (2*16 + 5) input patch of the NEXT tile (777 pixels of 16 bytes: 13 one-KiB DMA pieces) arrives by LDS-DMA during
//     the K loop, hardware zero-fill for the padding; a K step of v_mfma_f32_32x32x16_f16 is two taps, its B fragment one 16-byte read
//     per lane at (2 oy + kh, 2 ox + kw);
//   * epilogue in registers as in the 64-channel kernel (bias / ReLU / residual / mask, 16-byte stores), BatchNorm partial sums in
//     registers, folded once per block.
// K order: tap-major, 8 channels per tap (tap 49 is a zero row) -- the order of the weight layout [64][7*7*8].
#include "hd_common.h"
#include "conv_params.h"
#include <stdlib.h>

namespace {

constexpr int TH = 8, TW = 16;
constexpr int PH = 2 * TH + 5, PW = 2 * TW + 5;      // 21 x 37 input pixels
constexpr int NPIX = PH * PW;                        // 777
constexpr int NPIECE = (NPIX + 63) / 64;             // 13
constexpr int PPW = (NPIECE + 3) / 4;                // pieces per wave (4; three waves issue 3)
constexpr int STAGE_BYTES = 4 * PPW * 1024;          // 16 KiB
constexpr int KSTEPS = 25, NTAP = 49, KROW = NTAP * 8;
constexpr int RED_OFF = 2 * STAGE_BYTES;             // statistics transpose [4 waves][64][65] floats
constexpr int BIAS_OFF = RED_OFF + 4 * 64 * 65 * 4;  // the bias vector (EPI & 2): read per tile from LDS, not from memory
constexpr int LDS_BYTES = BIAS_OFF + 64 * 4;
constexpr unsigned OOBB = 0x80000000u;

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds_dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds_dst, 16, voff, 0, 0, 0);
}

#define HD_STEM_MFMA0(ACC, WF, BF) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(ACC) : "a"(WF), "v"(BF))
#define HD_STEM_MFMA(ACC, WF, BF) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "a"(WF), "v"(BF))
#define HD_STEM_DRAIN(A0, A1) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(A0), "+v"(A1))

constexpr int RING = 6;

// byte offset of tap t inside the patch, relative to the lane's pixel (2 oy, 2 ox); the zero tap re-reads the last real one
__host__ __device__ constexpr int tap_off(int t) { return t < NTAP ? ((t / 7) * PW + (t % 7)) * 16 : ((6 * PW) + 6) * 16; }

// EPI: 1 residual, 2 bias, 4 ReLU mask, 8 ReLU; STATS: BatchNorm partial sums (sum y, sum y^2 of the fp16-rounded output, one row per block)
template <bool STATS, int EPI>
__global__ __launch_bounds__(256) void conv7x7s2_stem_kernel(ConvP p, int tiles_total) {
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

  // ---- patch fill tables: piece k of this wave is global piece k * 4 + wave; its lane fetches patch pixel (k * 4 + wave) * 64 + lane
  int rel[PPW], pyx[PPW];
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    const int pp = (k * 4 + wave) * 64 + lane;
    const int py = pp / PW, px = pp - py * PW;
    rel[k] = ((py - 3) * W + (px - 3)) * 16;
    pyx[k] = pp < NPIX ? (py | (px << 8)) : 0x4000;                // bit 14: not a patch pixel
  }
  auto tile_pos = [&](int t, int& n, int& ty, int& tx) {
    const int r1 = t / tiles_x;
    tx = t - r1 * tiles_x;
    n = r1 / tiles_y;
    ty = r1 - n * tiles_y;
  };
  auto issue_patch = [&](int n, int ty, int tx, int stage, int k) {
    if (k * 4 + wave >= NPIECE) return;                            // uniform per wave
    const int base = ((n * H + ty * (2 * TH)) * W + tx * (2 * TW)) * 16;
    const int py = pyx[k] & 0xff, px = (pyx[k] >> 8) & 0x3f;
    const int iy = ty * (2 * TH) - 3 + py, ix = tx * (2 * TW) - 3 + px;
    const bool ok = !(pyx[k] & 0x4000) && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    dma16(rx, lds + stage * STAGE_BYTES + (k * 4 + wave) * 1024, ok ? (unsigned)(base + rel[k]) : OOBB);
  };

  int cn = 0, cty = 0, ctx = 0;
  if (t_begin < t_end) {
    tile_pos(t_begin, cn, cty, ctx);
#pragma unroll
    for (int k = 0; k < PPW; ++k) issue_patch(cn, cty, ctx, 0, k);
  }
  // ---- weights: row 32 b + pl, K step s, half h = tap 2 s + h, all 8 channels (16 contiguous bytes); the 50th tap is a zero row
  f16x8 wr[2][KSTEPS];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int t = 2 * s + h;
      f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (t < NTAP) v = *reinterpret_cast<const f16x8*>(p.w + (size_t)(32 * b + pl) * KROW + t * 8);
      wr[b][s] = v;
    }

  // ---- B fragments: this lane's output pixel (two tile rows per wave) -> patch pixel (2 oy, 2 ox)
  const int y0l = 2 * wave + ((lane >> 4) & 1), x0l = lane & 15;
  const int bbase = ((2 * y0l) * PW + 2 * x0l) * 16;

  float s1[2][16], s2[2][16];
  if (STATS) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) s1[b][r] = s2[b][r] = 0.f;
  }
  const f16* __restrict__ resp = p.res;
  const f16* __restrict__ maskp = p.mask;
  f16* __restrict__ yp = reinterpret_cast<f16*>(p.y);
  if (EPI & 2) {
    if (tid < 64) reinterpret_cast<float*>(lds + BIAS_OFF)[tid] = p.bias[tid];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (ordered against the readers by the tile loop's first barrier)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  for (int t = t_begin; t < t_end; ++t) {
    const int stage = (t - t_begin) & 1;
    __builtin_amdgcn_s_barrier();          // every wave has retired its pieces of tile t and is done reading the other stage
    const bool more = t + 1 < t_end;
    int nn = 0, nty = 0, ntx = 0;
    if (more) tile_pos(t + 1, nn, nty, ntx);
    const char* sb = lds + stage * STAGE_BYTES + bbase;
    f32x16 acc0, acc1;
    f16x8 bf[RING];
#define HD_STEM_B(S) (*reinterpret_cast<const f16x8*>(sb + (h ? tap_off(2 * (S) + 1) : tap_off(2 * (S)))))
#pragma unroll
    for (int s = 0; s < RING; ++s) bf[s] = HD_STEM_B(s);
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const f16x8 cur = bf[s % RING];
      if (s == 0) {
        HD_STEM_MFMA0(acc0, wr[0][s], cur);
        HD_STEM_MFMA0(acc1, wr[1][s], cur);
      } else {
        HD_STEM_MFMA(acc0, wr[0][s], cur);
        HD_STEM_MFMA(acc1, wr[1][s], cur);
      }
      if (s + RING < KSTEPS) bf[s % RING] = HD_STEM_B(s + RING);
      if (s % 6 == 1 && more) issue_patch(nn, nty, ntx, stage ^ 1, s / 6);      // s = 1, 7, 13, 19: the four pieces of the next patch
    }
#undef HD_STEM_B
    HD_STEM_DRAIN(acc0, acc1);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue in registers: acc{b}[4g + i] = channel 32b + 8g + 4h + i of this lane's pixel
    const int oy = cty * TH + y0l, ox = ctx * TW + x0l;
    const bool okp = oy < p.Ho && ox < p.Wo;
    const unsigned eoff = (unsigned)(((cn * p.Ho + oy) * p.Wo + ox) * 64);
    f16x4 rv[8], mv[8];
    f32x4 bv[8];
    if (EPI & 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) rv[j] = okp ? *reinterpret_cast<const f16x4*>(resp + eoff + 8 * j + 4 * h) : (f16x4){0, 0, 0, 0};
    }
    if (EPI & 4) {
#pragma unroll
      for (int j = 0; j < 8; ++j) mv[j] = okp ? *reinterpret_cast<const f16x4*>(maskp + eoff + 8 * j + 4 * h) : (f16x4){0, 0, 0, 0};
    }
    if (EPI & 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) bv[j] = *reinterpret_cast<const f32x4*>(lds + BIAS_OFF + (8 * j + 4 * h) * 4);
    }
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int gp = 0; gp < 4; gp += 2) {
        unsigned pk[2][2];
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int g = gp + gg, j = 4 * b + g;
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = b ? acc1[4 * g + i] : acc0[4 * g + i];
          if (EPI & 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += (float)rv[j][i];
          }
          if (EPI & 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += bv[j][i];
          }
          if (EPI & 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ((float)mv[j][i] > 0.f) ? v[i] : 0.f;
          }
          if (STATS) {
            const float keep = okp ? 1.f : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float vr = (float)(f16)v[i] * keep;
              s1[b][4 * g + i] += vr;
              s2[b][4 * g + i] += vr * vr;
            }
          }
          if (EPI & 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
          }
          const f16x2 o01 = {(f16)v[0], (f16)v[1]}, o23 = {(f16)v[2], (f16)v[3]};
          pk[gg][0] = __builtin_bit_cast(unsigned, o01);
          pk[gg][1] = __builtin_bit_cast(unsigned, o23);
        }
        const auto q0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
        const auto q1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
        const u32x4 o = {q0[0], q1[0], q0[1], q1[1]};
        if (okp) *reinterpret_cast<u32x4*>(yp + eoff + 32 * b + 8 * (gp + h)) = o;
      }
    cn = nn; cty = nty; ctx = ntx;
    __builtin_amdgcn_sched_barrier(0);
  }

  if (STATS) {
    // one partial row per block: transpose the 64 per-lane sums of a wave through LDS (pitch 65 floats), lane j adds value j over the
    // 32 pixel lanes of each half, then 128 threads add the four waves -- fixed order throughout
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds + RED_OFF) + wave * (64 * 65);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        red[(b * 16 + r) * 65 + lane] = s1[b][r];
        red[(32 + b * 16 + r) * 65 + lane] = s2[b][r];
      }
    __syncthreads();
    float a0 = 0.f, a1 = 0.f;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
      a0 += red[lane * 65 + i];
      a1 += red[lane * 65 + 32 + i];
    }
    float* red2 = reinterpret_cast<float*>(lds);       // [wave][value][half]
    red2[(wave * 64 + lane) * 2 + 0] = a0;
    red2[(wave * 64 + lane) * 2 + 1] = a1;
    __syncthreads();
    if (tid < 128) {
      const int v = tid >> 1, hh = tid & 1;
      float s = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) s += red2[(w4 * 64 + v) * 2 + hh];
      const int which = v >> 5, b = (v >> 4) & 1, r = v & 15;
      const int c = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * hh;
      p.stats[((size_t)blockIdx.x * 2 + which) * 64 + c] = s;
    }
  }
}

}  // namespace

// 7x7 / s2 / p3, one 8-channel source, 64 output channels, NHWC f16 out.  (Nothing here may depend on p.stats.)
bool hd_conv_stem_eligible(const ConvP& p) {
  if (p.KH != 7 || p.KW != 7 || p.stride != 2 || p.pad != 3 || p.in_dil != 1 || p.up1) return false;
  if (p.C1 != 8 || p.C2 != 0 || p.x2 || p.Cout != 64 || p.out_mode != HD_OUT_NHWC_F16 || p.in_scale || p.bs_y) return false;
  if (p.Hsrc != p.Hin || p.Wsrc != p.Win || p.Ho != (p.Hin + 6 - 7) / 2 + 1 || p.Wo != (p.Win + 6 - 7) / 2 + 1) return false;
  if (p.act != HD_ACT_NONE && p.act != HD_ACT_RELU) return false;
  if (p.xbytes & 0xC0000000u) return false;
  if ((int64_t)p.N * p.Ho * p.Wo * 64 >= (int64_t)1 << 31) return false;
  return true;
}

static int stem_tiles(const ConvP& p) { return p.N * hd_cdiv(p.Ho, TH) * hd_cdiv(p.Wo, TW); }

int hd_conv_stem_rows(const ConvP& p) {
  const int t = stem_tiles(p);
  return t < 256 ? t : 256;
}

void hd_conv_launch_stem(ConvP& p, hipStream_t s) {
  const int tiles = stem_tiles(p);
  dim3 grid(hd_conv_stem_rows(p));
  const int epi = (p.res ? 1 : 0) | (p.bias ? 2 : 0) | (p.mask ? 4 : 0) | (p.act == HD_ACT_RELU ? 8 : 0);
#define LAUNCH(E) case E: if (p.stats) hipLaunchKernelGGL((conv7x7s2_stem_kernel<true, E>), grid, dim3(256), 0, s, p, tiles); \
                          else hipLaunchKernelGGL((conv7x7s2_stem_kernel<false, E>), grid, dim3(256), 0, s, p, tiles); break
  switch (epi) {
    LAUNCH(0); LAUNCH(1); LAUNCH(2); LAUNCH(3); LAUNCH(4); LAUNCH(5); LAUNCH(6); LAUNCH(7);
    LAUNCH(8); LAUNCH(9); LAUNCH(10); LAUNCH(11); LAUNCH(12); LAUNCH(13); LAUNCH(14); LAUNCH(15);
  }
#undef LAUNCH
}
