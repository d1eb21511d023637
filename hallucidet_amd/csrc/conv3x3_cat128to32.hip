// 3x3 / stride 1 / pad 1 convolution over cat([nearest_2x(a), skip]) with 64 + 64 input and 32 output channels, WEIGHTS RESIDENT IN
// REGISTERS (gfx950): decoder block 3's first convolution of the hallucination network at 256 x 320
// (src/segmentation_models/decoders/unet/decoder.py:37-46: F.interpolate(x, 2) -> torch.cat -> Conv2dReLU; 48 GFLOP, 117 us in the 4-wave
// implicit-GEMM family whose 32-wide N tile re-gathers the 1 152-deep K for every 128 pixels).  The 32 x 1 152 weight matrix is 72 KiB --
// the size conv3x3_c64.hip keeps in registers -- so the same design applies, with the upsample + concat folded into the patch DMA:
//   * 4 waves per block, one wave per SIMD, one persistent block per CU over 8 x 16-pixel tiles; a wave owns 32 pixels x all 32 channels;
//   * 72 A fragments per lane (one cout block x 72 K steps; 64 in AGPRs, 8 in VGPRs), loaded once per block straight from memory;
//   * TWO patches of 128-byte pixels per tile, each laid out and swizzled exactly as the 64-channel kernel's: region A holds
//     a[(y >> 1), (x >> 1)] (the nearest-2x upsample is a halved source coordinate), region B holds skip[y, x]; an LDS-DMA instruction
//     takes one buffer resource, so the two sources never share a 1-KiB piece; 2 x 23 pieces per tile arrive during the previous tile's
//     K loop, hardware zero-fill for the padding;
//   * a K step is (tap, 16-channel block): blocks 0..3 read region A, 4..7 region B; even and odd K steps accumulate into two independent
//     chains that are added in the epilogue; register epilogue (16-byte stores), BatchNorm partial sums in registers, folded once per block.
// K order: tap-major, the 64 upsampled then the 64 skip channels, 16 at a time -- the order of the implicit-GEMM kernels.
#include "hd_common.h"
#include "conv_params.h"
#include <stdlib.h>

namespace {

constexpr int TH = 8, TW = 16, PH = TH + 2, PW = TW + 2;
constexpr int NPIX = PH * PW;                        // 180 patch pixels of 128 bytes per region
constexpr int PIECES = 6;                            // 1-KiB DMA pieces per wave per region (4 x 6 = 24 >= 22.5)
constexpr int REGION_BYTES = 4 * PIECES * 1024;      // 24 KiB
constexpr int STAGE_BYTES = 2 * REGION_BYTES;        // 48 KiB
constexpr int KSTEPS = 72, KROW = 1152;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;           // 96 KiB
constexpr unsigned OOBB = 0x80000000u;

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds_dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds_dst, 16, voff, 0, 0, 0);
}

#define HD_CAT_MFMA0_A(ACC, WF, BF) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(ACC) : "a"(WF), "v"(BF))
#define HD_CAT_MFMA_A(ACC, WF, BF) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "a"(WF), "v"(BF))
#define HD_CAT_MFMA_V(ACC, WF, BF) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(WF), "v"(BF))
#define HD_CAT_DRAIN(A0, A1) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(A0), "+v"(A1))

constexpr int RING = 6;
constexpr int NA = 64;         // K steps whose weight fragment lives in AGPRs (64 x 4 = 256 registers)

template <bool STATS>
__global__ __launch_bounds__(256) void conv3x3_cat128to32_kernel(ConvP p, int tiles_total) {
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
  const int H = p.Hin, W = p.Win, Hs = p.Hsrc, Ws = p.Wsrc;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.x), 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.x2), 0, p.x2bytes, 0x00020000);

  // ---- patch fill tables (both regions): unit u = (k*4 + wave)*64 + lane is 16-byte slot (u & 7) of patch pixel u >> 3
  int relu_[PIECES], rels_[PIECES], pyx[PIECES];
#pragma unroll
  for (int k = 0; k < PIECES; ++k) {
    const int u = (k * 4 + wave) * 64 + lane;
    const int pp = u >> 3, slot = u & 7;
    const int py = (pp * 3641) >> 16, px = pp - py * PW;          // pp / 18 (exact for pp < 2 000)
    const int cg = slot ^ ((px >> 1) & 7);
    relu_[k] = (((py - 1) >> 1) * Ws + ((px - 1) >> 1)) * 128 + cg * 16;     // upsampled source: halved coordinates (floor, also for -1)
    rels_[k] = ((py - 1) * W + (px - 1)) * 128 + cg * 16;
    pyx[k] = pp < NPIX ? (py | (px << 8)) : 0x4000;                // bit 14: not a patch pixel
  }
  auto tile_pos = [&](int t, int& n, int& ty, int& tx) {
    const int r1 = t / tiles_x;
    tx = t - r1 * tiles_x;
    n = r1 / tiles_y;
    ty = r1 - n * tiles_y;
  };
  // piece k (0 .. 2*PIECES-1) of the next patch: k < PIECES region A (upsampled source), else region B (skip source)
  auto issue_patch = [&](int n, int ty, int tx, int stage, int k) {
    const int kk = k < PIECES ? k : k - PIECES;
    const int py = pyx[kk] & 0xff, px = (pyx[kk] >> 8) & 0x3f;
    const int iy = ty * TH - 1 + py, ix = tx * TW - 1 + px;
    const bool ok = !(pyx[kk] & 0x4000) && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    if (k < PIECES) {
      const int base = ((n * Hs + ty * (TH / 2)) * Ws + tx * (TW / 2)) * 128;
      dma16(rx, lds + stage * STAGE_BYTES + (kk * 4 + wave) * 1024, ok ? (unsigned)(base + relu_[kk]) : OOBB);
    } else {
      const int base = ((n * H + ty * TH) * W + tx * TW) * 128;
      dma16(rx2, lds + stage * STAGE_BYTES + REGION_BYTES + (kk * 4 + wave) * 1024, ok ? (unsigned)(base + rels_[kk]) : OOBB);
    }
  };

  int cn = 0, cty = 0, ctx = 0;
  if (t_begin < t_end) {
    tile_pos(t_begin, cn, cty, ctx);
#pragma unroll
    for (int k = 0; k < 2 * PIECES; ++k) issue_patch(cn, cty, ctx, 0, k);
  }
  // ---- weights: row pl (cout), K step s, half h: K values 16 s + 8 h .. + 7 = 16 contiguous bytes of the row
  f16x8 wr[KSTEPS];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) wr[s] = *reinterpret_cast<const f16x8*>(p.w + (size_t)pl * KROW + s * 16 + h * 8);

  // ---- B fragment addresses: this lane's pixel (two tile rows per wave) at tap (kh, kw), 16-channel group c of a region, half h
  const int y0l = 2 * wave + ((lane >> 4) & 1), x0l = lane & 15;
  int ab[3][4];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int xk = x0l + kw;
#pragma unroll
    for (int c = 0; c < 4; ++c) ab[kw][c] = (y0l * PW + xk) * 128 + (((2 * c + h) ^ ((xk >> 1) & 7)) & 7) * 16;
  }

  float s1[16], s2[16];
  if (STATS) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s1[r] = s2[r] = 0.f;
  }
  f16* __restrict__ yp = reinterpret_cast<f16*>(p.y);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  for (int t = t_begin; t < t_end; ++t) {
    const int stage = (t - t_begin) & 1;
    __builtin_amdgcn_s_barrier();          // every wave has retired its pieces of tile t and is done reading the other stage
    const bool more = t + 1 < t_end;
    int nn = 0, nty = 0, ntx = 0;
    if (more) tile_pos(t + 1, nn, nty, ntx);
    const char* sb = lds + stage * STAGE_BYTES;
    f32x16 acc0, acc1;
    f16x8 bf[RING];
    // K step s: tap s >> 3 = (kh, kw); 16-channel block s & 7: region (s & 7) >> 2, group (s & 3)
#define HD_CAT_B(S) (*reinterpret_cast<const f16x8*>(sb + (((S) & 7) >> 2) * REGION_BYTES + ab[((S) >> 3) % 3][(S) & 3] + ((S) / 24) * (PW * 128)))
#pragma unroll
    for (int s = 0; s < RING; ++s) bf[s] = HD_CAT_B(s);
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const f16x8 cur = bf[s % RING];
      if (s == 0) HD_CAT_MFMA0_A(acc0, wr[s], cur);
      else if (s == 1) HD_CAT_MFMA0_A(acc1, wr[s], cur);
      else if (s < NA) {
        if (s & 1) HD_CAT_MFMA_A(acc1, wr[s], cur);
        else HD_CAT_MFMA_A(acc0, wr[s], cur);
      } else {
        if (s & 1) HD_CAT_MFMA_V(acc1, wr[s], cur);
        else HD_CAT_MFMA_V(acc0, wr[s], cur);
      }
      if (s + RING < KSTEPS) bf[s % RING] = HD_CAT_B(s + RING);
      // the next tile's two patches: one piece every six K steps (12 pieces over 72 steps)
      if (s % 6 == 1 && more) issue_patch(nn, nty, ntx, stage ^ 1, s / 6);
    }
#undef HD_CAT_B
    HD_CAT_DRAIN(acc0, acc1);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue in registers: (acc0 + acc1)[4g + i] = channel 8g + 4h + i of this lane's pixel
    const int oy = cty * TH + y0l, ox = ctx * TW + x0l;
    const bool okp = oy < p.Ho && ox < p.Wo;
    const unsigned eoff = (unsigned)(((cn * p.Ho + oy) * p.Wo + ox) * 32);
#pragma unroll
    for (int gp = 0; gp < 4; gp += 2) {
      unsigned pk[2][2];
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        const int g = gp + gg;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc0[4 * g + i] + acc1[4 * g + i];
        if (STATS) {
          const float keep = okp ? 1.f : 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float vr = (float)(f16)v[i] * keep;
            s1[4 * g + i] += vr;
            s2[4 * g + i] += vr * vr;
          }
        }
        const f16x2 o01 = {(f16)v[0], (f16)v[1]}, o23 = {(f16)v[2], (f16)v[3]};
        pk[gg][0] = __builtin_bit_cast(unsigned, o01);
        pk[gg][1] = __builtin_bit_cast(unsigned, o23);
      }
      const auto q0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
      const auto q1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
      const u32x4 o = {q0[0], q1[0], q0[1], q1[1]};
      if (okp) *reinterpret_cast<u32x4*>(yp + eoff + 8 * (gp + h)) = o;
    }
    cn = nn; cty = nty; ctx = ntx;
    __builtin_amdgcn_sched_barrier(0);
  }

  if (STATS) {
    // one partial row per block: transpose the 32 per-lane sums of a wave through LDS (pitch 65 floats), lane j < 32 adds value j over the
    // 32 pixel lanes of each half, then 64 threads add the four waves -- fixed order throughout
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds) + wave * (32 * 65);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      red[r * 65 + lane] = s1[r];
      red[(16 + r) * 65 + lane] = s2[r];
    }
    __syncthreads();
    float a0 = 0.f, a1 = 0.f;
    if (lane < 32) {
#pragma unroll 8
      for (int i = 0; i < 32; ++i) {
        a0 += red[lane * 65 + i];
        a1 += red[lane * 65 + 32 + i];
      }
    }
    __syncthreads();
    float* red2 = reinterpret_cast<float*>(lds) + 4 * 32 * 65;       // [wave][value][half]
    if (lane < 32) {
      red2[(wave * 32 + lane) * 2 + 0] = a0;
      red2[(wave * 32 + lane) * 2 + 1] = a1;
    }
    __syncthreads();
    if (tid < 64) {
      const int v = tid >> 1, hh = tid & 1;
      float s = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) s += red2[(w4 * 32 + v) * 2 + hh];
      const int which = v >> 4, r = v & 15;
      const int c = (r & 3) + 8 * (r >> 2) + 4 * hh;
      p.stats[((size_t)blockIdx.x * 2 + which) * 32 + c] = s;
    }
  }
}

}  // namespace

// 3x3 / s1 / p1 over cat([nearest_2x(x), x2]) with 64 + 64 input channels, 32 output channels, plain NHWC f16 output (+ BN partial sums).
// (Nothing here may depend on p.stats.)
bool hd_conv_cat128to32_eligible(const ConvP& p) {
  if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.in_dil != 1 || !p.up1 || !p.x2) return false;
  if (p.C1 != 64 || p.C2 != 64 || p.Cout != 32 || p.out_mode != HD_OUT_NHWC_F16) return false;
  if (p.Ho != p.Hin || p.Wo != p.Win || p.Hin != 2 * p.Hsrc || p.Win != 2 * p.Wsrc || p.in_scale || p.bs_y) return false;
  if (p.act != HD_ACT_NONE || p.bias || p.res || p.mask) return false;
  if ((p.xbytes | p.x2bytes) & 0xC0000000u) return false;
  return true;
}

static int cat_tiles(const ConvP& p) { return p.N * hd_cdiv(p.Ho, TH) * hd_cdiv(p.Wo, TW); }

int hd_conv_cat128to32_rows(const ConvP& p) {
  const int t = cat_tiles(p);
  return t < 256 ? t : 256;
}

void hd_conv_launch_cat128to32(ConvP& p, hipStream_t s) {
  const int tiles = cat_tiles(p);
  dim3 grid(hd_conv_cat128to32_rows(p));
  if (p.stats) hipLaunchKernelGGL((conv3x3_cat128to32_kernel<true>), grid, dim3(256), 0, s, p, tiles);
  else hipLaunchKernelGGL((conv3x3_cat128to32_kernel<false>), grid, dim3(256), 0, s, p, tiles);
}
