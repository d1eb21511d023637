// Weight gradient of the 3x3 / stride 1 / pad 1 convolutions with a THIN output (Cout <= 32): the U-Net decoder's 16/32-channel layers
// at 256x320 / 512x640 and the segmentation head (Cin in {16,32}), and decoder block 3's first conv (Cin = 64 upsampled + 64 skip
// channels -> 32, src/segmentation_models/decoders/unet/decoder.py:38-46: 209 us in the general kernel, the longest launch of the step).
//
// In the general wgrad_kernel these launches run at ~4x their HBM time: the im2col operand is refetched tap by tap in
// 16-byte pieces of 32-byte pixels, and K = 144 fills 1.1 of its two 128-column tiles.  Here a block walks 8x32-pixel
// tiles and stages, per tile, the (8+2)x(32+2) input patch and the 8x32 dY tile ONCE, in their natural [pixel][channel]
// layout (plain 16-byte copies).  The reduction index of dW[co][tap,ci] = sum_pix dY[pix][co] * X[pix+tap][ci] is the
// pixel, and a v_mfma_f32_16x16x32_f16 operand lane needs 8 consecutive reduction elements of one channel: the gfx950
// transposing LDS read (ds_read_b64_tr_b16, as in wgrad.hip) delivers exactly that from pixel-major rows, and a tap's
// shift is just a different first row -- no software transpose, no shifted copies.  (A first version transposed with
// 2-byte LDS stores into three kw-shifted channel-major copies: 24 stores per 16-byte load made it LDS-store bound.)
// All 9 * (Cin/16) * (Cout/16) 16x16 output tiles stay in registers (split over the 4 waves by (tap, ci tile)) across every
// tile the block visits; one fp32 partial [Cout][9*Cin] per block goes to the slab (summed by hd_wgrad_reduce).
#include "hd_common.h"

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int TH = 8, TW = 32, PH = TH + 2, PW = TW + 2;
constexpr int PS = 32;          // halves per staged dY pixel (64 B: an odd multiple of 64 B keeps the 4 k-rows of a transposed read apart)

template <int PITCH>
__device__ __forceinline__ f16x8 tr_frag8(const f16* row0_ptr) {
  // two transposed 4x16 block reads: reduction rows [0,4) and [4,8) relative to row0_ptr (rows are PITCH halves apart)
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(row0_ptr));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(row0_ptr + 4 * PITCH));
  f16x4 fa = __builtin_bit_cast(f16x4, a), fb = __builtin_bit_cast(f16x4, b);
  f16x8 r = {fa[0], fa[1], fa[2], fa[3], fb[0], fb[1], fb[2], fb[3]};
  return r;
}

// DUAL: the decoder concat gathered in place -- channels [0, CIN/2) from the nearest-2x upsampled tensor x (pixel (y >> 1, x >> 1)),
// channels [CIN/2, CIN) from the skip tensor x2.
template <int CIN, int MT, bool DUAL>
__global__ __launch_bounds__(256) void wgrad3x3_small_kernel(const f16* __restrict__ x, const f16* __restrict__ x2, const f16* __restrict__ dy,
                                                             float* __restrict__ slab,
                                                             int N, int Hsrc, int Wsrc, int H, int W, int Cout, int up1, int tiles_total,
                                                             int tiles_x, int tiles_y, const float* __restrict__ in_scale,
                                                             const float* __restrict__ in_shift, int in_relu) {
  constexpr int PSX = CIN <= 32 ? 32 : CIN + 32;      // halves per staged input pixel: an odd multiple of 64 bytes
  constexpr int CT = CIN / 16;
  constexpr int UNITS = 9 * CT;                 // (tap, 16-channel ci tile) pairs; a wave owns units wave, wave+4, ... for every cout tile
  constexpr int UPW = (UNITS + 3) / 4;
  constexpr int KTOT = 9 * CIN;
  __shared__ __attribute__((aligned(16))) f16 s_x[PH * PW * PSX];      // [py][px][channel], 21.8 KB (108.8 KB for the 128-channel concat)
  __shared__ __attribute__((aligned(16))) f16 s_dy[TH * TW * PS];      // [oy][ox][channel], 16 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int tq = li >> 2, tp = li & 3;          // transposed-read lane geometry inside a 16-lane group (see wgrad.hip)
  const int krow = 8 * g + tq;                  // first reduction row (pixel along x) this lane addresses
  const int dyc8 = Cout / 8;                    // 16-byte chunks per dY pixel (Cout % 8 == 0)

  f32x4_t acc[UPW][MT];
#pragma unroll
  for (int q = 0; q < UPW; ++q)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[q][m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // channels beyond Cout (padding rows of the 16-row MFMA tile) stay zero
  for (int e = tid; e < TH * TW * PS / 8; e += 256) reinterpret_cast<f16x8*>(s_dy)[e] = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};

  // software pipeline: the next tile's global loads are in flight while the current tile is multiplied
  constexpr int C8 = CIN / 8;
  constexpr int XL = (PH * PW * C8 + 255) / 256;   // 16-byte input loads per thread per tile (3 / 6)
  constexpr int DL = 2 * MT;                       // dY loads per thread per tile (Cout / 8 <= 2 * MT)
  f16x8 rx[XL], rd[DL];
  // consumer-side BatchNorm of the x operand (hd_wgrad_args.in_scale): see conv3x3_small.hip
  const bool fuse_bn = in_scale != nullptr;
  float isc[8], ish[8];
  if (fuse_bn) {
    const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(in_scale + (tid % C8) * 8), a1 = *reinterpret_cast<const f32x4_t*>(in_scale + (tid % C8) * 8 + 4);
    const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(in_shift + (tid % C8) * 8), b1 = *reinterpret_cast<const f32x4_t*>(in_shift + (tid % C8) * 8 + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      isc[k] = a0[k]; isc[4 + k] = a1[k];
      ish[k] = b0[k]; ish[4 + k] = b1[k];
    }
  }
  unsigned vmask = 0;
  auto gload = [&](int tile) {
    vmask = 0;
    const bool live = tile < tiles_total;
    int b = live ? tile : 0;
    const int tx = b % tiles_x;
    b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    constexpr int CSRC = DUAL ? CIN / 2 : CIN;             // channels of each source tensor
    const f16* xb = x + (size_t)n * Hsrc * Wsrc * CSRC;
    const f16* x2b = DUAL ? x2 + (size_t)n * H * W * CSRC : nullptr;
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int e = tid + i * 256;
      const int c8 = e % C8, pp = e / C8;
      const int py = pp / PW, px = pp - py * PW;
      const int hi = y0 + py - 1, wi = x0 + px - 1;
      f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (live && e < PH * PW * C8 && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) {
        if (DUAL && c8 >= C8 / 2) {
          v = *reinterpret_cast<const f16x8*>(x2b + ((size_t)hi * W + wi) * CSRC + (c8 - C8 / 2) * 8);
        } else {
          const int hs = up1 ? (hi >> 1) : hi, ws = up1 ? (wi >> 1) : wi;
          v = *reinterpret_cast<const f16x8*>(xb + ((size_t)hs * Wsrc + ws) * CSRC + c8 * 8);
        }
        vmask |= 1u << i;
      }
      rx[i] = v;
    }
    const f16* db = dy + (size_t)n * H * W * Cout;
#pragma unroll
    for (int i = 0; i < DL; ++i) {                 // thread = pixel tid, chunk i
      const int oy = tid / TW, ox = tid - oy * TW;
      f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (live && i < dyc8 && y0 + oy < H && x0 + ox < W) v = *reinterpret_cast<const f16x8*>(db + ((size_t)(y0 + oy) * W + x0 + ox) * Cout + i * 8);
      rd[i] = v;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int e = tid + i * 256;
      f16x8 v = rx[i];
      if (fuse_bn && ((vmask >> i) & 1u)) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float f = hd_bn_affine((float)v[k], isc[k], ish[k]);
          if (in_relu) f = fmaxf(f, 0.f);
          v[k] = (f16)f;
        }
      }
      if (e < PH * PW * C8) *reinterpret_cast<f16x8*>(s_x + (e / C8) * PSX + (e % C8) * 8) = v;
    }
#pragma unroll
    for (int i = 0; i < DL; ++i)
      if (i < dyc8) *reinterpret_cast<f16x8*>(s_dy + tid * PS + i * 8) = rd[i];
  };

  gload(blockIdx.x);
  for (int tile = blockIdx.x; tile < tiles_total; tile += gridDim.x) {
    __syncthreads();                          // the previous tile's fragment reads are done
    lstore();
    __syncthreads();
    gload(tile + gridDim.x);
    // ---- one MFMA per (unit, cout tile, output row): K = the row's 32 pixels
#pragma unroll
    for (int q = 0; q < UPW; ++q) {
      const int u = wave + 4 * q;              // wave-uniform: EXEC stays full for the transposed reads
      if (u < UNITS) {
        const int c = u % CT, t = u / CT;
        const int kh = t / 3, kw = t - kh * 3;
        const f16* ap = s_dy + krow * PS + 4 * tp;
        const f16* bp = s_x + (kh * PW + kw + krow) * PSX + c * 16 + 4 * tp;
#pragma unroll
        for (int oy = 0; oy < TH; ++oy) {
          const f16x8 bf = tr_frag8<PSX>(bp + oy * PW * PSX);
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const f16x8 af = tr_frag8<PS>(ap + oy * TW * PS + m * 16);
            acc[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc[q][m], 0, 0, 0);
          }
        }
      }
    }
  }
  // ---- partial dW of this block: C[i = co][j = ci] of (unit u = (t, c), cout tile m): lane holds column j = li, rows 4g + r
  float* out = slab + (size_t)blockIdx.x * Cout * KTOT;
#pragma unroll
  for (int q = 0; q < UPW; ++q) {
    const int u = wave + 4 * q;
    if (u < UNITS) {
      const int c = u % CT, t = u / CT;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = m * 16 + g * 4 + r;
          if (co < Cout) out[(size_t)co * KTOT + t * CIN + c * 16 + li] = acc[q][m][r];
        }
    }
  }
}

}  // namespace

bool hd_wgrad_small_eligible(const hd_wgrad_args* a) {
  if (!(a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad == 1 && a->Cout % 8 == 0 && a->Cout <= 32 && a->Ho == a->Hin && a->Wo == a->Win)) return false;
  if (a->x2) return a->up1 && a->C1 == 64 && a->C2 == 64 && !a->in_scale;      // decoder concat 64 (upsampled) + 64 (skip) channels
  return a->C2 == 0 && (a->C1 == 16 || a->C1 == 32);
}

void hd_wgrad_small_launch(const hd_wgrad_args* a, hipStream_t s) {
  const int tiles_x = hd_cdiv(a->Wo, TW), tiles_y = hd_cdiv(a->Ho, TH);
  const int total = a->N * tiles_x * tiles_y;
  dim3 grid(a->nsplit);
  const f16* x = (const f16*)a->x;
  const f16* x2 = (const f16*)a->x2;
  const f16* dy = (const f16*)a->dy;
#define LAUNCH(CI, M_, DU)                                                                                                                  \
  hipLaunchKernelGGL((wgrad3x3_small_kernel<CI, M_, DU>), grid, dim3(256), 0, s, x, x2, dy, a->slab, a->N, a->Hsrc, a->Wsrc, a->Hin, a->Win, \
                     a->Cout, a->up1, total, tiles_x, tiles_y, a->in_scale, a->in_shift, a->in_relu)
  if (a->x2) { if (a->Cout <= 16) LAUNCH(128, 1, true); else LAUNCH(128, 2, true); }
  else if (a->C1 == 16) { if (a->Cout <= 16) LAUNCH(16, 1, false); else LAUNCH(16, 2, false); }
  else { if (a->Cout <= 16) LAUNCH(32, 1, false); else LAUNCH(32, 2, false); }
#undef LAUNCH
}
