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
// Bank conflicts: a transposed 8-byte read is serviced per 32-lane half, i.e. by TWO 16-lane groups.  With the natural reduction order
// (group g holds pixels 8g .. 8g+7) the two groups address pixel rows 8 apart, which any pitch that keeps 4 consecutive rows in
// distinct 64-byte quarters maps to the same banks: a 2-way conflict on every fragment read (LDS bank-conflict share 0.51 in
// profiles/r02_conv_mfma_util.json).  The reduction order is free as long as both operands agree, so group g takes pixels
// 4g .. 4g+3 and 16+4g .. 16+4g+3: a 32-lane half now reads 8 CONSECUTIVE pixel rows, 32 bytes each, and a pitch of an odd multiple of
// 32 bytes spreads them over all 256 bytes of the banks.
// All 9 * (Cin/16) * (Cout/16) 16x16 output tiles stay in registers (split over the 4 waves by (tap, ci tile)) across every
// tile the block visits; one fp32 partial [Cout][9*Cin] per block goes to the slab (summed by hd_wgrad_reduce).
#include "hd_common.h"

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

constexpr int TH = 8, TW = 32, PH = TH + 2, PW = TW + 2;

template <int PITCH>
__device__ __forceinline__ f16x8 tr_frag8(const f16* row0_ptr) {
  // two transposed 4x16 block reads: pixel rows [0,4) and [16,20) relative to row0_ptr (rows are PITCH halves apart)
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(row0_ptr));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(row0_ptr + 16 * PITCH));
  f16x4 fa = __builtin_bit_cast(f16x4, a), fb = __builtin_bit_cast(f16x4, b);
  f16x8 r = {fa[0], fa[1], fa[2], fa[3], fb[0], fb[1], fb[2], fb[3]};
  return r;
}

// DUAL: the decoder concat gathered in place -- channels [0, CIN/2) from the nearest-2x upsampled tensor x (pixel (y >> 1, x >> 1)),
// channels [CIN/2, CIN) from the skip tensor x2.
// NT: threads per block.  256 for the 16/32-channel layers (several blocks per CU); 512 for the concat, whose 72 units x 2 cout tiles of
// accumulators plus the 98 KB patch prefetch do not fit the registers of four waves (the scheduler spills), and whose one block per
// CU otherwise leaves a single wave per SIMD to cover its own LDS latency.
template <int CIN, int MT, bool DUAL, int NT>
__global__ __launch_bounds__(NT) void wgrad3x3_small_kernel(const f16* __restrict__ x, const f16* __restrict__ x2, const f16* __restrict__ dy,
                                                             float* __restrict__ slab,
                                                             int N, int Hsrc, int Wsrc, int H, int W, int Cout, int up1, int tiles_total,
                                                             int tiles_x, int tiles_y, const float* __restrict__ in_scale,
                                                             const float* __restrict__ in_shift, int in_relu) {
  constexpr int PSX = CIN % 32 == 0 ? CIN + 16 : CIN;   // halves per staged input pixel: an odd multiple of 32 bytes (32 / 96 / 288 B)
  constexpr int PS = MT == 1 ? 16 : 48;                 // halves per staged dY pixel, likewise (32 / 96 B)
  constexpr int CT = CIN / 16;
  constexpr int NW = NT / 64;
  // A wave owns units wave, wave+NW, ... for every cout tile.  A unit is a (tap, 16-channel ci tile) pair; in the concat variant
  // (KHS) it is a (kw, ci tile) pair carrying the accumulators of all three kh: one fragment of patch row r then feeds the
  // output rows r, r-1, r-2 of kh = 0, 1, 2, so each patch row is read once instead of three times.
  constexpr bool KHS = DUAL;
  constexpr int UNITS = (KHS ? 3 : 9) * CT;
  constexpr int UPW = (UNITS + NW - 1) / NW;
  constexpr int UPWF = UNITS / NW;              // rounds in which every wave has a unit
  constexpr int NACC = KHS ? 3 * UPW : UPW;     // accumulator sets per wave (x MT cout tiles)
  static_assert(!KHS || UNITS % NW == 0, "the kh-sharing loop has no partial round");
  constexpr int KTOT = 9 * CIN;
  __shared__ __attribute__((aligned(16))) f16 s_x[PH * PW * PSX];      // [py][px][channel], 10.9 / 32.6 KB (97.9 KB for the 128-channel concat)
  __shared__ __attribute__((aligned(16))) f16 s_dy[TH * TW * PS];      // [oy][ox][channel], 8 / 24 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int swave = __builtin_amdgcn_readfirstlane(wave);
  const int li = lane & 15, g = lane >> 4;
  const int tq = li >> 2, tp = li & 3;          // transposed-read lane geometry inside a 16-lane group (see wgrad.hip)
  const int krow = 4 * g + tq;                  // first pixel (along x) this lane addresses; the second block is 16 pixels on
  const int dyc8 = Cout / 8;                    // 16-byte chunks per dY pixel (Cout % 8 == 0)
  constexpr int DC8 = 2 * MT;                   // chunks per staged dY pixel (the cout tiles' 16 rows each; chunks >= dyc8 are zero)

  f32x4_t acc[NACC][MT];
#pragma unroll
  for (int q = 0; q < NACC; ++q)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[q][m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // software pipeline: the next tile's global loads are in flight while the current tile is multiplied
  constexpr int C8 = CIN / 8;
  constexpr int XL = (PH * PW * C8 + NT - 1) / NT;   // 16-byte input loads per thread per tile (3 / 6; 11 for the concat)
  constexpr int DL = TH * TW * DC8 / NT;             // dY loads per thread per tile
  f16x8 rx[XL], rd[DL];
  // consumer-side BatchNorm of the x operand (hd_wgrad_args.in_scale): see conv3x3_small.hip
  const bool fuse_bn = !DUAL && in_scale != nullptr;     // (the concat's operands are never raw: hd_wgrad_small_eligible)
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
      const int e = tid + i * NT;
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
    for (int i = 0; i < DL; ++i) {                 // consecutive threads: consecutive chunks of a pixel, then consecutive pixels
      const int e = tid + i * NT;
      const int ch = e % DC8, pix = e / DC8;
      const int oy = pix / TW, ox = pix - oy * TW;
      f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};         // channels beyond Cout (padding rows of the 16-row MFMA tile) are zero
      if (live && ch < dyc8 && y0 + oy < H && x0 + ox < W) v = *reinterpret_cast<const f16x8*>(db + ((size_t)(y0 + oy) * W + x0 + ox) * Cout + ch * 8);
      rd[i] = v;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int e = tid + i * NT;
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
    for (int i = 0; i < DL; ++i) {
      const int e = tid + i * NT;
      *reinterpret_cast<f16x8*>(s_dy + (e / DC8) * PS + (e % DC8) * 8) = rd[i];
    }
  };

  gload(blockIdx.x);
  for (int tile = blockIdx.x; tile < tiles_total; tile += gridDim.x) {
    __syncthreads();                          // the previous tile's fragment reads are done
    lstore();
    __syncthreads();
    gload(tile + gridDim.x);
    // ---- one MFMA per (row, unit, cout tile): K = the row's 32 pixels.  The row is the OUTER loop so that its dY fragments are read
    // once and reused by every unit of the wave (the LDS pipe, not the MFMA pipe, bounds this kernel: with the unit outermost each MFMA
    // pair cost three fragment reads)
    const f16* ap = s_dy + krow * PS + 4 * tp;
    const f16* bp0 = s_x + krow * PSX + 4 * tp;
    auto unit_frag = [&](int q, int oy) {        // x fragment of unit swave + NW q (wave-uniform: EXEC stays full for the transposed reads)
      const int u = swave + NW * q;
      const int c = u % CT, t = u / CT;
      const int kh = t / 3, kw = t - kh * 3;
      return tr_frag8<PSX>(bp0 + ((kh + oy) * PW + kw) * PSX + c * 16);
    };
    if constexpr (KHS) {
      f16x8 afw[3][MT];                          // dY fragments of the last three output rows
#pragma unroll
      for (int r = 0; r < PH; ++r) {
        if (r < TH) {
#pragma unroll
          for (int m = 0; m < MT; ++m) afw[r % 3][m] = tr_frag8<PS>(ap + r * TW * PS + m * 16);
        }
#pragma unroll
        for (int q = 0; q < UPW; ++q) {
          const int u = swave + NW * q;
          const int c = u % CT, kw = u / CT;
          const f16x8 bf = tr_frag8<PSX>(bp0 + (r * PW + kw) * PSX + c * 16);
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            const int oy = r - kh;
            if (oy >= 0 && oy < TH) {
#pragma unroll
              for (int m = 0; m < MT; ++m)
                acc[q * 3 + kh][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afw[oy % 3][m], bf, acc[q * 3 + kh][m], 0, 0, 0);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);       // keep the scheduler from hoisting every row's fragment reads to the top (it then spills)
      }
    } else {
#pragma unroll
      for (int oy = 0; oy < TH; ++oy) {          // the units every wave owns: straight-line code, no branch between the reads
        f16x8 af[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) af[m] = tr_frag8<PS>(ap + oy * TW * PS + m * 16);
#pragma unroll
        for (int q = 0; q < UPWF; ++q) {
          const f16x8 bf = unit_frag(q, oy);
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[q][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m], bf, acc[q][m], 0, 0, 0);
        }
        if (UPWF * MT > 8 && (oy & 1)) __builtin_amdgcn_sched_barrier(0);
      }
      if (UPWF < UPW && swave < UNITS - NW * UPWF) {   // the last, partial round of units (waves 0 .. UNITS % NW - 1)
#pragma unroll
        for (int oy = 0; oy < TH; ++oy) {
          const f16x8 bf = unit_frag(UPWF, oy);
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[UPW - 1][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tr_frag8<PS>(ap + oy * TW * PS + m * 16), bf, acc[UPW - 1][m], 0, 0, 0);
        }
      }
    }
  }
  // ---- partial dW of this block: C[i = co][j = ci] of (unit u = (t, c), cout tile m): lane holds column j = li, rows 4g + r
  float* out = slab + (size_t)blockIdx.x * Cout * KTOT;
#pragma unroll
  for (int ai = 0; ai < NACC; ++ai) {
    const int u = wave + NW * (KHS ? ai / 3 : ai);
    if (u < UNITS) {
      const int c = u % CT, t = KHS ? (ai % 3) * 3 + u / CT : u / CT;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = m * 16 + g * 4 + r;
          if (co < Cout) out[(size_t)co * KTOT + t * CIN + c * 16 + li] = acc[ai][m][r];
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
#define LAUNCH(CI, M_, DU)                                                                                                                          \
  hipLaunchKernelGGL((wgrad3x3_small_kernel<CI, M_, DU, (DU ? 512 : 256)>), grid, dim3(DU ? 512 : 256), 0, s, x, x2, dy, a->slab, a->N, a->Hsrc, \
                     a->Wsrc, a->Hin, a->Win, a->Cout, a->up1, total, tiles_x, tiles_y, a->in_scale, a->in_shift, a->in_relu)
  if (a->x2) { if (a->Cout <= 16) LAUNCH(128, 1, true); else LAUNCH(128, 2, true); }
  else if (a->C1 == 16) { if (a->Cout <= 16) LAUNCH(16, 1, false); else LAUNCH(16, 2, false); }
  else { if (a->Cout <= 16) LAUNCH(32, 1, false); else LAUNCH(32, 2, false); }
#undef LAUNCH
}
