// 3x3 / stride 1 / pad 1 convolution for SMALL channel counts (Cin in {8,16,32}, Cout in {16,32}): the U-Net decoder's
// 16/32-channel layers at 256x320 / 512x640 and their data gradients.
//
// Why a separate kernel: in the implicit-GEMM family these layers run at ~4x their HBM time (16->16 @512x640: 108 us for
// 168 MB): a pixel is 32 bytes, the fill moves it as 16-byte pieces, 18 per pixel (9 taps x 2 chunks), and the 32-wide
// MFMA tile is half empty.  Here a block walks 8x32 output tiles, stages each tile's (8+2)x(32+2) input patch ONCE (16-byte
// loads, each input byte read ~1.3x; the next tile's patch is prefetched into registers during the MFMAs), keeps the
// whole weight matrix in LDS, and uses v_mfma_f32_16x16x32_f16 with the
// WEIGHTS as the A operand: C[cout][pixel], so a lane ends up with 4 consecutive output channels of one pixel and a
// wave-store writes 16 pixels x 32 bytes contiguously -- no transpose through LDS.
//   K index k = tap*Cin + ci (the igemm weight layout); a 32-deep K step is 4 / 2 / 1 taps for Cin = 8 / 16 / 32; the K
//   tail (tap >= 9) multiplies zero weights.
// Epilogue: NHWC f16 with optional BatchNorm partial sums of the f16-rounded output (per block: [2][Cout]; no bias /
// activation: conv -> BN -> ReLU units and their data gradients), or NCHW fp32 with bias + activation for <= 16 real
// output channels (the segmentation head).  No residual / mask.
#include "hd_common.h"
#include "conv_params.h"

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 32;               // output tile
constexpr int PH = TH + 2, PW = TW + 2;      // input patch

// sum over the 16 lanes of a DPP row, left in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv3x3_small_kernel(ConvP p) {
  constexpr int KTOT = 9 * CIN;
  constexpr int KSTEPS = (KTOT + 31) / 32;
  constexpr int KPAD = KSTEPS * 32;
  constexpr int MT = COUT / 16;               // 16-row (cout) tiles
  constexpr int TPS = 32 / CIN;               // taps per K step
  constexpr int PATCH_HALVES = PH * PW * CIN > 16 * TH * TW * 2 ? PH * PW * CIN : 16 * TH * TW * 2;   // also the head's fp32 [16][8][32] staging tile
  __shared__ __attribute__((aligned(16))) f16 s_patch[PATCH_HALVES];
  __shared__ __attribute__((aligned(16))) f16 s_w[COUT * KPAD];
  __shared__ float s_red[4][COUT][2];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
  const int tiles_total = p.N * tiles_x * tiles_y;

  // ---- weights -> LDS once per block, zero-padded K tail
  for (int e = tid; e < COUT * KPAD / 8; e += 256) {
    const int co = e / (KPAD / 8), k8 = (e - co * (KPAD / 8)) * 8;
    f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (k8 < KTOT && co < p.Cout) v = *reinterpret_cast<const f16x8*>(p.w + (size_t)co * KTOT + k8);     // KTOT % 8 == 0
    *reinterpret_cast<f16x8*>(s_w + co * KPAD + k8) = v;
  }
  // ---- software pipeline over the block's tiles: the next input patch is in flight while this one is multiplied
  constexpr int C8 = CIN / 8;
  constexpr int XL = (PH * PW * C8 + 255) / 256;
  f16x8 rx[XL];
  // consumer-side BatchNorm (hd_conv_args.in_scale): a thread always stages the same 8 channels (256 % C8 == 0), so their coefficients
  // live in registers; `vmask` remembers which staged vectors are real pixels -- zero padding must stay zero after the affine map
  const bool fuse_bn = p.in_scale != nullptr;
  float isc[8], ish[8];
  if (fuse_bn) {
    const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(p.in_scale + (tid % C8) * 8), a1 = *reinterpret_cast<const f32x4_t*>(p.in_scale + (tid % C8) * 8 + 4);
    const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(p.in_shift + (tid % C8) * 8), b1 = *reinterpret_cast<const f32x4_t*>(p.in_shift + (tid % C8) * 8 + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      isc[k] = a0[k]; isc[4 + k] = a1[k];
      ish[k] = b0[k]; ish[4 + k] = b1[k];
    }
  }
  const bool in_relu = p.in_relu != 0;
  unsigned vmask = 0;
  auto gload = [&](int tile) {
    vmask = 0;
    const bool live = tile < tiles_total;
    int b = live ? tile : 0;
    const int tx = b % tiles_x;
    b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const f16* xb = p.x + (size_t)n * p.Hsrc * p.Wsrc * CIN;
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int e = tid + i * 256;
      const int c8 = e % C8, pp = e / C8;
      const int py = pp / PW, px = pp - py * PW;
      const int hi = ty * TH + py - 1, wi = tx * TW + px - 1;
      f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (live && e < PH * PW * C8 && (unsigned)hi < (unsigned)p.Hin && (unsigned)wi < (unsigned)p.Win) {
        const int hs = p.up1 ? (hi >> 1) : hi, ws = p.up1 ? (wi >> 1) : wi;
        v = *reinterpret_cast<const f16x8*>(xb + ((size_t)hs * p.Wsrc + ws) * CIN + c8 * 8);
        vmask |= 1u << i;
      }
      rx[i] = v;
    }
  };
  const int pl = lane & 15, g = lane >> 4;    // pixel (B column) / cout (A row) index, K group

  gload(blockIdx.x);
  for (int tile = blockIdx.x; tile < tiles_total; tile += gridDim.x) {
    int bid = tile;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int n = bid / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    __syncthreads();                          // the previous tile's reads of s_patch / s_red are done (first pass: s_w is written)
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int e = tid + i * 256;
      f16x8 v = rx[i];
      if (fuse_bn && ((vmask >> i) & 1u)) {       // hd_bn_apply's arithmetic, applied on the way into LDS
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float f = hd_bn_affine((float)v[k], isc[k], ish[k]);
          if (in_relu) f = fmaxf(f, 0.f);
          v[k] = (f16)f;
        }
      }
      if (e < PH * PW * C8) *reinterpret_cast<f16x8*>(s_patch + (e / C8) * CIN + (e % C8) * 8) = v;
    }
    __syncthreads();
    gload(tile + gridDim.x);

    // ---- MFMA: wave w owns output rows 2w, 2w+1 (4 pixel tiles of 16)
    f32x4_t acc[4][MT];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[t][m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      int tap = ks * TPS + (g * 8) / CIN;       // this lane's 8 K values: one tap, channels ci0..ci0+7
      const int ci0 = (g * 8) % CIN;
      if (tap > 8) tap = 8;                      // K tail: zero weights; read something valid
      const int kh = tap / 3, kw = tap - kh * 3;
      f16x8 af[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) af[m] = *reinterpret_cast<const f16x8*>(s_w + (m * 16 + pl) * KPAD + ks * 32 + g * 8);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int oy = wave * 2 + (t >> 1), ox = (t & 1) * 16 + pl;
        const f16x8 bf = *reinterpret_cast<const f16x8*>(s_patch + ((oy + kh) * PW + ox + kw) * CIN + ci0);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m], bf, acc[t][m], 0, 0, 0);
      }
    }

    // ---- epilogue: lane holds couts m*16 + g*4 .. +3 of pixel (oy, ox)
    if (p.out_mode == HD_OUT_NCHW_F32) {
      // segmentation head (base/heads.py:23-27): Cout <= 16 real channels, bias, activation, fp32 planes; 16 consecutive
      // pixels of a plane per store
      // (through LDS: a lane holds 4 channels of ONE pixel, a plane row wants 32 pixels of ONE channel per store)
      float* yf = reinterpret_cast<float*>(p.y) + (size_t)n * p.Cout * p.Ho * p.Wo;
      float* s_out = reinterpret_cast<float*>(s_patch);      // [Cout <= 16][8][32] fp32 (s_patch is sized for it)
      __syncthreads();                                       // every wave is done reading the patch
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int oyl = wave * 2 + (t >> 1), oxl = (t & 1) * 16 + pl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = g * 4 + r;
          if (co < p.Cout) {
            float v = acc[t][0][r] + (p.bias ? p.bias[co] : 0.f);
            if (p.act == HD_ACT_RELU) v = fmaxf(v, 0.f);
            else if (p.act == HD_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
            s_out[(co * TH + oyl) * TW + oxl] = v;
          }
        }
      }
      __syncthreads();
      {
        const int oyl = tid / TW, oxl = tid - oyl * TW;
        if (y0 + oyl < p.Ho && x0 + oxl < p.Wo)
          for (int co = 0; co < p.Cout; ++co) yf[((size_t)co * p.Ho + y0 + oyl) * p.Wo + x0 + oxl] = s_out[(co * TH + oyl) * TW + oxl];
      }
      continue;
    }
    if (p.pool2) {
      // out_pool2 (round 5): this launch is the data gradient of a decoder block's first convolution whose whole input is the
      // nearest-2x upsampled tensor (no skip: decoders/unet/decoder.py:38-41 with skip = None), so the gradient the block below
      // receives is the 2 x 2 SUM of what this kernel would write -- formed here, in fp32, from the two tile rows a lane already
      // holds (t, t + 2) and its column neighbour (lane ^ 1): the full-resolution tensor (168 MB at 8 x 512 x 640 x 32) is neither
      // written nor read back by a pooling launch.
      f16* yb2 = reinterpret_cast<f16*>(p.y) + (size_t)n * (p.Ho >> 1) * (p.Wo >> 1) * COUT;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int oy = y0 + wave * 2, ox = x0 + t * 16 + pl;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          f16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[t][m][r] + acc[t + 2][m][r];
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]: lane ^ 1
            o[r] = (f16)v;
          }
          if (!(pl & 1) && oy < p.Ho && ox < p.Wo)
            *reinterpret_cast<f16x4*>(yb2 + ((size_t)(oy >> 1) * (p.Wo >> 1) + (ox >> 1)) * COUT + m * 16 + g * 4) = o;
        }
      }
      continue;
    }
    f16* yb = reinterpret_cast<f16*>(p.y) + (size_t)n * p.Ho * p.Wo * COUT;
    float ssum[MT][4], ssq[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) ssum[m][r] = ssq[m][r] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int oy = y0 + wave * 2 + (t >> 1), ox = x0 + (t & 1) * 16 + pl;
      if (oy < p.Ho && ox < p.Wo) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          f16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            o[r] = (f16)acc[t][m][r];
            const float vr = (float)o[r];
            ssum[m][r] += vr;
            ssq[m][r] += vr * vr;
          }
          *reinterpret_cast<f16x4*>(yb + ((size_t)oy * p.Wo + ox) * COUT + m * 16 + g * 4) = o;
        }
      }
    }
    if (p.stats) {
      // the 16 lanes of a K group hold the same couts for 16 different pixels: fold them, then the 4 waves through LDS
      // (DPP row operations: one VALU instruction per step -- the __shfl_xor form was a ds_bpermute + wait + add, 64 LDS-pipe round trips
      //  per tile; after quad xor-1, quad xor-2, half-row mirror, row mirror every lane holds the sum over its 16-lane row)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ssum[m][r] = row16_sum(ssum[m][r]);
          ssq[m][r] = row16_sum(ssq[m][r]);
        }
      if (pl == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s_red[wave][m * 16 + g * 4 + r][0] = ssum[m][r];
            s_red[wave][m * 16 + g * 4 + r][1] = ssq[m][r];
          }
      }
      __syncthreads();
      if (tid < COUT) {
        float s = 0.f, s2 = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) {
          s += s_red[w4][tid][0];
          s2 += s_red[w4][tid][1];
        }
        p.stats[((size_t)tile * 2 + 0) * COUT + tid] = s;
        p.stats[((size_t)tile * 2 + 1) * COUT + tid] = s2;
      }
    }
  }
}

}  // namespace

bool hd_conv_small_eligible(const ConvP& p) {
  if (!(p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.in_dil == 1 && p.C2 == 0 && p.x2 == nullptr &&
        (p.C1 == 8 || p.C1 == 16 || p.C1 == 32) && !p.res && !p.mask && p.Ho == p.Hin && p.Wo == p.Win))
    return false;
  if (p.out_mode == HD_OUT_NHWC_F32) return false;
  if (p.out_mode == HD_OUT_NCHW_F32) return p.Cout <= 16 && !p.stats;                       // head: bias + activation allowed
  return (p.Cout == 16 || p.Cout == 32) && !p.bias && p.act == HD_ACT_NONE;                   // conv -> BN units, data gradients
}

// out_pool2: the 2 x 2 sum-pooled output form (plain NHWC f16 output, no statistics, even extent)
bool hd_conv_small_pool2_ok(const ConvP& p) {
  return hd_conv_small_eligible(p) && p.out_mode == HD_OUT_NHWC_F16 && !p.stats && !p.in_scale && (p.Ho % 2) == 0 && (p.Wo % 2) == 0 &&
         (p.pool2 == 0 || (p.pool2 == p.Cout && !p.y2));
}

int hd_conv_small_tiles(const ConvP& p) { return p.N * hd_cdiv(p.Ho, TH) * hd_cdiv(p.Wo, TW); }

void hd_conv_launch_small(ConvP& p, hipStream_t s) {
  // persistent blocks: a few per CU, each walking tiles blockIdx.x, blockIdx.x + grid, ... (BN partial sums stay per TILE)
  const int tiles = hd_conv_small_tiles(p);
  dim3 grid(tiles < 1536 ? tiles : 1536);
#define LAUNCH(CI, CO) hipLaunchKernelGGL((conv3x3_small_kernel<CI, CO>), grid, dim3(256), 0, s, p)
  if (p.C1 == 8) { if (p.Cout <= 16) LAUNCH(8, 16); else LAUNCH(8, 32); }
  else if (p.C1 == 16) { if (p.Cout <= 16) LAUNCH(16, 16); else LAUNCH(16, 32); }
  else { if (p.Cout <= 16) LAUNCH(32, 16); else LAUNCH(32, 32); }
#undef LAUNCH
}
