// Data gradient of a 7x7 / stride-2 / pad-3 convolution with FEW input channels (the ResNet stem: torchvision ResNet.conv1 [EXT], reached
// from src/models/detector.py:24-141 through the frozen detector's backward pass -- it produces dL/d(hallucinated image), the gradient that
// trains HalluciDet) in SUB-PIXEL form.  The implicit-GEMM route treats it as a stride-1 convolution over the zero-dilated gradient
// with Cout = 8 padded to a 32-wide tile: 36 GFLOP of MFMA work for 3.4 GFLOP of arithmetic (95 us).  Here the 2x2 output pixels
// (2I + a, 2J + b) that share a low-resolution position are ONE GEMM row-block: they depend on the same 4 x 4 window of the gradient
// dy[I - 1 .. I + 2, J - 1 .. J + 2, 0..63] through
//     dx[2I + a, 2J + b, c] = sum_{di, dj, ch} dy[I + di, J + dj, ch] * w[ch][a + 3 - 2 di][b + 3 - 2 dj][c]      (kh, kw outside 0..6: zero)
// so M = N * Hl * Wl low-resolution pixels, N = 4 parity classes x 4 channels (3 real) = 16, K = 16 taps x 64 channels = 1024: a 16-row
// weight matrix that lives in REGISTERS (32 fragments of v_mfma_f32_16x16x32_f16 per lane), the gradient patch staged once in LDS.
// Same products as the im2col form, fp32 accumulation in a different order.
#include "hd_common.h"

namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 32;                 // low-resolution tile
constexpr int PH = TH + 3, PW = TW + 3;        // window rows / columns -1 .. +2
constexpr int CH = 64;
constexpr int PITCH = CH + 8;                  // halves per patch pixel in LDS: 144 bytes -- sixteen lanes that read the same chunk of consecutive
                                               // pixels start 36 banks apart (distinct 4-bank groups), and every tap is a CONSTANT byte offset
constexpr int KSTEPS = 32;                     // 16 taps x 2 halves of 32 channels
constexpr int CHUNKS = PH * PW * 8;            // 16-byte chunks of the patch
constexpr int XL = (CHUNKS + 255) / 256;

__global__ __launch_bounds__(256) void conv7x7s2_dgrad_thin_kernel(const f16* __restrict__ dy, const f16* __restrict__ maskz,
                                                                      const f16* __restrict__ w16, f16* __restrict__ dx, int N, int Hl, int Wl,
                                                                      int H, int W, int tiles_x, int tiles_y) {
  __shared__ __attribute__((aligned(16))) f16 s_patch[PH * PW * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pl = lane & 15, g = lane >> 4;
  const int tiles_total = N * tiles_x * tiles_y;

  // ---- the whole 16 x 1024 weight matrix of this lane's row / K group: 32 fragments
  f16x8 wr[KSTEPS];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) wr[s] = *reinterpret_cast<const f16x8*>(w16 + (size_t)pl * (KSTEPS * 32) + s * 32 + g * 8);

  f16x8 rx[XL];
  auto gload = [&](int tile) {
    const bool live = tile < tiles_total;
    int b = live ? tile : 0;
    const int tx = b % tiles_x;
    b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int e = tid + i * 256;
      const int c8 = e & 7, pp = e >> 3;
      const int py = pp / PW, px = pp - py * PW;
      const int hi = ty * TH + py - 1, wi = tx * TW + px - 1;
      f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (live && e < CHUNKS && (unsigned)hi < (unsigned)Hl && (unsigned)wi < (unsigned)Wl) {
        const size_t off = (((size_t)n * Hl + hi) * Wl + wi) * CH + c8 * 8;
        v = *reinterpret_cast<const f16x8*>(dy + off);
        if (maskz) {                            // ReLU backward of the stem fused into the staging: dy * (z > 0)
          const f16x8 z = *reinterpret_cast<const f16x8*>(maskz + off);
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = ((float)z[k] > 0.f) ? v[k] : (f16)0.f;
        }
      }
      rx[i] = v;
    }
  };

  gload(blockIdx.x);
  for (int tile = blockIdx.x; tile < tiles_total; tile += gridDim.x) {
    int bid = tile;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int n = bid / tiles_y;
    __syncthreads();                            // the previous tile's reads of s_patch are done
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int e = tid + i * 256;
      const int c8 = e & 7, pp = e >> 3;
      if (e < CHUNKS) *reinterpret_cast<f16x8*>(s_patch + pp * PITCH + c8 * 8) = rx[i];
    }
    __syncthreads();
    gload(tile + gridDim.x);

    // ---- wave w owns low-resolution rows 2w, 2w + 1: four pixel groups of 16.  The 128 MFMAs of a tile (32 K steps x 4 pixel groups) run
    //      as ONE software-pipelined stream: the fragment of MFMA j + RING is requested right after MFMA j is issued (left to the compiler
    //      every MFMA waited for its own fragment: an LDS round trip per 16 clocks of matrix work, 15 us per tile).  A fragment's address
    //      is the pixel group's base (one register each) plus a compile-time byte offset (tap and channel half).
    f32x4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const char* pb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int oy = wave * 2 + (t >> 1), ox = (t & 1) * 16 + pl;
      pb[t] = reinterpret_cast<const char*>(s_patch) + ((oy * PW + ox) * PITCH + g * 8) * 2;
    }
    constexpr int RING = 6, NMM = KSTEPS * 4;
#define HD_SD_OFF(S) (((((S) >> 1) >> 2) * PW + (((S) >> 1) & 3)) * PITCH + ((S) & 1) * 32) * 2
#define HD_SD_B(J) (*reinterpret_cast<const f16x8*>(pb[(J) & 3] + HD_SD_OFF((J) >> 2)))
    f16x8 bq[RING];
#pragma unroll
    for (int j = 0; j < RING; ++j) bq[j] = HD_SD_B(j);
#pragma unroll
    for (int j = 0; j < NMM; ++j) {
      __builtin_amdgcn_sched_barrier(0);
      acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[j >> 2], bq[j % RING], acc[j & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (j + RING < NMM) bq[j % RING] = HD_SD_B(j + RING);
    }
#undef HD_SD_B
#undef HD_SD_OFF
    // ---- lane (pl, g): the four channels of output pixel (2I + (g >> 1), 2J + (g & 1)) -> one 16-byte store (channels 3..7 zero)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int I = ty * TH + wave * 2 + (t >> 1), J = tx * TW + (t & 1) * 16 + pl;
      const int p = 2 * I + (g >> 1), q = 2 * J + (g & 1);
      if (I < Hl && J < Wl && p < H && q < W) {
        f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
        o[0] = (f16)acc[t][0]; o[1] = (f16)acc[t][1]; o[2] = (f16)acc[t][2]; o[3] = (f16)acc[t][3];
        *reinterpret_cast<f16x8*>(dx + (((size_t)n * H + p) * W + q) * 8) = o;
      }
    }
  }
}

}  // namespace

extern "C" int hd_conv7x7s2_dgrad_thin(const void* dy, const void* mask_z, const void* w16, void* dx, int N, int Hl, int Wl, int H, int W, void* stream) {
  HD_CHECK_ARG(dy && w16 && dx && N > 0 && Hl > 0 && Wl > 0, "hd_conv7x7s2_dgrad_thin: bad args");
  HD_CHECK_ARG(Hl == (H + 6 - 7) / 2 + 1 && Wl == (W + 6 - 7) / 2 + 1, "hd_conv7x7s2_dgrad_thin: (Hl, Wl) = (%d, %d) is not the 7x7 / stride-2 / pad-3 output extent of (%d, %d)", Hl, Wl, H, W);
  const int tiles_x = (Wl + TW - 1) / TW, tiles_y = (Hl + TH - 1) / TH;
  const int tiles = N * tiles_x * tiles_y;
  // an odd H / W leaves the last input row / column without an output pixel pair of its own: rows 2I + a with I < Hl cover 0 .. 2 Hl - 1 >= H - 1
  hipLaunchKernelGGL(conv7x7s2_dgrad_thin_kernel, dim3(tiles < 256 ? tiles : 256), dim3(256), 0, (hipStream_t)stream, (const f16*)dy, (const f16*)mask_z,
                     (const f16*)w16, (f16*)dx, N, Hl, Wl, H, W, tiles_x, tiles_y);
  HD_CHECK_LAUNCH();
  return HD_OK;
}
