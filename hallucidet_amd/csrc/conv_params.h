// Shared parameter block of the implicit-GEMM convolution kernels (conv_igemm_bk32.hip / conv_igemm_bk64.hip).
#pragma once
#include <stdlib.h>

#include "hd_common.h"

struct ConvP {
  const f16* x;
  const f16* x2;
  const f16* w;
  const float* bias;
  const f16* res;
  const f16* mask;
  void* y;
  void* y2;            // hd_conv_args.y2 (out_pool2 with a skip half: conv3x3_c32to128.hip)
  float* stats;
  unsigned xbytes, x2bytes, wbytes;
  int N, Hsrc, Wsrc, Hin, Win, C1, C2, Cin, Ho, Wo, Cout, KH, KW, stride, pad, up1, in_dil, act, out_mode;
  int M, cin8, nchunks, nk, Ktot;
  int gm, gn;          // grid extent in M / N tiles
  int tgroup;          // tile list order: super-rows of `tgroup` M tiles, M fastest inside (1 = plain N-fastest list): hd_conv_tile_order
  float inv_cin8, inv_kw;
  // Data gradient of a stride-2 convolution as FOUR output-parity classes (blockIdx.y = 2*ph + pw): output pixel (2i+ph, 2j+pw)
  // only sees the taps with kh = (pad - ph) mod 2 (+2, +4, ...) -- the others hit the zeros of the dilated input -- so a class
  // walks a quarter of the taps (none at all for three classes of a 1x1).  `par` is set by the dispatcher, the rest by the kernel.
  int par, ph, pw, Hc, Wc, t0h, t0w;
  int pool2;           // hd_conv_args.out_pool2: leading output channels that leave 2 x 2 sum-pooled (0 = off)
  int prio;            // 8-wave families: s_setprio policy (experiment knob HD_W8_PRIO: 0 none, 1 MFMA phase, 2 MEM phase)
  const float* in_scale;   // consumer-side BatchNorm of the x operand (hd_conv_args.in_scale / in_shift / in_relu): small-channel kernel only
  const float* in_shift;
  int in_relu;
  // producer-side sums of the next BatchNorm backward (hd_conv_args.bs_*): 8-wave 3x3 kernels; rows go to `stats`
  const f16* bs_y;
  const f16* bs_z;
  const float *bs_mean, *bs_invstd, *bs_gamma, *bs_beta;
  int bs_relu;
#ifdef HD_CONV_TRACE
  unsigned long long* trace;   // profiling builds only (tools/conv_trace.py): 16 stamps per block
  int trace_tid;               // which thread of the block stamps (HD_TRACE_TID, default 0)
#endif
};

// Several independent convolutions of ONE kernel variant in one grid (hd_conv2d_multi): the per-level convolutions of the FPN and of
// the RPN / RetinaNet / FCOS heads (same layer type on 4-5 feature maps, the small ones 16-150 tiles) ride in the tail of the largest
// level's launch instead of each paying a launch of their own.  Blocks [first[i], first[i+1]) belong to problem i.
constexpr int HD_CONV_MULTI_MAX = 10;
struct ConvMulti {
  ConvP p[HD_CONV_MULTI_MAX];
  int first[HD_CONV_MULTI_MAX + 1];
  int n;
};
static_assert(sizeof(ConvMulti) <= 3968, "ConvMulti travels in the kernel arguments (4 KiB)");

// per-block timeline stamps, compiled only into the profiling build (build.py --trace, tools/conv_trace.py)
#ifdef HD_CONV_TRACE
__device__ __forceinline__ unsigned long long hw_ids() {
  unsigned a, b;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(a));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(b));
  return ((unsigned long long)b << 32) | a;
}
#define HD_TRACE(slot, expr)                                                            \
  do {                                                                                  \
    if (threadIdx.x == p.trace_tid && p.trace) p.trace[(size_t)blockIdx.x * 16 + (slot)] = (expr); \
  } while (0)
#else
#define HD_TRACE(slot, expr) do {} while (0)
#endif

// a / d for 0 <= a < 2^24, d > 0 by a float reciprocal and one correction step (an integer division by a run-time divisor is ~40
// VALU instructions; the kernels' prologue does two per operand row: 1 000-1 500 of the ~4 400 set-up clocks of a block whose whole
// life is 20-25 k clocks on the short-K layers).  `inv` = hd_rcp(d).  Callers fall back to `/` when a may reach 2^24.
__device__ __forceinline__ float hd_rcp(int d) { return __builtin_amdgcn_rcpf((float)d); }
__device__ __forceinline__ int hd_fdiv(int a, int d, float inv) {
  int q = (int)((float)a * inv);
  int r = a - q * d;
  q = r < 0 ? q - 1 : q;
  r = r < 0 ? r + d : r;
  q = r >= d ? q + 1 : q;
  return q;
}

// Parity-class set-up (device): picks the class from blockIdx.y, shrinks M / nk to the class; returns false if this block has
// no tile in its class.
template <int BM, int CPT_>
__device__ __forceinline__ bool hd_par_setup(ConvP& p, int tile_m) {
  p.ph = blockIdx.y >> 1;
  p.pw = blockIdx.y & 1;
  p.Hc = (p.Ho - p.ph + 1) >> 1;
  p.Wc = (p.Wo - p.pw + 1) >> 1;
  p.M = p.N * p.Hc * p.Wc;
  p.t0h = (p.pad - p.ph) & 1;
  p.t0w = (p.pad - p.pw) & 1;
  const int nth = p.t0h < p.KH ? (p.KH - 1 - p.t0h) / 2 + 1 : 0;
  const int ntw = p.t0w < p.KW ? (p.KW - 1 - p.t0w) / 2 + 1 : 0;
  p.nk = nth * ntw * (p.cin8 / CPT_);
  return tile_m * BM < p.M;
}
// output pixel index (in the full N x Ho x Wo image) of GEMM row `m` of this block's class
__device__ __forceinline__ int hd_par_pixel(const ConvP& p, int m) {
  const int hw = p.Hc * p.Wc;
  const bool fast = p.M < (1 << 24);
  const int n = fast ? hd_fdiv(m, hw, hd_rcp(hw)) : m / hw;
  const int rem = m - n * hw;
  const int i = fast ? hd_fdiv(rem, p.Wc, hd_rcp(p.Wc)) : rem / p.Wc;
  const int jj = rem - i * p.Wc;
  return (n * p.Ho + 2 * i + p.ph) * p.Wo + 2 * jj + p.pw;
}

// Which tiles share an XCD's L2 (round 5).  Blocks are dealt round-robin over the 8 XCDs and each XCD gets a contiguous run of the
// tile list.  With the N tiles fastest (group 1) a run is a few M tiles x ALL N tiles, walked M tile by M tile: the pixels of an M
// tile are fetched once and the weight matrix is re-walked for every M tile -- free while it stays in the XCD's 4 MiB L2 (the U-Net,
// the large detector maps), but the detector's deep stages and the box head have weight matrices of 2 - 26 MB: 10x10x512 3x3 at
// batch 8 streamed its 4.7 MB of weights from beyond L2 in every XCD for every M tile (1.18 MB per 64x64 block at the 33 - 49 GB/s a
// CU gets from the Infinity Cache: what bound those launches), fc6's data gradient its 25.7 MB four times per XCD (~820 MB of
// traffic for a 137 MB problem).  Grouped order (the list is cut into super-rows of `group` M tiles, M tiles fastest inside one):
// an XCD's run is then `group` M tiles x a range of N tiles, the pixels of the group stay L2-resident and every weight tile is
// fetched once per XCD that shares the super-row.  Measured (tools/probe_ks.py, same box): fc6's data gradient 212.9 -> 197.7 us, the
// 10x10x512 3x3 layers 26.0 -> 25.3 / 31.5 -> 30.8 us; applied to the 2 MB matrices too it cost 16.5 -> 21.7 us on 24x10x10x2048 -> 512,
// hence the L2-size threshold below.  With s XCDs per super-row the traffic beyond L2 is s * A + 8 * W / s (A, W =
// input / weight bytes): s is chosen to minimise it among {1, 2, 4, 8} with the group's pixels (A * s / 8) within half the L2.
// The arithmetic does not change (bit-identical).  HD_CONV_TGROUP=0: always N-fastest (A/B).
static inline int hd_conv_tile_order(const ConvP& p) {
  static const int on = [] { const char* v = getenv("HD_CONV_TGROUP"); return v ? atoi(v) : 1; }();
  const double A = (double)p.xbytes + (double)p.x2bytes, W = (double)p.wbytes;
  if (!on || p.par || p.gm < 2 || p.gn < 2 || W <= 4.0e6) return 1;   // a weight matrix that fits the 4 MiB L2 is re-walked for free
  int best_s = 1;
  double best = 1e30;
  for (int s = 1; s <= 8; s *= 2) {
    if (s > 1 && A * s / 8.0 > 2.0e6) break;
    const double t = s * A + 8.0 * W / s;
    if (t < best) { best = t; best_s = s; }
  }
  const int g = (p.gm * best_s + 7) / 8;
  return g < 1 ? 1 : g;
}
// tile of list position `bid` under that order
__device__ __forceinline__ void hd_conv_tile_of(const ConvP& p, int bid, int& tile_m, int& tile_n) {
  if (p.tgroup <= 1) {
    tile_m = bid / p.gn;
    tile_n = bid - tile_m * p.gn;
  } else {
    const int width = p.tgroup * p.gn;
    const int gid = bid / width;
    const int first_m = gid * p.tgroup;
    const int gsz = p.gm - first_m < p.tgroup ? p.gm - first_m : p.tgroup;
    const int r = bid - gid * width;
    tile_n = r / gsz;
    tile_m = first_m + r - tile_n * gsz;
  }
}

void hd_conv_launch_bk32(ConvP& p, int bm, int bn, bool deep, hipStream_t s);
void hd_conv_launch_bk64(ConvP& p, int bm, int bn, bool deep, hipStream_t s);
// multi-problem forms: every problem single-source, no parity classes, the same (per-lane tap) addressing mode; false = not launched
bool hd_conv_launch_bk32_multi(ConvMulti& mp, int bm, int bn, bool deep, hipStream_t s);
bool hd_conv_launch_bk64_multi(ConvMulti& mp, int bm, int bn, bool deep, hipStream_t s);
// conv3x3_w8.hip: 8-wave family with LDS-staged input patches (3x3 / s1 / p1, Cin % 64 == 0); cfg = tile id
bool hd_conv_p8_eligible(const ConvP& p);
bool hd_conv_p8_pool2_ok(const ConvP& p);
int hd_conv_p8_tiles(const ConvP& p, int cfg);
void hd_conv_launch_p8(ConvP& p, int cfg, hipStream_t s);
// conv3x3_m160.hip: the same family on th x 40-pixel x 64-channel tiles, th in {4, 8} (v_mfma_f32_16x16x32_f16): 256 blocks for the 12-GFLOP
// U-Net layers; hd_conv_m160_tiles = BatchNorm partial-sum rows (one per 160 pixels)
bool hd_conv_m160_eligible(const ConvP& p);
bool hd_conv_m160_pool2_ok(const ConvP& p);
int hd_conv_m160_tiles(const ConvP& p, int th, int tw);
void hd_conv_launch_m160(ConvP& p, int th, int tw, hipStream_t s);      // (th, tw) in {(4, 40), (8, 40), (4, 24)}
void hd_conv_launch_m96_multi(ConvMulti& mp, hipStream_t s);              // several 4 x 24-pixel-tile problems in one grid
// the same tile grid with the blocks of an 8-wave weight gradient behind it (one launch; conv3x3_w8.hip)
void hd_conv_launch_p8_wgrad(ConvP& p, int cfg, const hd_wgrad_args* wa, hipStream_t s);
// wgrad.hip: would hd_wgrad run this weight gradient in the 8-wave patch-staged kernel?
bool hd_wgrad_takes_w8(const hd_wgrad_args* a);
// conv3x3_small.hip: 3x3 / stride 1 / pad 1, Cin in {8,16,32}, Cout in {16,32}, plain NHWC f16 output (+ BN partial sums)
bool hd_conv_small_eligible(const ConvP& p);
bool hd_conv_small_pool2_ok(const ConvP& p);
int hd_conv_small_tiles(const ConvP& p);
void hd_conv_launch_small(ConvP& p, hipStream_t s);
// conv3x3_c64.hip: 3x3 / stride 1 / pad 1, 64 -> 64 channels, weights resident in registers, persistent blocks; rows = partial-sum rows
bool hd_conv_c64_eligible(const ConvP& p);
int hd_conv_c64_rows(const ConvP& p);
// conv3x3_c32to128.hip: 3x3 / stride 1 / pad 1, 32 -> 128 channels, plain output (decoder block 3's data gradient), weights in registers
bool hd_conv_c32to128_eligible(const ConvP& p);
void hd_conv_launch_c32to128(ConvP& p, hipStream_t s);
// conv3x3_cat128to32.hip: 3x3 / stride 1 / pad 1 over cat([nearest_2x(x), x2]), 64 + 64 -> 32 channels (decoder block 3's first conv)
bool hd_conv_cat128to32_eligible(const ConvP& p);
int hd_conv_cat128to32_rows(const ConvP& p);
void hd_conv_launch_cat128to32(ConvP& p, hipStream_t s);
// conv7x7s2_stem.hip: the ResNet stem (7x7 / stride 2 / pad 3, 8 -> 64 channels), weights resident in registers
bool hd_conv_stem_eligible(const ConvP& p);
int hd_conv_stem_rows(const ConvP& p);
void hd_conv_launch_stem(ConvP& p, hipStream_t s);
void hd_conv_launch_c64(ConvP& p, hipStream_t s);
