// Weight-gradient implicit GEMM on MFMA:  dW[co][kk] = sum_pix dY[pix][co] * Xg[pix][kk]
// (kk = (kh*KW+kw)*Cin + ci walks the same 16-byte chunk order as conv_igemm).
//
// Both operands live in memory as [pixel][channel] (NHWC), i.e. the reduction index
// is the SLOW axis.  Tiles are staged to LDS in that natural layout and the MFMA
// fragments (8 consecutive reduction elements per lane) are produced by the gfx950
// transposing LDS read ds_read_b64_tr_b16 -- no software transpose anywhere.
// Row strides are odd multiples of 64 B so the four k-rows a 32-lane half touches
// fall in distinct bank quarters.
//
// The pixel range is split over gridDim.z; each slice writes its fp32 partial tile to
// a slab and hd_wgrad_reduce sums the slices in a fixed order (deterministic, no
// float atomics) while converting to the OIHW fp32 layout of the master gradient.
#include <stdlib.h>

#include "hd_common.h"

namespace {

constexpr int BP = 32;    // pixels per reduction tile
constexpr int TN = 128;   // kk columns per block

struct WgP {
  const f16* x;
  const f16* x2;
  const f16* dy;
  float* slab;
  int N, Hsrc, Wsrc, Hin, Win, C1, C2, Cin, Ho, Wo, Cout, KH, KW, stride, pad, up1;
  int M, cin8, nchunks, Ktot, per_split;
  unsigned xbytes, x2bytes, dybytes;
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ f16x8 tr_frag(const f16* row0_ptr, int row_stride) {
  // two transposed 4x16 block reads: k rows [0,4) and [4,8) relative to row0_ptr
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(row0_ptr));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(row0_ptr + 4 * row_stride));
  f16x4 fa = __builtin_bit_cast(f16x4, a), fb = __builtin_bit_cast(f16x4, b);
  f16x8 r = {fa[0], fa[1], fa[2], fa[3], fb[0], fb[1], fb[2], fb[3]};
  return r;
}

template <int TM, int WM, int WN>
__global__ __launch_bounds__(256) void wgrad_kernel(WgP p) {
  constexpr int MT = TM / (WM * 32);
  constexpr int NT = TN / (WN * 32);
  constexpr int RSA = (TM == 128) ? 160 : (TM == 64 ? 96 : 32);  // halves; bytes = odd multiple of 64
  constexpr int RSB = 160;
  constexpr int A_CH = TM / 8;                       // chunks per pixel row of dY tile
  constexpr int B_CH = TN / 8;                       // 16
  constexpr int A_LOADS = (BP * A_CH + 255) / 256;   // 2,1,1
  constexpr int B_LOADS = BP * B_CH / 256;           // 2
  constexpr int STAGE = BP * RSA + BP * RSB;
  __shared__ __attribute__((aligned(16))) f16 lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int co0 = blockIdx.y * TM;
  const int kk0 = blockIdx.x * TN;  // first kk column of this block
  const int HoWo = p.Ho * p.Wo;
  const int pbeg = blockIdx.z * p.per_split;
  const int pend = min(p.M, pbeg + p.per_split);
  const int nk = (pend - pbeg + BP - 1) / BP;

  // ---- B (gathered input) assignment: fixed chunk column, 2 pixel rows
  const int bcc = tid % B_CH;
  const int bpr = tid / B_CH;  // 0..15
  const int bq = kk0 / 8 + bcc;
  const bool bq_valid = bq < p.nchunks;
  int btap = bq_valid ? bq / p.cin8 : 0;
  const int bc = (bq_valid ? bq - btap * p.cin8 : 0) * 8;
  const int bkh = btap / p.KW, bkw = btap - bkh * p.KW;
  // ---- A (dY) assignment
  const int acc_ = tid % A_CH;
  const int apr = tid / A_CH;
  const bool a_active = apr < BP;  // TM=32: 4 chunks/pixel -> 64 rows of threads, only 32 used per load
  const bool aco_valid = (co0 + acc_ * 8) < p.Cout;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  constexpr int A_ROWS_PER_LOAD = 256 / A_CH;  // 16, 32, 64
  // every global read is a raw buffer load; ragged pixel / channel tails and im2col padding are out-of-range offsets
  // (hardware zero fill) -- no branch around a load, so hipcc keeps counted vmcnt waits and two tiles stay in flight
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.dy), 0, p.dybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.x), 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.x2 ? p.x2 : p.x), 0, p.x2 ? p.x2bytes : p.xbytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFF0u;
  const bool first_src = bc < p.C1;

  // incremental pixel walk for the B (gather) rows: pixel index advances by BP per tile
  int bn_[B_LOADS], bho[B_LOADS], bwo[B_LOADS];
#pragma unroll
  for (int i = 0; i < B_LOADS; ++i) {
    int pix = pbeg + bpr + i * 16;
    int pp = pix < p.M ? pix : 0;
    bn_[i] = pp / HoWo;
    int rem = pp - bn_[i] * HoWo;
    bho[i] = rem / p.Wo;
    bwo[i] = rem - bho[i] * p.Wo;
  }
  int kt_issue = 0;

  auto gload = [&](u32x4 (&ra)[A_LOADS], u32x4 (&rb)[B_LOADS]) {
    const int pb = pbeg + kt_issue * BP;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      int prow = apr + i * A_ROWS_PER_LOAD;
      int pix = pb + prow;
      bool v = (prow < BP) && (pix < pend) && aco_valid;
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rdy, v ? (unsigned)(((size_t)pix * p.Cout + co0 + acc_ * 8) * 2) : OOB, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
      int pix = pb + bpr + i * 16;
      int hi = bho[i] * p.stride - p.pad + bkh, wi = bwo[i] * p.stride - p.pad + bkw;
      bool v = (pix < pend) && bq_valid && ((unsigned)hi < (unsigned)p.Hin) && ((unsigned)wi < (unsigned)p.Win);
      int hs = p.up1 ? (hi >> 1) : hi, ws = p.up1 ? (wi >> 1) : wi;
      unsigned o1 = (unsigned)((((size_t)bn_[i] * p.Hsrc + hs) * p.Wsrc + ws) * p.C1 + bc) * 2u;
      unsigned o2 = (unsigned)((((size_t)bn_[i] * p.Hin + hi) * p.Win + wi) * p.C2 + (bc - p.C1)) * 2u;
      u32x4 a1 = __builtin_amdgcn_raw_buffer_load_b128(rx, (v && first_src) ? o1 : OOB, 0, 0);
      if (p.x2) {
        u32x4 a2 = __builtin_amdgcn_raw_buffer_load_b128(rx2, (v && !first_src) ? o2 : OOB, 0, 0);
        a1 = a1 | a2;
      }
      rb[i] = a1;
      // advance this row's pixel by BP (BP <= Wo is not guaranteed: loop)
      bwo[i] += BP;
      while (bwo[i] >= p.Wo) {
        bwo[i] -= p.Wo;
        if (++bho[i] == p.Ho) {
          bho[i] = 0;
          ++bn_[i];
        }
      }
    }
    ++kt_issue;
  };
  auto lstore = [&](int buf, const u32x4 (&ra)[A_LOADS], const u32x4 (&rb)[B_LOADS]) {
    f16* sa = lds + buf * STAGE;
    f16* sb = sa + BP * RSA;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      int prow = apr + i * A_ROWS_PER_LOAD;
      if (prow < BP) *reinterpret_cast<u32x4*>(sa + prow * RSA + acc_ * 8) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) *reinterpret_cast<u32x4*>(sb + (bpr + i * 16) * RSB + bcc * 8) = rb[i];
  };
  (void)a_active;

  // transposed-read lane geometry (see header comment): 16-lane group g, lane-in-group li
  const int li = lane & 15, g = lane >> 4;
  const int tq = li >> 2, tp = li & 3;
  const int th = g >> 1, thalf = g & 1;
  const int krow = 8 * th + tq;                 // + ks*16 (+4 for the second read)
  const int coff = 16 * thalf + 4 * tp;         // + tile column base

  auto compute = [&](int buf) {
    const f16* sa = lds + buf * STAGE;
    const f16* sb = sa + BP * RSA;
    // both K sub-steps' fragments are requested before the first MFMA (two register sets, pinned by the scheduling barriers: the
    // compiler otherwise re-uses one set and waits for the LDS in the middle of every sub-step; see conv_igemm_bk64.hip)
    f16x8 af[2][MT], bf[2][NT];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int a = 0; a < MT; ++a) af[ks][a] = tr_frag(sa + (ks * 16 + krow) * RSA + (wm * MT * 32 + a * 32) + coff, RSA);
#pragma unroll
      for (int b = 0; b < NT; ++b) bf[ks][b] = tr_frag(sb + (ks * 16 + krow) * RSB + (wn * NT * 32 + b * 32) + coff, RSB);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks][a], bf[ks][b], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // tiles beyond nk load zeros (pix >= pend): the loop runs an even number of tiles
  u32x4 ra0[A_LOADS], rb0[B_LOADS], ra1[A_LOADS], rb1[B_LOADS];
  gload(ra0, rb0);
  gload(ra1, rb1);
  lstore(0, ra0, rb0);
  __syncthreads();
  const int npair = (nk + 1) >> 1;
  for (int it = 0; it < npair; ++it) {
    gload(ra0, rb0);
    compute(0);
    lstore(1, ra1, rb1);
    __syncthreads();
    gload(ra1, rb1);
    compute(1);
    lstore(0, ra0, rb0);
    __syncthreads();
  }

  // ---- epilogue: fp32 partial tile -> slab[z][co][kk]
  float* out = p.slab + (size_t)blockIdx.z * p.Cout * p.Ktot;
#pragma unroll
  for (int b = 0; b < NT; ++b) {
    const int kk = kk0 + wn * NT * 32 + b * 32 + (lane & 31);
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * MT * 32 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (co < p.Cout && kk < p.Ktot) out[(size_t)co * p.Ktot + kk] = acc[a][b][r];
      }
  }
}

// ---- slab reduction.  Geometry of one weight tensor (also one entry of hd_wgrad_reduce_multi's table):
struct RedG {
  const float* __restrict__ slab;
  float* __restrict__ dw;
  int nsplit, Cout, KH, KW, Cin, Cin_real, accumulate;
  float scale;
  int64_t Ktot, total4;
  size_t sstride;
};

__device__ __forceinline__ RedG make_redg(const float* slab, float* dw, int nsplit, int Cout_slab, int Cout, int KH, int KW, int Cin, int Cin_real,
                                          float scale, int accumulate) {
  RedG g;
  g.slab = slab; g.dw = dw; g.nsplit = nsplit; g.Cout = Cout; g.KH = KH; g.KW = KW; g.Cin = Cin; g.Cin_real = Cin_real;
  g.accumulate = accumulate; g.scale = scale;
  g.Ktot = (int64_t)KH * KW * Cin;
  g.total4 = (int64_t)Cout * g.Ktot / 4;
  g.sstride = (size_t)Cout_slab * g.Ktot;
  return g;
}

// the four sums of quad i4 (slab elements (co, kk..kk+3); kk = t*Cin + ci, Cin % 8 == 0 so the four share a tap) -> OIHW
__device__ __forceinline__ void red_store(const RedG& g, int64_t i4, f32x4 acc) {
  const int64_t i = i4 * 4;
  const int taps = g.KH * g.KW;
  const int co = (int)(i / g.Ktot);
  const int kk = (int)(i - (int64_t)co * g.Ktot);
  const int t = kk / g.Cin;
  const int ci = kk - t * g.Cin;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (ci + u >= g.Cin_real) continue;
    const size_t o = ((size_t)co * g.Cin_real + ci + u) * taps + t;
    const float v = acc[u] * g.scale;
    g.dw[o] = g.accumulate ? g.dw[o] + v : v;
  }
}

// one thread per quad: 16-byte coalesced slab reads (the bulk of the traffic: nsplit x |W|), four independent accumulators,
// strided OIHW writes (|W| once)
__device__ __forceinline__ void red_plain(const RedG& g, int64_t i4) {
  const float* s = g.slab + i4 * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 4 <= g.nsplit; k += 4) {
    f32x4 a0 = *reinterpret_cast<const f32x4*>(s + (size_t)(k + 0) * g.sstride);
    f32x4 a1 = *reinterpret_cast<const f32x4*>(s + (size_t)(k + 1) * g.sstride);
    f32x4 a2 = *reinterpret_cast<const f32x4*>(s + (size_t)(k + 2) * g.sstride);
    f32x4 a3 = *reinterpret_cast<const f32x4*>(s + (size_t)(k + 3) * g.sstride);
    acc += (a0 + a1) + (a2 + a3);
  }
  for (; k < g.nsplit; ++k) acc += *reinterpret_cast<const f32x4*>(s + (size_t)k * g.sstride);
  red_store(g, i4, acc);
}

// SMALL weight tensors with MANY splits (the 16/32-channel full-resolution layers: 2 304 - 9 216 weights, 128 - 768 pixel
// splits): one WAVE per quad, lane l sums the splits l, l+64, ..., then a shuffle tree (fixed order: deterministic).  The
// one-thread-per-quad form walks all the splits in one dependent chain there: 45 - 67 us per launch for a few KB of output.
__device__ __forceinline__ void red_wave(const RedG& g, int64_t i4, int lane) {
  const float* s = g.slab + i4 * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = lane; k < g.nsplit; k += 64) acc += *reinterpret_cast<const f32x4*>(s + (size_t)k * g.sstride);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] += __shfl_xor(acc[u], d);
  }
  if (lane == 0) red_store(g, i4, acc);
}

// Split-parallel form for weight tensors whose one-thread-per-quad grid is small while the split chains are long (the 64- and
// 128-channel 3x3 layers behind the 8-wave weight gradient: 9 216 / 36 864 quads x 256 / 64 splits): a group of G waves owns 64
// CONSECUTIVE quads (every wave-load is 1 KiB of one slice, contiguous), wave w of the group walks the slices w, w+G, ..., four
// loads in flight; the G partials are added through LDS in wave order (deterministic).  `part` = the group's [G][64] LDS rows;
// contains a __syncthreads(): every thread of the block must call it.
template <int G>
__device__ __forceinline__ void red_split(const RedG& g, int64_t i4, bool live, int w, int lane, f32x4 (*part)[64]) {
  const float* s = g.slab + (live ? i4 : 0) * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int k = w;
  if (live) {
    for (; k + 3 * G < g.nsplit; k += 4 * G) {
      f32x4 a0 = *reinterpret_cast<const f32x4*>(s + (size_t)(k + 0 * G) * g.sstride);
      f32x4 a1 = *reinterpret_cast<const f32x4*>(s + (size_t)(k + 1 * G) * g.sstride);
      f32x4 a2 = *reinterpret_cast<const f32x4*>(s + (size_t)(k + 2 * G) * g.sstride);
      f32x4 a3 = *reinterpret_cast<const f32x4*>(s + (size_t)(k + 3 * G) * g.sstride);
      acc += (a0 + a1) + (a2 + a3);
    }
    for (; k < g.nsplit; k += G) acc += *reinterpret_cast<const f32x4*>(s + (size_t)k * g.sstride);
  }
  part[w][lane] = acc;
  __syncthreads();
  if (w != 0 || !live) return;
#pragma unroll
  for (int q = 1; q < G; ++q) acc += part[q][lane];
  red_store(g, i4, acc);
}

// ROW form for the large tensors with FEW splits (layer3 / layer4 / decoder 3x3 layers: 2.4 - 9.4 MB of weights, 4 - 40 slices): the
// one-thread-per-quad form stores every sum as four dwords 36 B apart (the (tap, ci) -> OIHW transpose), 2.4 M partial-line writes per
// tensor -- measured 2.7 TB/s on a launch of five such tensors while the split form streams at 4.9.  Here a block owns R whole output
// rows (co): the sums (same order as red_plain: bit-identical) go to LDS in slab order, and the block writes its rows of the OIHW tensor
// as one contiguous run; reads stay whole 1 KiB runs of a slice per wave.  (Measured slower than this: 64-channel x 9-tap units -- 1008
// of 1024 threads busy instead of 576 - 1024, but 256-byte read pieces; and issuing both trips of a 1 152-quad row at once.)  `lds`
// holds R * Ktot floats (<= 8192);
// every thread of the block must call it (contains a __syncthreads()).
__device__ __forceinline__ f32x4 red_quad(const RedG& g, const float* s) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 4 <= g.nsplit; k += 4) {
    f32x4 a0 = *reinterpret_cast<const f32x4*>(s + (size_t)(k + 0) * g.sstride);
    f32x4 a1 = *reinterpret_cast<const f32x4*>(s + (size_t)(k + 1) * g.sstride);
    f32x4 a2 = *reinterpret_cast<const f32x4*>(s + (size_t)(k + 2) * g.sstride);
    f32x4 a3 = *reinterpret_cast<const f32x4*>(s + (size_t)(k + 3) * g.sstride);
    acc += (a0 + a1) + (a2 + a3);
  }
  for (; k < g.nsplit; ++k) acc += *reinterpret_cast<const f32x4*>(s + (size_t)k * g.sstride);
  return acc;
}

__device__ __forceinline__ void red_rows(const RedG& g, int row0, int R, float* lds) {
  const int Q = (int)(g.Ktot / 4);
  const int K = (int)g.Ktot;
  const int nq = min(R, g.Cout - row0) * Q;                 // quads of this block (rows are consecutive in the slab)
  const float* base = g.slab + (size_t)row0 * K;
  f32x4* l4 = reinterpret_cast<f32x4*>(lds);
  for (int q = threadIdx.x; q < nq; q += blockDim.x) l4[q] = red_quad(g, base + (size_t)q * 4);
  __syncthreads();
  const int taps = g.KH * g.KW;
  for (int j = threadIdx.x; j < nq * 4; j += blockDim.x) {
    const int r = j / K;
    const int jj = j - r * K;
    const int ci = jj / taps;
    const int t = jj - ci * taps;
    const float v = lds[r * K + t * g.Cin + ci] * g.scale;
    const size_t o = (size_t)(row0 + r) * K + jj;
    g.dw[o] = g.accumulate ? g.dw[o] + v : v;
  }
}

// rows per block of the row form (0: the tensor does not take it)
inline int red_rows_per_block(int KH, int KW, int Cin, int Cin_real, int nsplit) {
  const int64_t Q = (int64_t)KH * KW * Cin / 4;
  if (Cin_real != Cin || KH * KW < 2 || Q < 128 || Q > 2048 || nsplit > 64) return 0;
  // quads a block takes: 1024 = one trip of its threads, 2048 = what the LDS tile holds (experiment knob; the plan hands R to the kernel)
  static const int cap = getenv("HD_RED_ROWS_QUADS") ? atoi(getenv("HD_RED_ROWS_QUADS")) : 1024;
  return Q >= 1024 ? 1 : (int)((cap < 1024 ? 1024 : cap > 2048 ? 2048 : cap) / Q);
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nsplit, int Cout_slab, int Cout,
                                    int KH, int KW, int Cin, int Cin_real, float scale, int accumulate) {
  const RedG g = make_redg(slab, dw, nsplit, Cout_slab, Cout, KH, KW, Cin, Cin_real, scale, accumulate);
  for (int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i4 < g.total4; i4 += (int64_t)gridDim.x * blockDim.x) red_plain(g, i4);
}

__global__ __launch_bounds__(256) void wgrad_reduce_wave_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nsplit, int Cout_slab,
                                                                int Cout, int KH, int KW, int Cin, int Cin_real, float scale, int accumulate) {
  const RedG g = make_redg(slab, dw, nsplit, Cout_slab, Cout, KH, KW, Cin, Cin_real, scale, accumulate);
  const int64_t i4 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i4 < g.total4) red_wave(g, i4, threadIdx.x & 63);
}

template <int G>
__global__ __launch_bounds__(64 * G) void wgrad_reduce_split_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nsplit, int Cout_slab,
                                                                    int Cout, int KH, int KW, int Cin, int Cin_real, float scale, int accumulate) {
  __shared__ f32x4 part[G][64];
  const RedG g = make_redg(slab, dw, nsplit, Cout_slab, Cout, KH, KW, Cin, Cin_real, scale, accumulate);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t i4 = (int64_t)blockIdx.x * 64 + lane;
  red_split<G>(g, i4, i4 < g.total4, w, lane, part);
}

// which form hd_wgrad_reduce picks for a tensor: 0 plain, 1 wave, 4 / 8 / 16 = split with that many waves per 64 quads
__host__ __device__ inline int red_mode(int64_t total4, int nsplit, bool split_on) {
  if (split_on && total4 >= 4096 && total4 <= 65536 && nsplit >= 16) {
    const int64_t blocks = (total4 + 63) / 64;
    return (nsplit >= 128 && blocks <= 256) ? 16 : (nsplit >= 32 && blocks <= 1024) ? 8 : 4;
  }
  return (total4 <= 16384 && nsplit >= 64) ? 1 : 0;
}

// Every weight tensor of a backward segment in ONE launch (hd_wgrad_reduce_multi): blocks [first_block, first_block + blocks) of
// the 1-D grid belong to table entry e and run the SAME form, in the same summation order, as hd_wgrad_reduce would for that tensor
// alone (bit-identical results).  1024 threads: 1024 quads (plain), 16 quads (wave) or 16 / G groups of 64 quads (split) per block.
// 47 separate reductions per training step were 0.48 ms, much of it the ~4.5 us floor of a launch that moves a few KB.
struct WredTab {
  hd_wred_desc d[HD_WRED_MAX];       // by value in the kernel arguments (1 KB): no table in device memory, nothing to keep alive
};

__global__ __launch_bounds__(1024) void wgrad_reduce_multi_kernel(const WredTab tab, int n) {
  __shared__ f32x4 lds4[2048];                     // split forms: [16][64] partials; row form: R * Ktot sums
  f32x4 (*part)[64] = reinterpret_cast<f32x4 (*)[64]>(lds4);
  int e = 0;
  for (int i = 1; i < n; ++i)
    if ((int)blockIdx.x >= tab.d[i].first_block) e = i;
  const hd_wred_desc& d = tab.d[e];
  const RedG g = make_redg(d.slab, d.dw_oihw, d.nsplit, d.Cout_slab, d.Cout, d.KH, d.KW, d.Cin, d.Cin_real, d.scale, d.accumulate);
  const int64_t b = (int64_t)blockIdx.x - d.first_block;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (d.mode == 2) {
    const int R = d.reserved;                     // rows per block, chosen by hd_wgrad_reduce_plan
    red_rows(g, (int)b * R, R, reinterpret_cast<float*>(lds4));
  } else if (d.mode == 0) {
    const int64_t i4 = b * 1024 + threadIdx.x;
    if (i4 < g.total4) red_plain(g, i4);
  } else if (d.mode == 1) {
    const int64_t i4 = b * 16 + wave;
    if (i4 < g.total4) red_wave(g, i4, lane);
  } else if (d.mode == 16) {
    const int64_t i4 = b * 64 + lane;
    red_split<16>(g, i4, i4 < g.total4, wave, lane, part);
  } else if (d.mode == 8) {
    const int64_t i4 = (b * 2 + (wave >> 3)) * 64 + lane;
    red_split<8>(g, i4, i4 < g.total4, wave & 7, lane, part + (wave >> 3) * 8);
  } else {
    const int64_t i4 = (b * 4 + (wave >> 2)) * 64 + lane;
    red_split<4>(g, i4, i4 < g.total4, wave & 3, lane, part + (wave >> 2) * 4);
  }
}

__global__ void weight_prep_kernel(const float* __restrict__ w, const float* __restrict__ oscale, f16* __restrict__ wf,
                                   f16* __restrict__ wd, int Cout, int Cin, int KH, int KW, int Cin_pad, int Cout_pad) {
  const int taps = KH * KW;
  // forward layout [Cout][tap][Cin_pad]
  if (wf) {
    const int64_t total = (int64_t)Cout * taps * Cin_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      int ci = (int)(i % Cin_pad);
      int t = (int)((i / Cin_pad) % taps);
      int co = (int)(i / ((int64_t)Cin_pad * taps));
      float v = 0.f;
      if (ci < Cin) {
        v = w[((size_t)co * Cin + ci) * taps + t];
        if (oscale) v *= oscale[co];
      }
      wf[i] = (f16)v;
    }
  }
  // data-gradient layout [Cin_pad][flipped tap][Cout_pad]
  if (wd) {
    const int64_t total = (int64_t)Cin_pad * taps * Cout_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      int co = (int)(i % Cout_pad);
      int t = (int)((i / Cout_pad) % taps);
      int ci = (int)(i / ((int64_t)Cout_pad * taps));
      float v = 0.f;
      if (ci < Cin && co < Cout) {
        v = w[((size_t)co * Cin + ci) * taps + (taps - 1 - t)];
        if (oscale) v *= oscale[co];
      }
      wd[i] = (f16)v;
    }
  }
}

// all layers of a network in ONE launch: blockIdx.y = layer (descriptor table in device memory), blockIdx.x strides over
// that layer's 32(co) x 32(ci) x taps tiles.  A tile is read in the source's own order (32 runs of 32*taps contiguous
// floats), parked in LDS and written out twice, each time with the destination's fastest index across the lanes: the first
// version read the OIHW tensor with a stride of taps (forward layout) / Cin*taps (data-gradient layout) floats between
// lanes -- 64 cache lines per wave-load -- and took 187 us per step for 24.4 M weights (33 us of HBM traffic).
__global__ __launch_bounds__(256) void weight_prep_multi_kernel(const hd_wprep_desc* __restrict__ tab) {
  const hd_wprep_desc d = tab[blockIdx.y];
  const float* __restrict__ w = d.w_oihw;
  f16* __restrict__ wf = (f16*)d.w_fwd;
  f16* __restrict__ wd = (f16*)d.w_dgrad;
  const int taps = d.KH * d.KW;
  if (taps > 9) {                            // 7x7 stem: a few thousand weights, element-wise form
    if (wf) {
      const int64_t total = (int64_t)d.Cout * taps * d.Cin_pad;
      for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int ci = (int)(i % d.Cin_pad);
        int t = (int)((i / d.Cin_pad) % taps);
        int co = (int)(i / ((int64_t)d.Cin_pad * taps));
        float v = 0.f;
        if (ci < d.Cin) v = w[((size_t)co * d.Cin + ci) * taps + t];
        wf[i] = (f16)v;
      }
    }
    if (wd) {
      const int64_t total = (int64_t)d.Cin_pad * taps * d.Cout_pad;
      for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int co = (int)(i % d.Cout_pad);
        int t = (int)((i / d.Cout_pad) % taps);
        int ci = (int)(i / ((int64_t)d.Cout_pad * taps));
        float v = 0.f;
        if (ci < d.Cin && co < d.Cout) v = w[((size_t)co * d.Cin + ci) * taps + (taps - 1 - t)];
        wd[i] = (f16)v;
      }
    }
    return;
  }
  constexpr int T = 32;
  __shared__ float tile[T][T * 9 + 1];
  const int co_ext = wd ? (d.Cout_pad > d.Cout ? d.Cout_pad : d.Cout) : d.Cout;
  const int nco = (co_ext + T - 1) / T, nci = (d.Cin_pad + T - 1) / T;
  const int run = T * taps;                  // floats per source run
  for (int tix = blockIdx.x; tix < nco * nci; tix += gridDim.x) {
    const int co0 = (tix / nci) * T, ci0 = (tix % nci) * T;
    __syncthreads();                         // previous tile fully written out
    // 16-byte accesses throughout: a VMEM instruction costs the issuing wave ~150 cycles whatever its width
    const bool vec_src = (d.Cin % 4) == 0;   // runs start 16-byte aligned (ci0 % 32 == 0)
    for (int e4 = threadIdx.x; e4 < T * run / 4; e4 += 256) {
      const int r = e4 / (run / 4), j = (e4 - r * (run / 4)) * 4;
      const int co = co0 + r;
      const float* src = w + ((size_t)co * d.Cin + ci0) * taps + j;
      if (vec_src && co < d.Cout && ci0 + (j + 3) / taps < d.Cin) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src);
        tile[r][j] = v[0]; tile[r][j + 1] = v[1]; tile[r][j + 2] = v[2]; tile[r][j + 3] = v[3];
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) tile[r][j + q] = (co < d.Cout && ci0 + (j + q) / taps < d.Cin) ? src[q] : 0.f;
      }
    }
    __syncthreads();
    if (wf) {                                // [co][t][ci_pad]: 8 consecutive ci per lane
      for (int e = threadIdx.x; e < T * taps * (T / 8); e += 256) {
        const int c8 = e % (T / 8), t = (e / (T / 8)) % taps, r = e / ((T / 8) * taps);
        const int co = co0 + r, ci = ci0 + c8 * 8;
        if (co < d.Cout && ci < d.Cin_pad) {
          f16x8 o;
#pragma unroll
          for (int q = 0; q < 8; ++q) o[q] = (f16)tile[r][(c8 * 8 + q) * taps + t];
          *reinterpret_cast<f16x8*>(wf + ((size_t)co * taps + t) * d.Cin_pad + ci) = o;
        }
      }
    }
    if (wd) {                                // [ci_pad][flipped t][co_pad]: 8 consecutive co per lane
      for (int e = threadIdx.x; e < T * taps * (T / 8); e += 256) {
        const int r8 = e % (T / 8), t = (e / (T / 8)) % taps, ci_l = e / ((T / 8) * taps);
        const int co = co0 + r8 * 8, ci = ci0 + ci_l;
        if (co < d.Cout_pad && ci < d.Cin_pad) {
          f16x8 o;
#pragma unroll
          for (int q = 0; q < 8; ++q) o[q] = (f16)tile[r8 * 8 + q][ci_l * taps + t];
          *reinterpret_cast<f16x8*>(wd + ((size_t)ci * taps + (taps - 1 - t)) * d.Cout_pad + co) = o;
        }
      }
    }
  }
}

}  // namespace

extern "C" int hd_weight_prep_multi(const hd_wprep_desc* table_dev, int n_layers, int blocks_per_layer, void* stream) {
  HD_CHECK_ARG(table_dev && n_layers > 0 && blocks_per_layer > 0 && blocks_per_layer <= 65535, "hd_weight_prep_multi: bad args");
  hipLaunchKernelGGL(weight_prep_multi_kernel, dim3(blocks_per_layer, n_layers), dim3(256), 0, (hipStream_t)stream, table_dev);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

static int g_wg_tm = -1;   // tuning hook (tools/tune_wgrad.py): force the Cout tile of hd_wgrad; -1 = by channel count
extern "C" int hd_wgrad_tune_override(int tm) {
  HD_CHECK_ARG(tm == -1 || tm == 32 || tm == 64 || tm == 128, "hd_wgrad_tune_override: tm in {32, 64, 128} or -1");
  g_wg_tm = tm;
  return HD_OK;
}

// wgrad3x3_small.hip: 3x3 / s1 / p1, Cin in {16,32}, Cout <= 32 (HD_WGRAD_SMALL=0 or a tile override keeps the general kernel)
bool hd_wgrad_small_eligible(const hd_wgrad_args* a);
void hd_wgrad_small_launch(const hd_wgrad_args* a, hipStream_t s);
// wgrad3x3_w8.hip: 3x3 / s1 / p1, >= 64 channels in and out: 8-wave patch-staged kernel (HD_WGRAD_W8=0 keeps the general kernel)
bool hd_wgrad_w8_eligible(const hd_wgrad_args* a);
void hd_wgrad_w8_launch(const hd_wgrad_args* a, hipStream_t s);
extern "C" int hd_wgrad_w8_blocks(const hd_wgrad_args* a) {
  if (!a) return 0;
  static const char* env = getenv("HD_WGRAD_W8");
  if ((env && env[0] == '0') || !hd_wgrad_w8_eligible(a)) return 0;
  return ((a->C1 + a->C2) / 64) * (a->Cout / 64);
}

extern "C" int hd_wgrad_direct_ok(const hd_wgrad_args* a) {
  if (!a || a->nsplit != 1 || a->in_scale) return 0;
  static const char* env8 = getenv("HD_WGRAD_W8");
  static const bool w8_on = !(env8 && env8[0] == '0');
  return (w8_on && g_wg_tm < 0 && !hd_wgrad_small_eligible(a) && hd_wgrad_w8_eligible(a)) ? 1 : 0;
}

bool hd_wgrad_takes_w8(const hd_wgrad_args* a) {
  static const char* env8 = getenv("HD_WGRAD_W8");
  static const bool w8_on = !(env8 && env8[0] == '0');
  if (!a || !a->x || !a->dy || !(a->slab || (a->dw_oihw && a->nsplit == 1)) || a->nsplit < 1 || a->in_scale) return false;
  if (g_wg_tm >= 0 || hd_wgrad_small_eligible(a)) return false;
  return w8_on && hd_wgrad_w8_eligible(a);
}

extern "C" int hd_wgrad(const hd_wgrad_args* a, void* stream) {
  HD_CHECK_ARG(a && a->x && a->dy && (a->slab || a->dw_oihw), "hd_wgrad: null pointer");
  HD_CHECK_ARG(!a->dw_oihw || hd_wgrad_direct_ok(a), "hd_wgrad: dw_oihw (direct output) needs nsplit == 1 and the 8-wave 3x3 kernel (hd_wgrad_direct_ok)");
  HD_CHECK_ARG(a->C1 > 0 && a->C1 % 8 == 0 && a->C2 % 8 == 0 && a->Cout % 8 == 0, "hd_wgrad: channels must be multiples of 8");
  HD_CHECK_ARG((a->C2 == 0) == (a->x2 == nullptr), "hd_wgrad: x2/C2 mismatch");
  HD_CHECK_ARG(a->nsplit >= 1, "hd_wgrad: nsplit");
  HD_CHECK_ARG((a->in_scale == nullptr) == (a->in_shift == nullptr), "hd_wgrad: in_scale / in_shift must be given together");
  {
    static const char* env = getenv("HD_WGRAD_SMALL");
    static const bool small_on = !(env && env[0] == '0');
    if ((small_on || a->in_scale) && g_wg_tm < 0 && hd_wgrad_small_eligible(a)) {
      hd_wgrad_small_launch(a, (hipStream_t)stream);
      HD_CHECK_LAUNCH();
      return HD_OK;
    }
  }
  HD_CHECK_ARG(!a->in_scale, "hd_wgrad: consumer-side BatchNorm (in_scale / in_shift) is implemented by the small-channel 3x3 kernel only "
                             "(3x3 / stride 1 / pad 1, one source, C1 in {16,32}, Cout <= 32)");
  {
    static const char* env8 = getenv("HD_WGRAD_W8");
    static const bool w8_on = !(env8 && env8[0] == '0');
    if (w8_on && g_wg_tm < 0 && hd_wgrad_w8_eligible(a)) {
      hd_wgrad_w8_launch(a, (hipStream_t)stream);
      HD_CHECK_LAUNCH();
      return HD_OK;
    }
  }
  WgP p;
  p.x = (const f16*)a->x; p.x2 = (const f16*)a->x2; p.dy = (const f16*)a->dy; p.slab = a->slab;
  p.N = a->N; p.Hsrc = a->Hsrc; p.Wsrc = a->Wsrc; p.Hin = a->Hin; p.Win = a->Win; p.C1 = a->C1; p.C2 = a->C2;
  p.Cin = a->C1 + a->C2; p.Ho = a->Ho; p.Wo = a->Wo; p.Cout = a->Cout; p.KH = a->KH; p.KW = a->KW;
  p.stride = a->stride; p.pad = a->pad; p.up1 = a->up1;
  p.M = a->N * a->Ho * a->Wo;
  p.cin8 = p.Cin / 8;
  p.nchunks = a->KH * a->KW * p.cin8;
  p.Ktot = a->KH * a->KW * p.Cin;
  int per = hd_cdiv(p.M, a->nsplit);
  per = hd_cdiv(per, BP) * BP;
  p.per_split = per;
  {
    int64_t xb = (int64_t)a->N * a->Hsrc * a->Wsrc * a->C1 * 2, x2b = a->x2 ? (int64_t)a->N * a->Hin * a->Win * a->C2 * 2 : 0;
    int64_t db = (int64_t)p.M * a->Cout * 2;
    HD_CHECK_ARG(xb < 0xFFFFFFF0ll && x2b < 0xFFFFFFF0ll && db < 0xFFFFFFF0ll, "hd_wgrad: tensor larger than 4 GiB (buffer addressing)");
    p.xbytes = (unsigned)xb; p.x2bytes = (unsigned)x2b; p.dybytes = (unsigned)db;
  }
  hipStream_t s = (hipStream_t)stream;
  const int tm = g_wg_tm > 0 ? g_wg_tm : (p.Cout > 64 ? 128 : (p.Cout > 32 ? 64 : 32));
  dim3 grid(hd_cdiv(p.Ktot, TN), hd_cdiv(p.Cout, tm), a->nsplit);
  if (tm == 128) hipLaunchKernelGGL((wgrad_kernel<128, 2, 2>), grid, dim3(256), 0, s, p);
  else if (tm == 64) hipLaunchKernelGGL((wgrad_kernel<64, 2, 2>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((wgrad_kernel<32, 1, 4>), grid, dim3(256), 0, s, p);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

// n independent weight gradients; ONE grid when every entry runs in the 8-wave patch-staged kernel (wgrad3x3_w8.hip), else n launches
void hd_wgrad_w8_launch_multi(const hd_wgrad_args* a, int n, hipStream_t s);
extern "C" int hd_wgrad_multi(const hd_wgrad_args* args, int n, void* stream) {
  HD_CHECK_ARG(args && n > 0, "hd_wgrad_multi: bad args");
  static const char* env = getenv("HD_WGRAD_MULTI");
  bool one = !(env && env[0] == '0') && n >= 2 && n <= HD_WGRAD_MULTI_MAX;
  for (int i = 0; i < n; ++i)
    HD_CHECK_ARG(!args[i].dw_oihw || hd_wgrad_direct_ok(args + i), "hd_wgrad_multi: dw_oihw (direct output) needs nsplit == 1 and the 8-wave 3x3 kernel");
  for (int i = 0; one && i < n; ++i) one = hd_wgrad_takes_w8(args + i);
  if (!one) {
    for (int i = 0; i < n; ++i) {
      const int rc = hd_wgrad(args + i, stream);
      if (rc) return rc;
    }
    return HD_OK;
  }
  for (int i = 0; i < n; ++i) {
    const hd_wgrad_args* a = args + i;
    HD_CHECK_ARG(a->C1 > 0 && a->C1 % 8 == 0 && a->C2 % 8 == 0 && a->Cout % 8 == 0 && (a->C2 == 0) == (a->x2 == nullptr) && a->nsplit >= 1,
                 "hd_wgrad_multi: bad entry");
  }
  hd_wgrad_w8_launch_multi(args, n, (hipStream_t)stream);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

static bool red_split_on() {
  static const int on = getenv("HD_WGRAD_REDUCE_SPLIT") ? atoi(getenv("HD_WGRAD_REDUCE_SPLIT")) : 1;
  return on != 0;
}

static bool red_rows_on() {
  static const int on = getenv("HD_WGRAD_REDUCE_ROWS") ? atoi(getenv("HD_WGRAD_REDUCE_ROWS")) : 1;
  return on != 0;
}

extern "C" int hd_wgrad_reduce(const float* slab, float* dw_oihw, int nsplit, int Cout_slab, int Cout, int KH, int KW, int Cin,
                               int Cin_real, float scale, int accumulate, void* stream) {
  HD_CHECK_ARG(slab && dw_oihw && nsplit >= 1 && Cout <= Cout_slab && Cin_real <= Cin, "hd_wgrad_reduce: bad args");
  int64_t total = (int64_t)Cout * Cin * KH * KW / 4;
  int g = (int)((total + 255) / 256);
  if (g > 8192) g = 8192;
  hipStream_t st = (hipStream_t)stream;
  const int mode = red_mode(total, nsplit, red_split_on());
  if (mode >= 4) {
    const int blocks = (int)((total + 63) / 64);
    if (mode == 16)
      hipLaunchKernelGGL((wgrad_reduce_split_kernel<16>), dim3(blocks), dim3(1024), 0, st, slab, dw_oihw, nsplit, Cout_slab, Cout, KH, KW, Cin, Cin_real, scale, accumulate);
    else if (mode == 8)
      hipLaunchKernelGGL((wgrad_reduce_split_kernel<8>), dim3(blocks), dim3(512), 0, st, slab, dw_oihw, nsplit, Cout_slab, Cout, KH, KW, Cin, Cin_real, scale, accumulate);
    else
      hipLaunchKernelGGL((wgrad_reduce_split_kernel<4>), dim3(blocks), dim3(256), 0, st, slab, dw_oihw, nsplit, Cout_slab, Cout, KH, KW, Cin, Cin_real, scale, accumulate);
  } else if (mode == 1) {      // few outputs, long split chains: one wave per quad
    hipLaunchKernelGGL(wgrad_reduce_wave_kernel, dim3((int)((total + 3) / 4)), dim3(256), 0, st, slab, dw_oihw, nsplit, Cout_slab,
                       Cout, KH, KW, Cin, Cin_real, scale, accumulate);
  } else {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(g), dim3(256), 0, st, slab, dw_oihw, nsplit, Cout_slab, Cout, KH, KW, Cin, Cin_real, scale, accumulate);
  }
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_wgrad_reduce_plan(hd_wred_desc* table_host, int n) {
  if (!table_host || n <= 0 || n > HD_WRED_MAX) { hd_set_error("hd_wgrad_reduce_plan: bad args"); return HD_E_ARG; }
  int64_t first = 0;
  for (int i = 0; i < n; ++i) {
    hd_wred_desc& d = table_host[i];
    if (!(d.slab && d.dw_oihw && d.nsplit >= 1 && d.Cout <= d.Cout_slab && d.Cin_real <= d.Cin && d.Cin % 8 == 0)) {
      hd_set_error("hd_wgrad_reduce_plan: bad entry");
      return HD_E_ARG;
    }
    const int64_t total = (int64_t)d.Cout * d.Cin * d.KH * d.KW / 4;
    d.mode = red_mode(total, d.nsplit, red_split_on());
    const int R = d.mode == 0 && red_rows_on() ? red_rows_per_block(d.KH, d.KW, d.Cin, d.Cin_real, d.nsplit) : 0;
    d.first_block = (int32_t)first;
    if (R > 0) {                       // row form: R whole output rows per block, contiguous OIHW writes
      d.mode = 2;
      d.reserved = R;
      first += (d.Cout + R - 1) / R;
    } else {
      const int64_t per = d.mode == 0 ? 1024 : d.mode == 1 ? 16 : 64 * (16 / d.mode);
      first += (total + per - 1) / per;
    }
    if (first > 0x7fffffff) { hd_set_error("hd_wgrad_reduce_plan: grid too large"); return HD_E_ARG; }
  }
  return (int)first;
}

extern "C" int hd_wgrad_reduce_multi(const hd_wred_desc* table_host, int n, int total_blocks, void* stream) {
  HD_CHECK_ARG(table_host && n > 0 && n <= HD_WRED_MAX && total_blocks > 0, "hd_wgrad_reduce_multi: bad args");
  WredTab tab;
  for (int i = 0; i < n; ++i) tab.d[i] = table_host[i];
  for (int i = n; i < HD_WRED_MAX; ++i) tab.d[i] = table_host[n - 1];
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(total_blocks), dim3(1024), 0, (hipStream_t)stream, tab, n);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_weight_prep(const float* w_oihw, const float* out_scale, void* w_fwd, void* w_dgrad, int Cout, int Cin, int KH,
                              int KW, int Cin_pad, int Cout_pad, void* stream) {
  HD_CHECK_ARG(w_oihw && (w_fwd || w_dgrad) && Cin_pad >= Cin && Cout_pad >= Cout && Cin_pad % 8 == 0 && Cout_pad % 8 == 0,
               "hd_weight_prep: bad args");
  int64_t total = (int64_t)Cout_pad * Cin_pad * KH * KW;
  int g = (int)((total + 255) / 256);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(weight_prep_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, w_oihw, out_scale, (f16*)w_fwd, (f16*)w_dgrad,
                     Cout, Cin, KH, KW, Cin_pad, Cout_pad);
  HD_CHECK_LAUNCH();
  return HD_OK;
}
