// Implicit-GEMM convolution on CDNA4 matrix cores (v_mfma_f32_32x32x16_f16).
//
// GEMM view:  M = N*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin.
// NHWC activations make a K-slice of 8 channels at one tap a single 16-byte
// load; K is walked in 16-byte "chunks" q = tap*(Cin/8) + c8 so any Cin that is
// a multiple of 8 works and a 32-deep K tile may straddle taps.
// The A gather also implements, for free:
//   * nearest-2x upsample + channel concat of two sources (U-Net decoder,
//     reference src/segmentation_models/decoders/unet/decoder.py:38-41),
//   * zero-dilated input (data-gradient of a strided convolution).
// Block = 256 threads = 4 waves; tile 128 x BN x 32; LDS rows padded to 80 B so
// every ds_read_b128 lane group hits 16 distinct 16-B slots (conflict free).
#include "hd_common.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDS_ROW = 40;  // halves per LDS row: 32 data + 8 pad  (80 bytes)

struct ConvP {
  const f16* x;
  const f16* x2;
  const f16* w;
  const float* bias;
  const f16* res;
  const f16* mask;
  void* y;
  float* stats;
  int N, Hsrc, Wsrc, Hin, Win, C1, C2, Cin, Ho, Wo, Cout, KH, KW, stride, pad, up1, in_dil, act, out_mode;
  int M, cin8, nchunks, nk, Ktot;
};

struct RowState {
  int n, hb, wb;
  bool valid;
};

__device__ __forceinline__ u32x4 load_a_chunk(const ConvP& p, const RowState& r, int kh, int kw, int c8, bool kvalid) {
  u32x4 v = {0u, 0u, 0u, 0u};
  if (!(r.valid && kvalid)) return v;
  int hi = r.hb + kh, wi = r.wb + kw;
  if (p.in_dil > 1) {
    if (hi < 0 || wi < 0) return v;
    int d = p.in_dil;
    int hq = hi / d, wq = wi / d;
    if (hq * d != hi || wq * d != wi || hq >= p.Hsrc || wq >= p.Wsrc) return v;
    size_t off = ((size_t)(r.n * p.Hsrc + hq) * p.Wsrc + wq) * p.C1 + c8 * 8;
    return *reinterpret_cast<const u32x4*>(p.x + off);
  }
  if ((unsigned)hi >= (unsigned)p.Hin || (unsigned)wi >= (unsigned)p.Win) return v;
  int c = c8 * 8;
  if (c < p.C1) {
    if (p.up1) {
      hi >>= 1;
      wi >>= 1;
    }
    size_t off = ((size_t)(r.n * p.Hsrc + hi) * p.Wsrc + wi) * p.C1 + c;
    return *reinterpret_cast<const u32x4*>(p.x + off);
  } else {
    size_t off = ((size_t)(r.n * p.Hin + hi) * p.Win + wi) * p.C2 + (c - p.C1);
    return *reinterpret_cast<const u32x4*>(p.x2 + off);
  }
}

template <int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvP p) {
  constexpr int MT = BM / (WM * 32);
  constexpr int NT = BN / (WN * 32);
  constexpr int A_LOADS = BM * 4 / 256;                 // 2
  constexpr int B_LOADS = (BN * 4 + 255) / 256;          // 2,1,1
  constexpr int STAGE = (BM + BN) * LDS_ROW;             // halves per stage
  __shared__ __attribute__((aligned(16))) f16 lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int j = tid & 3;
  const int HoWo = p.Ho * p.Wo;

  RowState rows[A_LOADS];
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    int pix = m0 + (tid >> 2) + i * 64;
    rows[i].valid = pix < p.M;
    int pp = rows[i].valid ? pix : 0;
    int n = pp / HoWo;
    int rem = pp - n * HoWo;
    int ho = rem / p.Wo;
    int wo = rem - ho * p.Wo;
    rows[i].n = n;
    rows[i].hb = ho * p.stride - p.pad;
    rows[i].wb = wo * p.stride - p.pad;
  }
  // B rows
  const f16* wrow[B_LOADS];
  bool wvalid[B_LOADS];
#pragma unroll
  for (int i = 0; i < B_LOADS; ++i) {
    int brow = (tid >> 2) + i * 64;
    int co = n0 + brow;
    wvalid[i] = (brow < BN) && (co < p.Cout);
    wrow[i] = p.w + (size_t)(wvalid[i] ? co : 0) * p.Ktot;
  }

  // K walk state for this thread's chunk column j
  int q = j;
  int tap = q / p.cin8;
  int c8 = q - tap * p.cin8;
  int kh = tap / p.KW;
  int kw = tap - kh * p.KW;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  u32x4 ra[A_LOADS], rb[B_LOADS];

  auto gload = [&]() {
    bool kvalid = q < p.nchunks;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) ra[i] = load_a_chunk(p, rows[i], kh, kw, c8, kvalid);
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (wvalid[i] && kvalid) v = *reinterpret_cast<const u32x4*>(wrow[i] + (size_t)q * 8);
      rb[i] = v;
    }
    // advance to next K tile
    q += 4;
    c8 += 4;
    while (c8 >= p.cin8) {
      c8 -= p.cin8;
      if (++kw == p.KW) {
        kw = 0;
        ++kh;
      }
    }
  };
  auto lstore = [&](int buf) {
    f16* sa = lds + buf * STAGE;
    f16* sb = sa + BM * LDS_ROW;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i)
      *reinterpret_cast<u32x4*>(sa + ((tid >> 2) + i * 64) * LDS_ROW + j * 8) = ra[i];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
      int brow = (tid >> 2) + i * 64;
      if (brow < BN) *reinterpret_cast<u32x4*>(sb + brow * LDS_ROW + j * 8) = rb[i];
    }
  };

  gload();
  lstore(0);
  __syncthreads();

  const int frow = lane & 31;
  const int fk = (lane >> 5) * 8;
  for (int kt = 0; kt < p.nk; ++kt) {
    const int buf = kt & 1;
    const bool more = (kt + 1) < p.nk;
    if (more) gload();
    const f16* sa = lds + buf * STAGE;
    const f16* sb = sa + BM * LDS_ROW;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8 af[MT], bf[NT];
#pragma unroll
      for (int a = 0; a < MT; ++a)
        af[a] = *reinterpret_cast<const f16x8*>(sa + (wm * MT * 32 + a * 32 + frow) * LDS_ROW + ks * 16 + fk);
#pragma unroll
      for (int b = 0; b < NT; ++b)
        bf[b] = *reinterpret_cast<const f16x8*>(sb + (wn * NT * 32 + b * 32 + frow) * LDS_ROW + ks * 16 + fk);
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
    if (more) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  float ssum[NT], ssq[NT];
#pragma unroll
  for (int b = 0; b < NT; ++b) ssum[b] = ssq[b] = 0.f;

#pragma unroll
  for (int b = 0; b < NT; ++b) {
    const int col = wn * NT * 32 + b * 32 + (lane & 31);
    const int co = n0 + col;
    const bool cvalid = co < p.Cout;
    const float bias = (p.bias && cvalid) ? p.bias[co] : 0.f;
#pragma unroll
    for (int a = 0; a < MT; ++a) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * MT * 32 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int pix = m0 + row;
        if (pix < p.M && cvalid) {
          float v = acc[a][b][r] + bias;
          if (p.res) v += (float)p.res[(size_t)pix * p.Cout + co];
          if (p.mask && !((float)p.mask[(size_t)pix * p.Cout + co] > 0.f)) v = 0.f;
          if (p.stats) {
            float vr = (float)(f16)v;
            ssum[b] += vr;
            ssq[b] += vr * vr;
          }
          if (p.act == HD_ACT_RELU) v = fmaxf(v, 0.f);
          else if (p.act == HD_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
          if (p.out_mode == HD_OUT_NHWC_F16) {
            reinterpret_cast<f16*>(p.y)[(size_t)pix * p.Cout + co] = (f16)v;
          } else {
            int n = pix / HoWo;
            int rem = pix - n * HoWo;
            reinterpret_cast<float*>(p.y)[((size_t)n * p.Cout + co) * HoWo + rem] = v;
          }
        }
      }
    }
  }

  if (p.stats) {
    // reduce the two lane halves, then across the WM waves that share columns
    float* red = reinterpret_cast<float*>(lds);  // [WM][BN][2]
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      float s = ssum[b] + __shfl_xor(ssum[b], 32);
      float s2 = ssq[b] + __shfl_xor(ssq[b], 32);
      if (lane < 32) {
        int col = wn * NT * 32 + b * 32 + lane;
        red[(wm * BN + col) * 2 + 0] = s;
        red[(wm * BN + col) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    if (tid < BN) {
      int co = n0 + tid;
      if (co < p.Cout) {
        float s = 0.f, s2 = 0.f;
#pragma unroll
        for (int m = 0; m < WM; ++m) {
          s += red[(m * BN + tid) * 2 + 0];
          s2 += red[(m * BN + tid) * 2 + 1];
        }
        p.stats[((size_t)blockIdx.x * 2 + 0) * p.Cout + co] = s;
        p.stats[((size_t)blockIdx.x * 2 + 1) * p.Cout + co] = s2;
      }
    }
  }
}

int pick_bn(int Cout) { return Cout > 64 ? 128 : (Cout > 32 ? 64 : 32); }

int fill_params(const hd_conv_args* a, ConvP& p) {
  HD_CHECK_ARG(a && a->x && a->w && a->y, "hd_conv2d: null pointer");
  HD_CHECK_ARG(a->C1 > 0 && a->C1 % 8 == 0 && a->C2 >= 0 && a->C2 % 8 == 0, "hd_conv2d: channel counts must be multiples of 8 (C1=%d C2=%d)", a->C1, a->C2);
  HD_CHECK_ARG((a->C2 == 0) == (a->x2 == nullptr), "hd_conv2d: x2/C2 mismatch");
  HD_CHECK_ARG(a->N > 0 && a->Ho > 0 && a->Wo > 0 && a->Cout > 0 && a->KH > 0 && a->KW > 0 && a->stride > 0, "hd_conv2d: bad extent");
  HD_CHECK_ARG(!(a->in_dil > 1 && (a->up1 || a->C2)), "hd_conv2d: in_dil excludes up1/x2");
  HD_CHECK_ARG(!a->up1 || (a->Hin == 2 * a->Hsrc && a->Win == 2 * a->Wsrc), "hd_conv2d: up1 needs Hin=2*Hsrc");
  HD_CHECK_ARG((int64_t)a->N * a->Ho * a->Wo < (1ll << 31), "hd_conv2d: too many pixels");
  p.x = (const f16*)a->x;
  p.x2 = (const f16*)a->x2;
  p.w = (const f16*)a->w;
  p.bias = a->bias;
  p.res = (const f16*)a->res;
  p.mask = (const f16*)a->mask;
  p.y = a->y;
  p.stats = a->stats;
  p.N = a->N; p.Hsrc = a->Hsrc; p.Wsrc = a->Wsrc; p.Hin = a->Hin; p.Win = a->Win;
  p.C1 = a->C1; p.C2 = a->C2; p.Cin = a->C1 + a->C2;
  p.Ho = a->Ho; p.Wo = a->Wo; p.Cout = a->Cout; p.KH = a->KH; p.KW = a->KW;
  p.stride = a->stride; p.pad = a->pad; p.up1 = a->up1; p.in_dil = a->in_dil < 1 ? 1 : a->in_dil;
  p.act = a->act; p.out_mode = a->out_mode;
  p.M = a->N * a->Ho * a->Wo;
  p.cin8 = p.Cin / 8;
  p.nchunks = a->KH * a->KW * p.cin8;
  p.nk = (p.nchunks + 3) / 4;
  p.Ktot = a->KH * a->KW * p.Cin;
  return HD_OK;
}

}  // namespace

extern "C" int hd_conv2d_stats_rows(const hd_conv_args* a) {
  if (!a) return HD_E_ARG;
  return hd_cdiv((int64_t)a->N * a->Ho * a->Wo, BM);
}

extern "C" int hd_conv2d(const hd_conv_args* a, void* stream) {
  ConvP p;
  int rc = fill_params(a, p);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int bn = pick_bn(p.Cout);
  dim3 grid(hd_cdiv(p.M, BM), hd_cdiv(p.Cout, bn));
  if (bn == 128) hipLaunchKernelGGL((conv_igemm_kernel<128, 2, 2>), grid, dim3(256), 0, s, p);
  else if (bn == 64) hipLaunchKernelGGL((conv_igemm_kernel<64, 2, 2>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((conv_igemm_kernel<32, 4, 1>), grid, dim3(256), 0, s, p);
  HD_CHECK_LAUNCH();
  return HD_OK;
}
