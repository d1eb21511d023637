// Implicit-GEMM convolution on CDNA4 matrix cores (v_mfma_f32_32x32x16_f16).
//
// GEMM view:  M = N*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin.
// NHWC activations make a K-slice of 8 channels at one tap a single 16-byte
// load; K is walked in 16-byte "chunks" q = tap*(Cin/8) + c8 so any Cin that is
// a multiple of 8 works and a 32-deep K tile may straddle taps.
// The A gather also implements, for free:
//   * nearest-2x upsample + channel concat of two sources (U-Net decoder,
//     reference src/segmentation_models/decoders/unet/decoder.py:38-41),
//   * zero-dilated input (data-gradient of a stride-2 convolution).
//
// Structure (v3):
//   * operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging VGPRs, no ds_write (the
//     VGPR->LDS write path, ~80 B/clk/CU, was the LDS bottleneck of the register-staged v2), three LDS stages so
//     two 64-deep K tiles are in flight behind the MFMAs, counted s_waitcnt vmcnt(N) + one raw s_barrier per K tile;
//   * K tile = 64: every 1-KiB DMA piece is 8 rows x one full 128-byte line (measured with the 32-deep tile: a
//     piece of 16 rows x 64 B costs ~175 cycles of the issuing wave -- 700 cycles of DMA issue per 256 cycles of
//     MFMA; half-line requests were the texture-address bottleneck), and one barrier now covers 16 MFMAs per wave;
//   * every read is a raw BUFFER access: im2col padding, ragged M/N/K tails are an out-of-range offset, which
//     the hardware turns into zeros written to LDS -- the load path has no branch;
//   * the DMA writes LDS lane-linearly (wave base + lane*16 B), so tiles are unpadded [row][64 B] and the bank
//     swizzle is applied on the SOURCE side: the lane that fills 16-B slot s of row r fetches logical chunk
//     s ^ ((r>>1)&7); fragment reads apply the same XOR -> every ds_read_b128 lane group hits 16 distinct slots;
//   * block = 256 threads = 4 waves; tile BM x BN x 32 with BM in {128,64}, BN in {128,64,32};
//   * blockIdx is remapped so that the M tiles an XCD works on are contiguous (neighbouring pixel tiles share
//     their 3x3 halo and all N tiles of one M tile share the gathered pixels in that XCD's L2).
#include "hd_common.h"
#include "conv_params.h"

namespace {

constexpr int BK = 64;
constexpr int LDS_ROW = 64;  // halves per LDS row (128 bytes = one cache line, unpadded: LDS-DMA writes lane-linearly)
constexpr int CPT = BK / 8;  // 16-byte chunks per row per K tile
constexpr unsigned OOB = 0xFFFFFFF0u;


typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, f16* lds_dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds_dst, 16, voff, 0, 0, 0);
}

template <int BM, int BN, int WM, int WN, bool DUAL, bool KGEN, int NSTAGE>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvP p) {
  constexpr int MT = BM / (WM * 32);
  constexpr int NT = BN / (WN * 32);
  constexpr int A_LOADS = BM * CPT / 256;               // 4 or 2
  constexpr int BROWS = BN < 64 ? 64 : BN;               // B region rows (every wave issues the same number of DMAs)
  constexpr int B_LOADS = BROWS * CPT / 256;             // 4,2,2
  constexpr int STAGE = (BM + BROWS) * LDS_ROW;          // halves per stage
  constexpr int L_TILE = A_LOADS + B_LOADS;              // DMA instructions per wave per K tile
  // the epilogue reuses the pipeline stages as an fp32 [BM][BN] tile: size the array for whichever is larger
  constexpr int LDS_HALVES = (NSTAGE * STAGE > BM * BN * 2) ? NSTAGE * STAGE : BM * BN * 2;
  __shared__ __attribute__((aligned(1024))) f16 lds[LDS_HALVES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- XCD-aware tile mapping: blocks are dealt round-robin over the 8 XCDs, so give XCD x the x-th contiguous
  //      eighth of the (n-tile fastest) tile list.  Bijective for any grid size.
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
    bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  }
  const int tile_m = bid / p.gn, tile_n = bid - tile_m * p.gn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int j = (tid & 7) ^ ((tid >> 4) & 7);   // logical chunk this lane fetches into slot tid&7 of row tid>>3
  const int HoWo = p.Ho * p.Wo;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.x), 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(DUAL ? p.x2 : p.x), 0, DUAL ? p.x2bytes : p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.w), 0, p.wbytes, 0x00020000);

  // ---- per-thread A rows
  int hb[A_LOADS], wb[A_LOADS];
  unsigned nb1[A_LOADS], nb2[A_LOADS];  // image base offsets (bytes) in x / x2
  bool rvalid[A_LOADS];
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    int pix = m0 + (tid >> 3) + i * 32;
    rvalid[i] = pix < p.M;
    int pp = rvalid[i] ? pix : 0;
    int n = pp / HoWo;
    int rem = pp - n * HoWo;
    int ho = rem / p.Wo;
    int wo = rem - ho * p.Wo;
    hb[i] = ho * p.stride - p.pad;
    wb[i] = wo * p.stride - p.pad;
    nb1[i] = (unsigned)n * (unsigned)(p.Hsrc * p.Wsrc) * (unsigned)p.C1 * 2u;
    nb2[i] = DUAL ? (unsigned)n * (unsigned)(p.Hin * p.Win) * (unsigned)p.C2 * 2u : 0u;
  }
  // ---- per-thread B rows
  unsigned wbase[B_LOADS];
  bool wvalid[B_LOADS];
#pragma unroll
  for (int i = 0; i < B_LOADS; ++i) {
    int brow = (tid >> 3) + i * 32;
    int co = n0 + brow;
    wvalid[i] = (brow < BN) && (co < p.Cout);   // rows >= BN (BN=32) fetch zeros
    wbase[i] = (unsigned)(wvalid[i] ? co : 0) * (unsigned)p.Ktot * 2u;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const bool dil2 = p.in_dil == 2;
  const bool up1 = p.up1 != 0;

  // K walk: chunk q = kt*8 + j.  Fast path (Cin % 64 == 0): the tap is uniform over the block, so the per-row pixel
  // offset / validity is recomputed only when the tap changes (every Cin/64 tiles); between changes a load address is
  // one add.  Generic path (Cin in {8,16,24,...}: stem, last decoder block, head): per-lane tap, full recompute.
  int kt_issue = 0;
  int kh_u = 0, kw_u = 0, c8_u = 0;
  unsigned po1[A_LOADS], po2[A_LOADS];   // byte offset of (pixel at the current tap, channel 0) in x / x2
  bool pv[A_LOADS];

  auto pixel_state = [&](int kh, int kw, int i, unsigned& o1, unsigned& o2, bool& v) {
    int hi = hb[i] + kh, wi = wb[i] + kw;
    v = rvalid[i];
    int hs, ws;
    if (dil2) {
      v = v && (hi >= 0) && (wi >= 0) && (((hi | wi) & 1) == 0);
      hs = hi >> 1;
      ws = wi >> 1;
      v = v && (hs < p.Hsrc) && (ws < p.Wsrc);
    } else {
      v = v && ((unsigned)hi < (unsigned)p.Hin) && ((unsigned)wi < (unsigned)p.Win);
      hs = up1 ? (hi >> 1) : hi;
      ws = up1 ? (wi >> 1) : wi;
    }
    o1 = nb1[i] + (unsigned)((hs * p.Wsrc + ws) * p.C1) * 2u;
    o2 = DUAL ? nb2[i] + (unsigned)((hi * p.Win + wi) * p.C2) * 2u : 0u;
  };
  if (!KGEN) {
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) pixel_state(0, 0, i, po1[i], po2[i], pv[i]);
  }

  auto gload = [&](int stage) {
    f16* sa = lds + stage * STAGE + wave * (8 * LDS_ROW);
    f16* sb = sa + BM * LDS_ROW;
    const int q = kt_issue * CPT + j;
    const bool kvalid = q < p.nchunks;
    int c;
    if (KGEN) {
      const int tap = (int)(((float)q + 0.5f) * p.inv_cin8);
      c = (q - tap * p.cin8) * 8;
      const int kh = (int)(((float)tap + 0.5f) * p.inv_kw);
      const int kw = tap - kh * p.KW;
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) pixel_state(kh, kw, i, po1[i], po2[i], pv[i]);
    } else {
      c = (c8_u + j) * 8;
    }
    if (DUAL && c8_u * 8 >= p.C1) {   // uniform: a K tile never straddles the concat boundary (C1 % 32 == 0)
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i)
        dma16(rx2, sa + i * (32 * LDS_ROW), (pv[i] && kvalid) ? po2[i] + (unsigned)(c - p.C1) * 2u : OOB);
    } else {
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i)
        dma16(rx, sa + i * (32 * LDS_ROW), (pv[i] && kvalid) ? po1[i] + (unsigned)c * 2u : OOB);
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) dma16(rw, sb + i * (32 * LDS_ROW), (wvalid[i] && kvalid) ? wbase[i] + (unsigned)q * 16u : OOB);
    // advance
    ++kt_issue;
    if (!KGEN) {
      c8_u += CPT;
      if (c8_u >= p.cin8) {      // uniform branch, no loads inside
        c8_u = 0;
        if (++kw_u == p.KW) {
          kw_u = 0;
          ++kh_u;
        }
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) pixel_state(kh_u, kw_u, i, po1[i], po2[i], pv[i]);
      }
    }
  };
  const int frow = lane & 31;
  const int fh = lane >> 5;
  const int swz = (frow >> 1) & 7;
  auto compute = [&](int stage) {
    const f16* sa = lds + stage * STAGE;
    const f16* sb = sa + BM * LDS_ROW;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int slot = ((ks * 2 + fh) ^ swz) * 8;
      f16x8 af[MT], bf[NT];
#pragma unroll
      for (int a = 0; a < MT; ++a)
        af[a] = *reinterpret_cast<const f16x8*>(sa + (wm * MT * 32 + a * 32 + frow) * LDS_ROW + slot);
#pragma unroll
      for (int b = 0; b < NT; ++b)
        bf[b] = *reinterpret_cast<const f16x8*>(sb + (wn * NT * 32 + b * 32 + frow) * LDS_ROW + slot);
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
  };

  // prologue: NSTAGE-1 tiles in flight
#pragma unroll
  for (int t = 0; t < NSTAGE - 1; ++t) gload(t);
  int rd = 0, wr = NSTAGE - 1;
  for (int kt = 0; kt < p.nk; ++kt) {
    // this wave's DMAs of tile kt have landed once at most (NSTAGE-2) tiles' worth remain outstanding
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * L_TILE) : "memory");
    __builtin_amdgcn_s_barrier();   // every wave's part of tile kt is in LDS; stage `wr` (read at kt-1) is free
    __builtin_amdgcn_sched_barrier(0);
    gload(wr);                      // tile kt+NSTAGE-1 (zeros beyond the last tile: out-of-range offsets)
    compute(rd);
    rd = (rd + 1 == NSTAGE) ? 0 : rd + 1;
    wr = (wr + 1 == NSTAGE) ? 0 : wr + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---------------- epilogue ----------------
  // Vector path (NHWC f16 output, Cout % 8 == 0): accumulators -> LDS fp32 tile -> each thread owns 8 consecutive
  // output channels of a pixel: 16-byte residual / mask loads and 16-byte coalesced stores (the per-lane 2-byte
  // stores of the direct path cost 4-5x the HBM time on the 16/32-channel decoder layers).
  if (p.out_mode == HD_OUT_NHWC_F16 && (p.Cout & 7) == 0) {
    float* ct = reinterpret_cast<float*>(lds);   // [BM][BN] fp32
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * MT * 32 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int col = wn * NT * 32 + b * 32 + (lane & 31);
          ct[row * BN + col] = acc[a][b][r];
        }
    __syncthreads();
    constexpr int CPR = BN / 8;            // 8-channel chunks per tile row
    constexpr int RPI = 256 / CPR;         // rows covered per iteration
    const int cch = tid % CPR, r0 = tid / CPR;
    const int co = n0 + cch * 8;
    const bool cvalid = co < p.Cout;       // Cout % 8 == 0 => whole chunk valid
    float bias8[8], ssum8[8], ssq8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      bias8[k] = (p.bias && cvalid) ? p.bias[co + k] : 0.f;
      ssum8[k] = ssq8[k] = 0.f;
    }
    if (cvalid) {
#pragma unroll 4
      for (int row = r0; row < BM; row += RPI) {
        const int pix = m0 + row;
        if (pix >= p.M) break;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(ct + row * BN + cch * 8);
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(ct + row * BN + cch * 8 + 4);
        float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
        const size_t off = (size_t)pix * p.Cout + co;
        if (p.res) {
          const f16x8 rv = *reinterpret_cast<const f16x8*>(p.res + off);
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] += (float)rv[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += bias8[k];
        if (p.mask) {
          const f16x8 mv = *reinterpret_cast<const f16x8*>(p.mask + off);
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (!((float)mv[k] > 0.f)) v[k] = 0.f;
        }
        f16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (p.stats) {
            float vr = (float)(f16)v[k];
            ssum8[k] += vr;
            ssq8[k] += vr * vr;
          }
          float w = v[k];
          if (p.act == HD_ACT_RELU) w = fmaxf(w, 0.f);
          else if (p.act == HD_ACT_SIGMOID) w = 1.f / (1.f + __expf(-w));
          o[k] = (f16)w;
        }
        *reinterpret_cast<f16x8*>(reinterpret_cast<f16*>(p.y) + off) = o;
      }
    }
    if (p.stats) {
      __syncthreads();                     // everyone is done reading the C tile
      float* red = reinterpret_cast<float*>(lds);   // [RPI][BN][2]
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        red[(r0 * BN + cch * 8 + k) * 2 + 0] = ssum8[k];
        red[(r0 * BN + cch * 8 + k) * 2 + 1] = ssq8[k];
      }
      __syncthreads();
      if (tid < BN && n0 + tid < p.Cout) {
        float s = 0.f, s2 = 0.f;
        for (int m = 0; m < RPI; ++m) {
          s += red[(m * BN + tid) * 2 + 0];
          s2 += red[(m * BN + tid) * 2 + 1];
        }
        p.stats[((size_t)tile_m * 2 + 0) * p.Cout + n0 + tid] = s;
        p.stats[((size_t)tile_m * 2 + 1) * p.Cout + n0 + tid] = s2;
      }
    }
    return;
  }

  // Direct path (NCHW fp32 output or ragged channel counts: head / RPN / predictor outputs, all tiny)
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
          float v = acc[a][b][r];
          if (p.res) v += (float)p.res[(size_t)pix * p.Cout + co];
          v += bias;
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
        p.stats[((size_t)tile_m * 2 + 0) * p.Cout + co] = s;
        p.stats[((size_t)tile_m * 2 + 1) * p.Cout + co] = s2;
      }
    }
  }
}


}  // namespace

template <int BM, int BN, int WM, int WN, int NS>
static void launch_variant_bk64(ConvP& p, hipStream_t s) {
  p.gm = hd_cdiv(p.M, BM);
  p.gn = hd_cdiv(p.Cout, BN);
  dim3 grid(p.gm * p.gn);
  const bool dual = p.x2 != nullptr;
  const bool kgen = (p.cin8 % (64 / 8)) != 0;
  if (dual) {
    if (kgen) return;
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, true, false, NS>), grid, dim3(256), 0, s, p);
  } else {
    if (kgen) hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, false, true, NS>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, false, false, NS>), grid, dim3(256), 0, s, p);
  }
}

// deep = more LDS stages (fewer co-resident blocks, more K in flight)
void hd_conv_launch_bk64(ConvP& p, int bm, int bn, bool deep, hipStream_t s) {
  p.nk = (p.nchunks + 64 / 8 - 1) / (64 / 8);
  if (bn == 32) { launch_variant_bk64<128, 32, 4, 1, 2>(p, s); return; }
  if (bm == 128) {
    if (bn == 128) { if (deep) launch_variant_bk64<128, 128, 2, 2, 3>(p, s); else launch_variant_bk64<128, 128, 2, 2, 2>(p, s); }
    else { if (deep) launch_variant_bk64<128, 64, 2, 2, 3>(p, s); else launch_variant_bk64<128, 64, 2, 2, 2>(p, s); }
  } else {
    if (bn == 128) { if (deep) launch_variant_bk64<64, 128, 2, 2, 3>(p, s); else launch_variant_bk64<64, 128, 2, 2, 2>(p, s); }
    else { if (deep) launch_variant_bk64<64, 64, 2, 2, 3>(p, s); else launch_variant_bk64<64, 64, 2, 2, 2>(p, s); }
  }
}
