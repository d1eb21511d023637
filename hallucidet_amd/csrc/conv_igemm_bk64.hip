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
#include "conv_epilogue.h"

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
__device__ __forceinline__ void conv_igemm_body(ConvP& p, int bid_in, int nwg_in) {
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
  HD_TRACE(0, wall_clock64());
  HD_TRACE(1, clock64());

  // ---- XCD-aware tile mapping: blocks are dealt round-robin over the 8 XCDs, so give XCD x the x-th contiguous
  //      eighth of the (n-tile fastest) tile list.  Bijective for any grid size.
  int bid = bid_in;
  {
    const int nwg = nwg_in, xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
    bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  }
  int tile_m, tile_n;
  hd_conv_tile_of(p, bid, tile_m, tile_n);
  if (p.par && !hd_par_setup<BM, 8>(p, tile_m)) return;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int j = (tid & 7) ^ ((tid >> 4) & 7);   // logical chunk this lane fetches into slot tid&7 of row tid>>3
  const int HoWo = p.Ho * p.Wo;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.x), 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(DUAL ? p.x2 : p.x), 0, DUAL ? p.x2bytes : p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.w), 0, p.wbytes, 0x00020000);

  // ---- per-thread A rows
  const bool fastdiv = p.N * HoWo < (1 << 24);          // (a parity class has fewer rows than the full output)
  int hb[A_LOADS], wb[A_LOADS];
  unsigned nb1[A_LOADS], nb2[A_LOADS];  // image base offsets (bytes) in x / x2
  bool rvalid[A_LOADS];
  // Row 0 of the thread is decomposed with (reciprocal) divisions, rows 1.. by stepping 32 pixels on: a few compares instead of
  // two more divisions and their quarter-rate integer multiplies per row (the A-row set-up was 1 500 of a block's 4 200 set-up clocks).
  // Coordinates are those of the parity class (ii, jj) when p.par, of the output image otherwise.
  const int rw_ = p.par ? p.Wc : p.Wo, rh_ = p.par ? p.Hc : p.Ho;
  int rn_, ri_, rj_;
  {
    const int pix0 = m0 + (tid >> 3);
    const int pp0 = pix0 < p.M ? pix0 : 0;
    const int hw = rw_ * rh_;
    rn_ = fastdiv ? hd_fdiv(pp0, hw, hd_rcp(hw)) : pp0 / hw;
    const int rem = pp0 - rn_ * hw;
    ri_ = fastdiv ? hd_fdiv(rem, rw_, hd_rcp(rw_)) : rem / rw_;
    rj_ = rem - ri_ * rw_;
  }
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    const int pix = m0 + (tid >> 3) + i * 32;
    rvalid[i] = pix < p.M;
    if (i > 0) {
      rj_ += 32;
      while (rj_ >= rw_) { rj_ -= rw_; ++ri_; }
      while (ri_ >= rh_) { ri_ -= rh_; ++rn_; }
    }
    const int n = rn_;
    const int ho = p.par ? 2 * ri_ + p.ph : ri_;
    const int wo = p.par ? 2 * rj_ + p.pw : rj_;
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
  int kh_u = p.par ? p.t0h : 0, kw_u = p.par ? p.t0w : 0, c8_u = 0;
  const int tap_step = p.par ? 2 : 1;
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
    for (int i = 0; i < A_LOADS; ++i) pixel_state(kh_u, kw_u, i, po1[i], po2[i], pv[i]);
  }

  auto gload = [&](int stage) {
    f16* sa = lds + stage * STAGE + wave * (8 * LDS_ROW);
    f16* sb = sa + BM * LDS_ROW;
    const int q = (!KGEN && p.par) ? (kh_u * p.KW + kw_u) * p.cin8 + c8_u + j : kt_issue * CPT + j;
    const bool kvalid = (!KGEN && p.par) ? kt_issue < p.nk : q < p.nchunks;
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
        kw_u += tap_step;
        if (kw_u >= p.KW) {
          kw_u = p.par ? p.t0w : 0;
          kh_u += tap_step;
        }
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) pixel_state(kh_u, kw_u, i, po1[i], po2[i], pv[i]);
      }
    }
  };
  const int frow = lane & 31;
  const int fh = lane >> 5;
  const int swz = (frow >> 1) & 7;
  // fragments of K sub-step ks + 1 are requested before the MFMAs of sub-step ks are issued (two register sets): left to itself the
  // compiler re-used ONE set and waited for lgkmcnt(0) in the middle of every sub-step
  // Fragments of K sub-step ks + 2 are requested right after the MFMAs of sub-step ks are issued (two register sets), and the first two
  // sub-steps' fragments BEFORE the next tile's DMA pieces (their LDS round trip runs under the DMA issue).  Left to itself the compiler
  // re-used ONE set and waited for lgkmcnt(0) in the middle of every sub-step; written as two sets it folded them back -- the
  // scheduling barriers pin the order.  (One code path: a run-time switch between this and the plain form doubled the register count
  // and halved the occupancy of every variant -- the A/B that missed it compared two slow halves of one binary.)
  f16x8 afd[2][MT], bfd[2][NT];
  auto frag = [&](int stage, int ks, int buf) {
    const f16* sa = lds + stage * STAGE;
    const f16* sb = sa + BM * LDS_ROW;
    const int slot = ((ks * 2 + fh) ^ swz) * 8;
#pragma unroll
    for (int a = 0; a < MT; ++a) afd[buf][a] = *reinterpret_cast<const f16x8*>(sa + (wm * MT * 32 + a * 32 + frow) * LDS_ROW + slot);
#pragma unroll
    for (int b = 0; b < NT; ++b) bfd[buf][b] = *reinterpret_cast<const f16x8*>(sb + (wn * NT * 32 + b * 32 + frow) * LDS_ROW + slot);
  };
  auto compute = [&](int stage) {
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afd[ks & 1][a], bfd[ks & 1][b], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 2 < BK / 16) frag(stage, ks + 2, ks & 1);
    }
  };

  // prologue: NSTAGE-1 tiles in flight
#pragma unroll
  for (int t = 0; t < NSTAGE - 1; ++t) gload(t);
  HD_TRACE(2, clock64());
  int rd = 0, wr = NSTAGE - 1;
  for (int kt = 0; kt < p.nk; ++kt) {
    // this wave's DMAs of tile kt have landed once at most (NSTAGE-2) tiles' worth remain outstanding
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * L_TILE) : "memory");
    __builtin_amdgcn_s_barrier();   // every wave's part of tile kt is in LDS; stage `wr` (read at kt-1) is free
    __builtin_amdgcn_sched_barrier(0);
#ifdef HD_CONV_TRACE
    if (kt == 0) HD_TRACE(3, clock64());
#endif
    frag(rd, 0, 0);
    frag(rd, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    gload(wr);                      // tile kt+NSTAGE-1 (zeros beyond the last tile: out-of-range offsets)
    compute(rd);
    rd = (rd + 1 == NSTAGE) ? 0 : rd + 1;
    wr = (wr + 1 == NSTAGE) ? 0 : wr + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  HD_TRACE(4, clock64());

  // ---------------- epilogue (conv_epilogue.h) ----------------
  conv_epilogue<BM, BN, WM, WN>(p, acc, lds, m0, n0, tile_m, HoWo);
  HD_TRACE(5, clock64());
  HD_TRACE(6, wall_clock64());
  HD_TRACE(7, hw_ids());
}


template <int BM, int BN, int WM, int WN, bool DUAL, bool KGEN, int NSTAGE>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvP p) {
  conv_igemm_body<BM, BN, WM, WN, DUAL, KGEN, NSTAGE>(p, blockIdx.x, gridDim.x);
}

// several problems in one grid (ConvMulti, conv_params.h): the block picks its problem, then runs the same body
template <int BM, int BN, int WM, int WN, bool KGEN, int NSTAGE>
__global__ __launch_bounds__(256) void conv_igemm_multi_kernel(ConvMulti mp) {
  int pi = 0;
  for (int i = 1; i < mp.n; ++i)
    if ((int)blockIdx.x >= mp.first[i]) pi = i;
  ConvP p = mp.p[pi];
  conv_igemm_body<BM, BN, WM, WN, false, KGEN, NSTAGE>(p, (int)blockIdx.x - mp.first[pi], mp.first[pi + 1] - mp.first[pi]);
}

}  // namespace

template <int BM, int BN, int WM, int WN, int NS>
static void launch_variant_bk64(ConvP& p, hipStream_t s) {
  p.gm = hd_cdiv(p.M, BM);
  p.gn = hd_cdiv(p.Cout, BN);
  p.tgroup = hd_conv_tile_order(p);
  dim3 grid(p.gm * p.gn, p.par ? 4 : 1);
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

template <int BM, int BN, int WM, int WN, int NS>
static bool launch_multi_bk64(ConvMulti& mp, hipStream_t s) {
  const bool kgen = (mp.p[0].cin8 % (64 / 8)) != 0;
  int total = 0;
  for (int i = 0; i < mp.n; ++i) {
    ConvP& p = mp.p[i];
    if (p.x2 || p.par || ((p.cin8 % (64 / 8)) != 0) != kgen) return false;
    p.nk = (p.nchunks + 64 / 8 - 1) / (64 / 8);
    p.gm = hd_cdiv(p.M, BM);
    p.gn = hd_cdiv(p.Cout, BN);
    p.tgroup = hd_conv_tile_order(p);
    mp.first[i] = total;
    total += p.gm * p.gn;
  }
  mp.first[mp.n] = total;
  if (kgen) hipLaunchKernelGGL((conv_igemm_multi_kernel<BM, BN, WM, WN, true, NS>), dim3(total), dim3(256), 0, s, mp);
  else hipLaunchKernelGGL((conv_igemm_multi_kernel<BM, BN, WM, WN, false, NS>), dim3(total), dim3(256), 0, s, mp);
  return true;
}

bool hd_conv_launch_bk64_multi(ConvMulti& mp, int bm, int bn, bool deep, hipStream_t s) {
  if (bn == 32) return launch_multi_bk64<128, 32, 4, 1, 2>(mp, s);
  if (bm == 128) {
    if (bn == 128) return deep ? launch_multi_bk64<128, 128, 2, 2, 3>(mp, s) : launch_multi_bk64<128, 128, 2, 2, 2>(mp, s);
    return deep ? launch_multi_bk64<128, 64, 2, 2, 3>(mp, s) : launch_multi_bk64<128, 64, 2, 2, 2>(mp, s);
  }
  if (bn == 128) return deep ? launch_multi_bk64<64, 128, 2, 2, 3>(mp, s) : launch_multi_bk64<64, 128, 2, 2, 2>(mp, s);
  return deep ? launch_multi_bk64<64, 64, 2, 2, 3>(mp, s) : launch_multi_bk64<64, 64, 2, 2, 2>(mp, s);
}
