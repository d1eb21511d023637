// 3x3 / stride 1 / pad 1 convolution, 8-wave family with LDS-STAGED INPUT PATCHES (Cin % 64 == 0, NHWC f16 out, Cout % 8 == 0).
//
// What bounds the im2col kernels on this chip (measured, tools/w8_trace.py + rocprofv3 PMC): every 1-KiB
// LDS-DMA piece costs the issuing wave 60-100+ clocks, and an im2col K step of a BM x BN tile needs (BM + BN) / 8 pieces
// for BM*BN*64*2/4096 clocks of MFMA work -- 6 pieces per wave per 1024 MFMA clocks at 256 x 128, twice that at BN = 64.
// A 3x3 window re-reads every input pixel nine times, so here a block stages the (TH+2) x (8+2) input patch of its
// TH x 8 output pixels ONCE per 64-channel chunk and the nine taps read shifted windows of it:
//     pieces per K step (one tap, 64 channels):  BN/8 (weights)  +  (TH+2)*10/8/9 (patch)  =  16 + 4.7 at 256 x 128
// i.e. 2.6 per wave instead of 6, and 1.6 instead of 5 at BN = 64 where the A operand dominated.
//
// Structure (shared with conv_igemm_w8.hip): 512 threads, every wave a 64 x 64 accumulator tile, waves as WM x WN x WK
// (WK = split of the four 16-deep MFMA sub-steps of a K step), PING-PONG wave groups (waves 0-3 compute while waves 4-7
// issue addresses + DMA, one barrier per phase), 3-deep weight ring + double-buffered patch, fp32 LDS epilogue tile.
//   * tile = TH rows x 8 columns of one image (TH = 32: BM 256, TH = 16: BM 128): the widths of this network's feature
//     maps (160/80/40/20, 75/38) are covered at 83-100 % by 8-wide tiles; an MFMA row block (32 pixels) is 4 rows x 8;
//   * patch stage: pixel-major, pitch 10 pixels, 8 slots of 16 B (64 channels) per pixel, filled by LDS-DMA (lane-linear:
//     unit u = pixel*8 + slot lands at byte u*16), swizzle on the SOURCE side: slot s of pixel (y, x) holds channel group
//     s ^ (((x >> 1) + 4*y) & 7) -- found by exhaustive search (conflict-free for all nine taps under ds_read_b128's
//     16-lane groups {0-3,12-15,20-27},{4-11,16-19,28-31});
//   * decoder convs gather the patch from two sources in place: channels < C1 from the nearest-2x upsampled low-resolution
//     tensor (pixel (y>>1, x>>1)), the rest from the skip tensor (reference: decoders/unet/decoder.py:38-41);
//   * K order: channel chunk (outer) x tap (inner); the patch of chunk c+1 is fetched during taps 1-6 of chunk c.
//   * TS (round 5, the 128 x 64 tile): the two ping-pong groups take ALTERNATE K steps instead of halves of every step, 6-deep weight
//     ring, weights requested two ROUNDS (four steps) ahead.  tools/w8_trace.py: the sub-step split of that tile requested its weights
//     two 540-clock steps ahead -- less than an L2 / Infinity-Cache round trip, so the 512-channel layers (4.7 MB of weights) sat in
//     s_waitcnt vmcnt; with the step split 16x20x512 runs 26.7 -> 21.2 us.  On the 128-wide tiles the step split measures equal or
//     slower (the LOAD phase issues two steps' DMA pieces and becomes as long as the 16-MFMA phase: 779 -> 788 clocks per step), so
//     they keep the sub-step split (HD_W8_TS=2 forces it everywhere; profiles/r05_probe_w8_ts.txt).
#include "hd_common.h"
#include "conv_params.h"
#include "wgrad3x3_w8_body.h"
#include "conv_w8_epilogue.h"

namespace {

constexpr int LDS_ROW = 64;   // halves per weight row per K step (128 B)
constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int TW = 8, PW = 10;

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, f16* lds_dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds_dst, 16, voff, 0, 0, 0);
}

__device__ __forceinline__ int swz_of(int y, int x) { return ((x >> 1) + 4 * y) & 7; }

// LDS of one block, in halves: max(main-loop ring, epilogue tile)
template <int TH, int WN, int WK, bool TS = false>
constexpr int p8_lds_halves() {
  constexpr int BM = TH * TW, BN = WN * 64;
  constexpr int NPIECE = ((TH + 2) * PW * 8 + 63) / 64;
  constexpr int RING = 2 * NPIECE * 512 + (TS ? 6 : 3) * BN * LDS_ROW;
  constexpr int EPI = WK * BM * (BN + 4) * 2;
  return RING > EPI ? RING : EPI;
}

// One output tile (block `bid_in` of a grid of `nwg_in` tiles); `lds`: p8_lds_halves<TH, WN, WK>() halves, 1 KiB aligned.  A device
// function so that the fused data-gradient + weight-gradient launch below can run it in the leading blocks of its grid.
template <int TH, int WN, int WK, bool DUAL, bool TS = false>
__device__ __forceinline__ void conv3x3_w8_body(ConvP& p, f16* lds, int bid_in, int nwg_in) {
  constexpr int BM = TH * TW, BN = WN * 64, WM = TH / 8;
  static_assert(WM * WN * WK == 8, "eight waves");
  static_assert(!TS || WK >= 2, "the step split needs two K groups");
  // TS (round 5): the two ping-pong wave groups take ALTERNATE K steps (group g the steps s = g mod 2) and a wave runs all of a step's
  // sub-steps that its group has not split further (WK 4: two waves of a group halve them) -- 16 (8) MFMAs per phase where the
  // sub-step split ran 8 (4) between the same two barriers.
  constexpr int KSP = TS ? 8 / WK : 4 / WK;
  constexpr int NRING = TS ? 6 : 3;
  constexpr int PH = TH + 2, PPX = PH * PW;
  constexpr int NPIECE = (PPX * 8 + 63) / 64;          // 1-KiB pieces per patch chunk (43 / 23)
  constexpr int PPW = (NPIECE + 7) / 8;                // pieces per wave (6 / 3)
  constexpr int PSTAGE = NPIECE * 512;                 // halves
  constexpr int B_LOADS = BN / 64;
  constexpr int BSTAGE = BN * LDS_ROW;
  constexpr int RING = 2 * PSTAGE + NRING * BSTAGE;
  constexpr int CP = BN + 4;                            // epilogue tile pitch in floats (16-byte LDS writes conflict-free)
  constexpr int EPI_HALVES = WK * BM * CP * 2;
  constexpr int LDS_HALVES = RING > EPI_HALVES ? RING : EPI_HALVES;
  static_assert(LDS_HALVES == p8_lds_halves<TH, WN, WK, TS>(), "p8_lds_halves out of date");
  f16* const patch0 = lds;
  f16* const bst0 = lds + 2 * PSTAGE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave / (WM * WN);
  const int wmn = wave - wk * (WM * WN);
  const int wm = wmn / WN, wn = wmn - wm * WN;
  const int grp = wave >> 2;
  HD_TRACE(0, wall_clock64());
  HD_TRACE(1, clock64());

  int bid = bid_in;
  {
    const int nwg = nwg_in, xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
    bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  }
  const int tile_m = bid / p.gn, tile_n = bid - tile_m * p.gn;
  const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
  const int n_img = tile_m / (tiles_x * tiles_y);
  const int trem = tile_m - n_img * tiles_x * tiles_y;
  const int tyi = trem / tiles_x;
  const int ty0 = tyi * TH, tx0 = (trem - tyi * tiles_x) * TW;
  const int n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.x), 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(DUAL ? p.x2 : p.x), 0, DUAL ? p.x2bytes : p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.w), 0, p.wbytes, 0x00020000);

  // An out-of-range lane keeps its offset at >= 2^31 whatever uniform offset is added later (all tensors here are < 2 GiB:
  // checked by hd_conv_p8_eligible), so the per-step address of a piece is ONE v_add and the hardware zero-fills.
  constexpr unsigned OOBB = 0x80000000u;
  // ---- patch fill: piece k of this wave is piece k*8 + wave of the chunk; fixed pixel / slot per lane for the whole K loop
  unsigned pb1[PPW], pb2[PPW];
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    const int u = (k * 8 + wave) * 64 + lane;
    const int pp = u >> 3, slot = u & 7;
    const int y = (pp * 6554) >> 16, x = pp - y * PW;       // pp / 10 (exact for pp < 16 384)
    const int iy = ty0 - 1 + y, ix = tx0 - 1 + x;
    const bool v = (u < PPX * 8) && ((unsigned)iy < (unsigned)p.Hin) && ((unsigned)ix < (unsigned)p.Win);
    const unsigned cg16 = (unsigned)((slot ^ swz_of(y, x)) & 7) * 16u;
    if (DUAL) {
      pb1[k] = v ? (unsigned)(((n_img * p.Hsrc + (iy >> 1)) * p.Wsrc + (ix >> 1)) * p.C1) * 2u + cg16 : OOBB;
      pb2[k] = v ? (unsigned)(((n_img * p.Hin + iy) * p.Win + ix) * p.C2) * 2u + cg16 : OOBB;
    } else {
      pb1[k] = v ? (unsigned)(((n_img * p.Hin + iy) * p.Win + ix) * p.C1) * 2u + cg16 : OOBB;
      pb2[k] = OOBB;
    }
  }
  // ---- weight fill: row (tid>>3) + i*64, slot swizzle by row as in the igemm kernels
  const int j = (tid & 7) ^ ((tid >> 4) & 7);
  unsigned wbase[B_LOADS];
#pragma unroll
  for (int i = 0; i < B_LOADS; ++i) {
    const int co = n0 + (tid >> 3) + i * 64;
    wbase[i] = co < p.Cout ? (unsigned)co * (unsigned)p.Ktot * 2u + (unsigned)j * 16u : OOBB;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int ncc = p.Cin >> 6;
  const int c1chunks = p.C1 >> 6;
  const unsigned lds0 = (unsigned)(size_t)lds;             // LDS byte address of the array (M0 / ds_read bases)

  // ---- fragment read tables (byte offsets, first 16-deep sub-step of this wave; sub-step q is the same address ^ (q << 5)):
  //        a0[a][tap]: patch pixel of this lane's row at that tap, slot = (kc0 ^ swizzle(pixel)); b0: weight row, slot by row
  const int frow = lane & 31, fh = lane >> 5;
  // TS: entry i of the table is ROUND i of the 9-round period (18 K steps = two channel chunks): this group's step of that round is
  // 2i + grp, i.e. tap (2i + grp) % 9 of the chunk pair's first (second, once 2i + grp >= 9) chunk -- the patch stage rides in the entry.
  const int kc0 = TS ? ((wk & (WK / 2 - 1)) * KSP) * 2 + fh : (wk * KSP) * 2 + fh;
  unsigned a0[2][9];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int e = TS ? 2 * t + grp : t;
      const int tap = e >= 9 ? e - 9 : e;
      const int ky = TS ? (tap * 11) >> 5 : t / 3;              // tap / 3 for tap < 9
      const int y = wm * 8 + a * 4 + (frow >> 3) + ky, x = (frow & 7) + (tap - 3 * ky);
      a0[a][t] = (unsigned)((y * PW + x) * 128 + ((kc0 ^ swz_of(y, x)) & 7) * 16) + (TS && e >= 9 ? (unsigned)(PSTAGE * 2) : 0u);
    }
  const unsigned b0 = (unsigned)(2 * PSTAGE * 2 + (wn * 64 + frow) * 128 + ((kc0 ^ ((frow >> 1) & 7)) & 7) * 16) + (TS ? (unsigned)(grp * BSTAGE * 2) : 0u);

  auto issue_patch_piece = [&](int cc, int k) {          // piece k of this wave, chunk cc -> patch stage cc & 1
    f16* dst = patch0 + (cc & 1) * PSTAGE + (k * 8 + wave) * 512;
    const bool second = DUAL && cc >= c1chunks;            // uniform
    const unsigned coff = cc < ncc ? (unsigned)(second ? cc - c1chunks : cc) * 128u : OOBB;
    if (second) dma16(rx2, dst, pb2[k] + coff);
    else dma16(rx, dst, pb1[k] + coff);
  };
  auto issue_b = [&](int cc, int tap, int stage) {
    f16* dst = bst0 + stage * BSTAGE + wave * 512;
    const unsigned koff = cc < ncc ? (unsigned)(tap * p.Cin + cc * 64) * 2u : OOBB;
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) dma16(rw, dst + i * (64 * LDS_ROW), wbase[i] + koff);
  };

  HD_TRACE(2, clock64());
#ifdef HD_CONV_TRACE
  long long tr_mem = 0, tr_b1 = 0, tr_mfma = 0, tr_b2 = 0, tr_t = 0;
#define TR_MARK(accu) do { long long n_ = clock64(); accu += n_ - tr_t; tr_t = n_; } while (0)
#else
#define TR_MARK(accu) do {} while (0)
#endif
  // ---- main loop: ping-pong between the wave groups (waves 0-3 / 4-7, one wave of each per SIMD), phases one barrier apart:
  //     LOAD(s):  this wave's 16 fragments of K step s: LDS -> registers; DMA pieces of K step s+2 (weights into ring stage
  //               (s+2) % 3, taps 1..PPW also one patch piece of the next channel chunk); s_waitcnt vmcnt(<pieces just issued>):
  //               everything issued in EARLIER phases has landed
  //     MFMA(s):  16 MFMAs straight from registers (the matrix pipe's phase is pure issue: no LDS latency, no counters)
  //     group 0:  LOAD(0) MFMA(0) LOAD(1) MFMA(1) ...      group 1:  (wait) LOAD(0) MFMA(0) LOAD(1) ...
  // A wave issues one instruction per ~4 clocks, so LOAD must be SHORT to hide under the partner's 512-clock MFMA phase
  // (measured: 115 instructions = 877 clocks with run-time tap arithmetic): the nine taps are unrolled, every per-lane address
  // is a table entry plus a block-uniform offset, the four sub-step addresses differ by an XOR of the slot bits.
  // Hazards: a piece issued in LOAD(s) by either group is retired by that group's vmcnt at the end of its LOAD(s+1), i.e. at
  // least one barrier before the first read of K step s+2 (group 0's LOAD(s+2)); ring stage (s+2) % 3 == (s-1) % 3 was last read
  // in LOAD(s-1), which for both groups ends at least one barrier before any LOAD(s) starts; the patch buffer of chunk c+1 was last
  // read in the LOADs of chunk c-1's tap 8 and is first written in LOAD(c, tap 1).
  if constexpr (TS) {
    // ---- step-split main loop.  Round r = K steps 2r (group 0) and 2r + 1 (group 1); phases as above, one barrier apart:
    //     LOAD(r):  this wave's 4 * KSP fragments of ITS step: LDS -> registers; DMA of the weights of BOTH steps of round r + 2 (ring
    //               stages (2r + 4) % 6 and (2r + 5) % 6, all eight waves share a stage's pieces as before); rounds 0-2 / 5-7 of the
    //               period also this wave's patch pieces of the next odd / even channel chunk; s_waitcnt vmcnt(<just issued>)
    //     MFMA(r):  4 * KSP * 4 MFMAs from registers
    // Hazards: LOAD(r) of group 1 runs one phase after group 0's and a barrier before group 0's LOAD(r + 1), so while any wave is in
    // LOAD(r + 1) every read of LOAD(<= r) has completed: the ring stages of round r + 2 are those of round r - 1 (free), and a piece
    // issued in LOAD(r) by either group has been waited for (end of that wave's LOAD(r + 1)) before LOAD(r + 2) of group 0 begins.
    // Patch buffer (chunk & 1): the odd chunk 2P + 1 is first read in round 9P + 4 and its buffer was last read in round 9P - 1 ->
    // issued in rounds 9P + {0, 1, 2}; the even chunk 2P + 2 is first read in round 9P + 9, its buffer last in round 9P + 4 ->
    // issued in rounds 9P + {5, 6, 7}.
    constexpr int PPR = (PPW + 2) / 3;                   // patch pieces per wave and round
    // (an odd number of chunks: group 1's step of the last round is step 9 * ncc, whose weights AND patch are the zero fill of an
    // out-of-range chunk -- it adds 0 * 0)
    auto issue_b_ts = [&](int cc, int tap, int stage) {
      f16* dst = bst0 + stage * BSTAGE + wave * 512;
      const unsigned koff = cc < ncc ? (unsigned)(tap * p.Cin + cc * 64) * 2u : OOBB;
#pragma unroll
      for (int i = 0; i < B_LOADS; ++i) dma16(rw, dst + i * (64 * LDS_ROW), wbase[i] + koff);
    };
#pragma unroll
    for (int k = 0; k < PPW; ++k)
      if (k * 8 + wave < NPIECE) issue_patch_piece(0, k);
    issue_b_ts(0, 0, 0);
    issue_b_ts(0, 1, 1);
    issue_b_ts(0, 2, 2);
    issue_b_ts(0, 3, 3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();

    HD_TRACE(3, clock64());
#ifdef HD_CONV_TRACE
    tr_t = clock64();
#endif
#define HD_P8_ROUND(I)                                                                                                     \
      {                                                                                                                    \
        constexpr int E0 = 2 * (I) + 4, E1 = 2 * (I) + 5;                                                                  \
        f16x8 af[KSP][2], bf[KSP][2];                                                                                      \
        {                                                                                                                  \
          const char* lb = reinterpret_cast<const char*>(lds);                                                             \
          const unsigned ba0 = a0[0][I], ba1 = a0[1][I], bb = b0 + ((2 * (I)) % 6) * BSTAGE * 2;                            \
          _Pragma("unroll") for (int q = 0; q < KSP; ++q) {                                                                \
            af[q][0] = *reinterpret_cast<const f16x8*>(lb + (ba0 ^ (unsigned)(q << 5)));                                   \
            af[q][1] = *reinterpret_cast<const f16x8*>(lb + (ba1 ^ (unsigned)(q << 5)));                                   \
            bf[q][0] = *reinterpret_cast<const f16x8*>(lb + (bb ^ (unsigned)(q << 5)));                                    \
            bf[q][1] = *reinterpret_cast<const f16x8*>(lb + (bb ^ (unsigned)(q << 5)) + 4096);                             \
          }                                                                                                                \
        }                                                                                                                  \
        issue_b_ts(2 * P + E0 / 9, E0 % 9, E0 % 6);                                                                        \
        issue_b_ts(2 * P + E1 / 9, E1 % 9, E1 % 6);                                                                        \
        {                                                                                                                  \
          constexpr bool PODD = (I) <= 2, PEVEN = (I) >= 5 && (I) <= 7;                                                    \
          constexpr int K0 = (PODD ? (I) : (I) - 5) * PPR;                                                                 \
          int np = 0;                                                                                                      \
          if (PODD || PEVEN) {                                                                                             \
            _Pragma("unroll") for (int k = K0; k < K0 + PPR && k < PPW; ++k)                                               \
              if (k * 8 + wave < NPIECE) {                                                                                 \
                issue_patch_piece(2 * P + (PODD ? 1 : 2), k);                                                              \
                ++np;                                                                                                      \
              }                                                                                                            \
          }                                                                                                                \
          if (np == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * B_LOADS) : "memory");                                  \
          else if (np == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * B_LOADS + 1) : "memory");                         \
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * B_LOADS + 2) : "memory");                                      \
        }                                                                                                                  \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        TR_MARK(tr_mem);                                                                                                   \
        __builtin_amdgcn_s_barrier();                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        TR_MARK(tr_b1);                                                                                                    \
        __builtin_amdgcn_s_setprio(1);                                                                                     \
        _Pragma("unroll") for (int q = 0; q < KSP; ++q)                                                                    \
          _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                                    \
            _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                                  \
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[q][b], af[q][a], acc[a][b], 0, 0, 0);                  \
        __builtin_amdgcn_s_setprio(0);                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        TR_MARK(tr_mfma);                                                                                                  \
        __builtin_amdgcn_s_barrier();                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        TR_MARK(tr_b2);                                                                                                    \
      }
    // whole periods (two channel chunks each), then the five rounds of a last odd chunk: no exit from the middle of the unrolled body
    // (nine exits, each carrying the 64 accumulator registers, had the allocator spill 500 VGPRs)
    const int nper = ncc >> 1;
    for (int P = 0; P < nper; ++P) {
      HD_P8_ROUND(0) HD_P8_ROUND(1) HD_P8_ROUND(2) HD_P8_ROUND(3) HD_P8_ROUND(4) HD_P8_ROUND(5) HD_P8_ROUND(6) HD_P8_ROUND(7) HD_P8_ROUND(8)
    }
    if (ncc & 1) {
      const int P = nper;
      HD_P8_ROUND(0) HD_P8_ROUND(1) HD_P8_ROUND(2) HD_P8_ROUND(3) HD_P8_ROUND(4)
    }
#undef HD_P8_ROUND
  } else {
#pragma unroll
  for (int k = 0; k < PPW; ++k)
    if (k * 8 + wave < NPIECE) issue_patch_piece(0, k);
  issue_b(0, 0, 0);
  issue_b(0, 1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();

  HD_TRACE(3, clock64());
#ifdef HD_CONV_TRACE
  tr_t = clock64();
#endif
  typedef __attribute__((address_space(3))) const f16x8 lds_f16x8;
  auto ldsread = [&](unsigned byte_addr) -> f16x8 { return *reinterpret_cast<const f16x8*>(reinterpret_cast<const char*>(lds) + (byte_addr - lds0)); };
  (void)ldsread;
  for (int cc = 0; cc < ncc; ++cc) {
    const unsigned pst = (unsigned)((cc & 1) * PSTAGE * 2);      // byte offset of this chunk's patch stage
#define HD_P8_STEP(TAP)                                                                                                    \
    {                                                                                                                      \
      constexpr int BS = (TAP) % 3, BS2 = ((TAP) + 2) % 3, T2 = ((TAP) + 2) % 9, CARRY = ((TAP) + 2) / 9;                    \
      f16x8 af[KSP][2], bf[KSP][2];                                                                                        \
      {                                                                                                                    \
        const char* lb = reinterpret_cast<const char*>(lds);                                                               \
        const unsigned ba0 = a0[0][TAP] + pst, ba1 = a0[1][TAP] + pst, bb = b0 + BS * BSTAGE * 2;                           \
        _Pragma("unroll") for (int q = 0; q < KSP; ++q) {                                                                  \
          af[q][0] = *reinterpret_cast<const f16x8*>(lb + (ba0 ^ (unsigned)(q << 5)));                                     \
          af[q][1] = *reinterpret_cast<const f16x8*>(lb + (ba1 ^ (unsigned)(q << 5)));                                     \
          bf[q][0] = *reinterpret_cast<const f16x8*>(lb + (bb ^ (unsigned)(q << 5)));                                      \
          bf[q][1] = *reinterpret_cast<const f16x8*>(lb + (bb ^ (unsigned)(q << 5)) + 4096);                               \
        }                                                                                                                  \
      }                                                                                                                    \
      issue_b(cc + CARRY, T2, BS2);                                                                                        \
      if ((TAP) >= 1 && (TAP) <= PPW && ((TAP) - 1) * 8 + wave < NPIECE) {                                                 \
        issue_patch_piece(cc + 1, (TAP) - 1);                                                                              \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(B_LOADS + 1) : "memory");                                                 \
      } else {                                                                                                             \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(B_LOADS) : "memory");                                                     \
      }                                                                                                                    \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                                   \
      TR_MARK(tr_mem);                                                                                                     \
      __builtin_amdgcn_s_barrier();                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                                   \
      TR_MARK(tr_b1);                                                                                                      \
      __builtin_amdgcn_s_setprio(1);                                                                                       \
      _Pragma("unroll") for (int q = 0; q < KSP; ++q)                                                                      \
        _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                                      \
          _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                                    \
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[q][b], af[q][a], acc[a][b], 0, 0, 0);                    \
      __builtin_amdgcn_s_setprio(0);                                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                                   \
      TR_MARK(tr_mfma);                                                                                                    \
      __builtin_amdgcn_s_barrier();                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                                   \
      TR_MARK(tr_b2);                                                                                                      \
    }
    HD_P8_STEP(0) HD_P8_STEP(1) HD_P8_STEP(2) HD_P8_STEP(3) HD_P8_STEP(4) HD_P8_STEP(5) HD_P8_STEP(6) HD_P8_STEP(7) HD_P8_STEP(8)
#undef HD_P8_STEP
  }
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  HD_TRACE(4, clock64());
#ifdef HD_CONV_TRACE
  HD_TRACE(8, (unsigned long long)tr_mem);
  HD_TRACE(9, (unsigned long long)tr_b1);
  HD_TRACE(10, (unsigned long long)tr_mfma);
  HD_TRACE(11, (unsigned long long)tr_b2);
#endif

  // ---------------- epilogue (as conv_igemm_w8.hip; tile row r = pixel (ty0 + r/8, tx0 + r%8)) ----------------
  // The weights are the MFMA's A operand, so a lane's accumulator registers 4g..4g+3 are four CONSECUTIVE output channels of
  // one pixel (C/D map: row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) = channel, column = lane & 31 = pixel): one 16-byte LDS
  // write per quad (16 per lane instead of 64 four-byte ones), pitch BN + 4 floats keeps the 8-lane write groups on
  // distinct banks.
  float* ct = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = wm * 64 + a * 32 + (lane & 31);
        const int col = wn * 64 + b * 32 + 8 * g + 4 * (lane >> 5);
        f32x4 v4 = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
        *reinterpret_cast<f32x4*>(ct + (wk * BM + row) * CP + col) = v4;
      }

  hd_w8_epilogue<BM, BN, WK, TW>(p, lds, n_img, ty0, tx0, n0, tile_m);
  HD_TRACE(5, clock64());
  HD_TRACE(6, wall_clock64());
  HD_TRACE(7, hw_ids());
}

template <int TH, int WN, int WK, bool DUAL, bool TS>
__global__ __launch_bounds__(512, 2) void conv3x3_w8_kernel(ConvP p) {
  __shared__ __attribute__((aligned(1024))) f16 lds[p8_lds_halves<TH, WN, WK, TS>()];
  conv3x3_w8_body<TH, WN, WK, DUAL, TS>(p, lds, blockIdx.x, gridDim.x);
}

// Data gradient AND weight gradient of one layer as ONE grid: blocks [0, n_conv) run the convolution tiles, the blocks behind them
// the 8-wave weight gradient (wgrad3x3_w8_body.h).  The deep U-Net layers have 160 convolution tiles for 256 CUs at batch 8 -- 37 % of
// the chip idles through every one of those launches, and through the tail of the 256-block weight gradient that follows it; both
// consume the same dY and are independent, so one grid of 160 + 256 blocks keeps every CU busy until the work of both is done.
// Same code, same summation orders: results are bit-identical to the two separate launches.
template <int TH, int WN, int WK, bool TS>
__global__ __launch_bounds__(512, 2) void conv3x3_w8_wgrad_kernel(ConvP p, hd_wg8::Wg8P q, int n_conv, int wg_gx) {
  constexpr int L1 = p8_lds_halves<TH, WN, WK, TS>(), L2 = hd_wg8::LDS_HALVES;
  __shared__ __attribute__((aligned(1024))) f16 lds[L1 > L2 ? L1 : L2];
  if ((int)blockIdx.x < n_conv) {
    conv3x3_w8_body<TH, WN, WK, false, TS>(p, lds, blockIdx.x, n_conv);
  } else {
    const int w = (int)blockIdx.x - n_conv;
    hd_wg8::wgrad3x3_w8_body(q, lds, w % wg_gx, w / wg_gx);
  }
}

template <int TH, int WN, int WK, bool TS = false>
void launch_p8_wgrad(ConvP& p, const hd_wg8::Wg8P& q, int wg_gx, int wg_gy, hipStream_t s) {
  p.gm = p.N * hd_cdiv(p.Ho, TH) * hd_cdiv(p.Wo, TW);
  p.gn = hd_cdiv(p.Cout, WN * 64);
  const int n_conv = p.gm * p.gn;
  hipLaunchKernelGGL((conv3x3_w8_wgrad_kernel<TH, WN, WK, TS>), dim3(n_conv + wg_gx * wg_gy), dim3(512), 0, s, p, q, n_conv, wg_gx);
}

template <int TH, int WN, int WK, bool TS = false>
void launch_p8(ConvP& p, hipStream_t s) {
  p.gm = p.N * hd_cdiv(p.Ho, TH) * hd_cdiv(p.Wo, TW);
  p.gn = hd_cdiv(p.Cout, WN * 64);
  dim3 grid(p.gm * p.gn);
  if (p.x2) hipLaunchKernelGGL((conv3x3_w8_kernel<TH, WN, WK, true, TS>), grid, dim3(512), 0, s, p);
  else hipLaunchKernelGGL((conv3x3_w8_kernel<TH, WN, WK, false, TS>), grid, dim3(512), 0, s, p);
}

}  // namespace

// 3x3 / s1 / p1, same extent in and out, Cin % 64 == 0 (both sources of a decoder concat), NHWC f16 out, Cout % 8 == 0;
// an upsampled source only together with a skip source (the decoder's conv1)
bool hd_conv_p8_eligible(const ConvP& p) {
  if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.in_dil != 1) return false;
  if (p.out_mode != HD_OUT_NHWC_F16 || (p.Cout & 7) != 0 || p.Ho != p.Hin || p.Wo != p.Win) return false;
  if ((p.xbytes | p.x2bytes | p.wbytes) & 0x80000000u) return false;      // out-of-range lanes are marked by bit 31 of the offset
  if (p.x2) return p.up1 && (p.C1 % 64) == 0 && (p.C2 % 64) == 0;
  return !p.up1 && (p.C1 % 64) == 0;
}

// out_pool2 on this family: plain data gradient (no residual / mask / bias / activation / sums), even extent, the pooled channel count a
// multiple of the widest channel tile so that no tile straddles the two halves
bool hd_conv_p8_pool2_ok(const ConvP& p) {
  return hd_conv_p8_eligible(p) && !p.res && !p.mask && !p.bias && !p.stats && !p.bs_y && p.act == HD_ACT_NONE && (p.Ho % 2) == 0 && (p.Wo % 2) == 0 &&
         p.pool2 > 0 && (p.pool2 % 128) == 0 && p.pool2 <= p.Cout && ((p.Cout - p.pool2) % 8) == 0 && (p.pool2 == p.Cout || p.y2 != nullptr);
}

// cfg & 3: 0 256x128 | 1 128x128 (WK 2) | 2 256x64 (WK 2) | 3 128x64 (WK 4);  cfg & 4: the step-split main loop (TS) of tiles 1-3
int hd_conv_p8_tiles(const ConvP& p, int cfg) {
  const int th = ((cfg & 3) == 0 || (cfg & 3) == 2) ? 32 : 16;
  return p.N * hd_cdiv(p.Ho, th) * hd_cdiv(p.Wo, TW);
}

// fused launch (see conv3x3_w8_wgrad_kernel): `p` is a single-source 3x3 conv eligible for tile `cfg`, `wa` an 8-wave weight gradient
void hd_conv_launch_p8_wgrad(ConvP& p, int cfg, const hd_wgrad_args* wa, hipStream_t s) {
  hd_wg8::Wg8P q;
  int gx, gy;
  hd_wg8::fill_params(wa, q, &gx, &gy);
  switch (cfg) {
    case 0: case 4: launch_p8_wgrad<32, 2, 1>(p, q, gx, gy, s); break;
    case 1: launch_p8_wgrad<16, 2, 2>(p, q, gx, gy, s); break;
    case 2: launch_p8_wgrad<32, 1, 2>(p, q, gx, gy, s); break;
    case 3: launch_p8_wgrad<16, 1, 4>(p, q, gx, gy, s); break;
    case 5: launch_p8_wgrad<16, 2, 2, true>(p, q, gx, gy, s); break;
    case 6: launch_p8_wgrad<32, 1, 2, true>(p, q, gx, gy, s); break;
    default: launch_p8_wgrad<16, 1, 4, true>(p, q, gx, gy, s); break;
  }
}

void hd_conv_launch_p8(ConvP& p, int cfg, hipStream_t s) {
  switch (cfg) {
    case 0: case 4: launch_p8<32, 2, 1>(p, s); break;
    case 1: launch_p8<16, 2, 2>(p, s); break;
    case 2: launch_p8<32, 1, 2>(p, s); break;
    case 3: launch_p8<16, 1, 4>(p, s); break;
    case 5: launch_p8<16, 2, 2, true>(p, s); break;
    case 6: launch_p8<32, 1, 2, true>(p, s); break;
    default: launch_p8<16, 1, 4, true>(p, s); break;
  }
}
