// 3x3 / stride 1 / pad 1 convolution on 160- / 320-PIXEL x 64-CHANNEL tiles (round 6): the tiles that land this network's 12-GFLOP U-Net
// layers on ONE round of the 256 CUs, computed by four CONSUMER waves that never issue a global load and fed by four PRODUCER waves that
// do nothing else.
//
// Why the tile (rounds 3-5): at batch 8 the ResNet-34 stages 2-3 and the decoder's first blocks have M = N*H*W = 40 960 /
// 10 240 output pixels on 128 / 256 channels.  The TH x 8-pixel tiles of conv3x3_w8.hip (256 x 128, 128 x 128, 256 x 64, 128 x 64) cut
// them into 160 or 320 blocks -- 160 leave 96 CUs idle for the whole launch, 320 cost a second round for a quarter of the chip.  A 40-pixel
// wide tile divides these maps exactly: 4 x 40 pixels x 64 channels on the 32x40 maps (10 240 x 256 = 64 x 4 = 256 blocks), 8 x 40 on
// the 64x80 maps (40 960 x 128 = 128 x 2 = 256 blocks).
//
// Why the roles (this round's measurements, profiles/r06_w8_trace_m160_v1.txt / _v2.txt).  The first two versions of this file kept
// conv3x3_w8.hip's ping-pong structure (two groups of four waves; a group loads -- LDS fragment reads + its share of the LDS-DMA issue --
// while the other multiplies).  Per-block stamps read a LOAD phase of 430 - 560 clocks against a 390-clock MFMA phase whatever the
// fragment count (14 reads per 20 MFMAs with 80 x 32 wave tiles: 568 clocks per 64-deep step at 1.81 GHz; 9 reads with 80 x 64 wave tiles:
// 659 at 2.07 GHz -- the same 11.4 us): the phase is not LDS-read time, it is the 100 - 185 clocks EACH LDS-DMA piece holds its issuing
// wave (MI355X_MICROARCH.md, cycle constants) -- 3 - 4 pieces per wave and round that no partner can hide, because the partner's MFMAs
// end when they end.  So here the waves that multiply issue NO vector-memory instruction:
//   * waves 4-7 (producers): per K step two 1-KiB weight pieces each (64 x 64 halves = 8 pieces) FIVE steps ahead of the MFMAs, and during
//     taps 0-4 of a channel chunk two or three pieces each of the NEXT chunk's input patch; one counted s_waitcnt vmcnt (two steps' pieces stay
//     in flight) + one s_barrier per K step;
//   * waves 0-3 (consumers, one per SIMD): software-pipelined over K steps -- the 9 fragments (4 weight + 5 pixel, ds_read_b128) of the
//     next 32-deep unit are requested, then the 20 MFMAs of the current unit run from the other register set -- one s_barrier per K step.
//     A consumer owns 80 pixels x 64 channels = 5 x 4 accumulator blocks of v_mfma_f32_16x16x32_f16 (80 registers; 160 pixels = five
//     32-row blocks do not split over four SIMDs, ten 16-pixel blocks do; and MI355X_MICROARCH.md, DVFS give-back item 7: that shape
//     holds a ~15 % higher clock than 32x32x16 at equal cycles per FLOP).  A 16-pixel block is 2 tile rows x 8 columns.
//       TH = 4: consumers = 2 (tile rows 0-1 / 2-3) x 2 (32-deep halves of every K step): two partial sums per output, added by the
//               shared epilogue (conv_w8_epilogue.h, WK = 2);
//       TH = 8: consumers = 4 row pairs, each runs both halves of every K step: no partial sums (epilogue tile [320][68] fp32).
// Hazards.  Step s (tap s % 9 of chunk s / 9) lives in ring stage s % 6; barrier s ends step s.  Between barriers s-1 and s a producer
// issues step s+5's weights and (taps 0-4) next-chunk patch pieces, then waits until everything it issued TWO OR MORE steps ago has landed; a
// consumer requests the fragments of step s+1 and multiplies step s.  So step s+3's bytes have landed (every producer's wait) before
// barrier s, and step s+1's are first read after barrier s-1; the stage written in step s (s+5) is that of step s-1, last read in step s-1; the patch buffer of chunk c+1
// (chunk parity) was last read in step (c-1, tap 7) -- its tap-8 fragments are requested there -- and is written from step (c, tap 0),
// two barriers later; it is complete before barrier (c, tap 7) (that step's wait covers everything issued in taps <= 5) and first read
// in step (c, tap 8).  Every consumer waits for its outstanding LDS reads before each barrier.  Both roles execute exactly one barrier
// per K step plus one in front: the counts match by construction (the same loop table drives both).
//
// Shared with conv3x3_w8.hip: weights are the MFMA A operand; the (TH+2) x 42-pixel input patch of a 64-channel chunk is staged ONCE in
// LDS (double-buffered, lane-linear LDS-DMA, swizzle on the source side) and the nine taps read shifted windows of it; decoder convs
// gather the patch from two sources in place (nearest-2x upsampled + skip); fp32 LDS epilogue tile with residual / mask / bias /
// activation / BatchNorm sums / bs_* / out_pool2.
//   * LDS swizzle, found by tools/search_swizzle_m160.py (exhaustive over the nine taps, both sub-steps, every block position, under
//     ds_read_b128's 16-lane groups): slot s of patch pixel (y, x) holds channel group s ^ (x & 7), pitch 42 pixels.  It depends on the
//     COLUMN only and a block is 8 columns wide, so a lane's swizzle is the same for its five blocks and three tap rows: THREE address
//     registers per consumer (tap column), everything else an instruction offset (block b: + b KiB, tap row: + ky * 5 376, stage).
// LDS: TH = 4: 2 x 32 KiB patch + 6 x 8 KiB weights = 112 KiB; TH = 8: 2 x 53 KiB + 48 KiB = 154 KiB (epilogue tile 85 KiB): one block per CU.
// The 4 x 24-pixel instance (TW = 24: the detector's 19 x 19 / 10 x 10 / 5 x 5 maps, ResNet-34 layer4): three 8-column blocks per row pair,
// 2 x 20 KiB patch + FIVE 8-KiB weight stages = 80 KiB and 116 registers -> TWO blocks per CU; the ring stage is a run-time counter there
// (s % 5, weights four steps ahead; five does not divide the 18-step table), everything else -- roles, hazards, swizzle -- as above.
#include "hd_common.h"
#include "conv_params.h"
#include "conv_w8_epilogue.h"

namespace {

constexpr int BN = 64;
constexpr int BSTAGE = BN * 64;                       // halves per weight stage (8 KiB)
constexpr int CP = BN + 4;

// TW = 40: the U-Net maps (and every map the 40-pixel width covers well); TW = 24 (TH = 4 only: 96 pixels x 64 channels): the detector's
// 19 x 19 maps, which a 40-wide tile covers to 47 % -- three 8-column blocks per row pair instead of five, same roles, same swizzle
// (tools/search_swizzle_m160.py: slot ^ (x & 7) is conflict-free at the 26-pixel pitch too).
template <int TH, int TW>
struct M160 {
  static constexpr int PW = TW + 2, NB = TW / 8;               // patch pitch in pixels; 8-column blocks per row pair
  // weight ring: 6 stages (a compile-time stage per step of the 18-step period); the 24-wide tile takes 5 -- 2 x 20 KiB of patch + 5 x 8 KiB
  // = 80 KiB: TWO blocks per CU (its consumers hold 12 accumulator blocks: 128 registers), each hiding the other's set-up and epilogue.
  // Five does not divide the period, so there the ring stage is a run-time counter (one scalar add per step in either role).
  static constexpr int NRING = TW == 24 ? 5 : 6;
  static constexpr int BMH = 4 * TW;                           // the epilogue's unit: 4 tile rows x 64 channels
  static constexpr int EPI_HALVES = 2 * BMH * CP * 2;
  static constexpr int PH = TH + 2;
  static constexpr int PPX = PH * PW;                          // 252 / 420 patch pixels (TW = 40)
  static constexpr int NPIECE = (PPX * 8 + 63) / 64;           // 32 / 53 one-KiB pieces per patch chunk
  static constexpr int PK = (NPIECE + 3) / 4;                  // 8 / 14 pieces per producer wave and chunk
  static constexpr int PSTAGE = NPIECE * 512;                  // halves per patch stage
  static constexpr int RING = 2 * PSTAGE + NRING * BSTAGE;
  static constexpr int LDS_HALVES = RING > EPI_HALVES ? RING : EPI_HALVES;
  static constexpr int NQ = TH == 8 ? 2 : 1;                   // 32-deep units of a K step a consumer runs
  static_assert(LDS_HALVES * 2 <= 160 * 1024, "LDS");
};

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, f16* lds_dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds_dst, 16, voff, 0, 0, 0);
}

template <int TH, int TW, bool DUAL>
__device__ __forceinline__ void conv3x3_m160_body(ConvP& p, f16* lds, int bid_in, int nwg_in) {
  using G = M160<TH, TW>;
  constexpr int PPX = G::PPX, NPIECE = G::NPIECE, PK = G::PK, PSTAGE = G::PSTAGE, NQ = G::NQ, PW = G::PW, NB = G::NB, BMH = G::BMH;
  constexpr int NRING = G::NRING, AHEAD = NRING - 1;          // a step's weights are issued AHEAD steps before its MFMAs
  constexpr bool RT_RING = NRING != 6;
  static_assert(!RT_RING || NQ == 1, "run-time ring stage: one 32-deep unit per consumer and step");
  f16* const patch0 = lds;
  f16* const bst0 = lds + 2 * PSTAGE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool consumer = wave < 4;
  HD_TRACE(0, wall_clock64());
  HD_TRACE(1, clock64());

  int bid = bid_in;
  {
    // blocks are dealt round-robin over the 8 XCDs: give each XCD a contiguous run of the tile list (N tiles fastest: the four / two
    // channel tiles of a pixel tile share its patch in that XCD's L2)
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
  const int ncc = p.Cin >> 6;
  const int nper = ncc >> 1;

  // consumer geometry (also the epilogue's: who lays which accumulator rows down)
  const int wm = TH == 8 ? (wave & 3) : ((wave >> 1) & 1);   // tile rows 2 wm, 2 wm + 1
  const int wq = TH == 8 ? 0 : (wave & 1);                   // TH = 4: the 32-deep half of every K step this consumer multiplies
  const int fp = lane & 15;
  f32x4 acc[NB][4];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[b][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // The step table both roles walk: a period = 18 K steps = two channel chunks (ring stage = step % 6 and patch stage = chunk & 1 are then
  // compile-time); whole periods, then the nine steps of a last odd chunk.  ONE barrier per step in either role.
#define HD_M160_STEPS9(S, B)  S(B + 0) S(B + 1) S(B + 2) S(B + 3) S(B + 4) S(B + 5) S(B + 6) S(B + 7) S(B + 8)
#define HD_M160_LOOP(S)                    \
  for (int P = 0; P < nper; ++P) {         \
    HD_M160_STEPS9(S, 0)                   \
    HD_M160_STEPS9(S, 9)                   \
  }                                        \
  if (ncc & 1) {                           \
    const int P = nper;                    \
    HD_M160_STEPS9(S, 0)                   \
  }

  if (!consumer) {
    // =====================================================  PRODUCER  =====================================================
    const int pw = wave - 4, ptid = tid - 256;
    const int c1chunks = p.C1 >> 6;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.x), 0, p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(DUAL ? p.x2 : p.x), 0, DUAL ? p.x2bytes : p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.w), 0, p.wbytes, 0x00020000);
    // An out-of-range lane keeps its offset at >= 2^31 whatever uniform offset is added later (all tensors here are < 2 GiB: checked by
    // hd_conv_m160_eligible), so the per-step address of a piece is ONE v_add and the hardware zero-fills.
    constexpr unsigned OOBB = 0x80000000u;
    // ---- patch fill: piece k of this wave is piece k * 4 + pw of the chunk; fixed pixel / slot per lane for the whole K loop
    unsigned pb1[PK], pb2[PK];
#pragma unroll
    for (int k = 0; k < PK; ++k) {
      const int u = (k * 4 + pw) * 64 + lane;
      const int pp = u >> 3, slot = u & 7;
      const int y = pp / PW, x = pp - y * PW;                 // (a constant divisor: multiply + shift)
      const int iy = ty0 - 1 + y, ix = tx0 - 1 + x;
      const bool v = (u < PPX * 8) && ((unsigned)iy < (unsigned)p.Hin) && ((unsigned)ix < (unsigned)p.Win);
      const unsigned cg16 = (unsigned)((slot ^ x) & 7) * 16u;
      if (DUAL) {
        pb1[k] = v ? (unsigned)(((n_img * p.Hsrc + (iy >> 1)) * p.Wsrc + (ix >> 1)) * p.C1) * 2u + cg16 : OOBB;
        pb2[k] = v ? (unsigned)(((n_img * p.Hin + iy) * p.Win + ix) * p.C2) * 2u + cg16 : OOBB;
      } else {
        pb1[k] = v ? (unsigned)(((n_img * p.Hin + iy) * p.Win + ix) * p.C1) * 2u + cg16 : OOBB;
        pb2[k] = OOBB;
      }
    }
    // ---- weight fill: two pieces per producer and K step: unit (ptid, j) = row (ptid >> 3) + 32 j of the 64-row stage (piece pw + 4 j),
    //      slot ptid & 7 holds channel group (ptid & 7) ^ ((row >> 1) & 7)
    unsigned wbase[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = (ptid >> 3) + 32 * j;
      const int cg = (ptid & 7) ^ ((row >> 1) & 7);
      const int co = n0 + row;
      wbase[j] = co < p.Cout ? (unsigned)co * (unsigned)p.Ktot * 2u + (unsigned)cg * 16u : OOBB;
    }
    auto issue_patch_piece = [&](int cc, int k) {          // piece k of this wave, chunk cc -> patch stage cc & 1
      f16* dst = patch0 + (cc & 1) * PSTAGE + (k * 4 + pw) * 512;
      const bool second = DUAL && cc >= c1chunks;            // uniform
      const unsigned coff = cc < ncc ? (unsigned)(second ? cc - c1chunks : cc) * 128u : OOBB;
      if (second) dma16(rx2, dst, pb2[k] + coff);
      else dma16(rx, dst, pb1[k] + coff);
    };
    auto issue_b = [&](int cc, int tap, int stage) {
      const unsigned koff = cc < ncc ? (unsigned)(tap * p.Cin + cc * 64) * 2u : OOBB;
      dma16(rw, bst0 + stage * BSTAGE + pw * 512, wbase[0] + koff);
      dma16(rw, bst0 + stage * BSTAGE + (pw + 4) * 512, wbase[1] + koff);
    };
    // prologue: the patch of chunk 0 and the weights of K steps 3-4 (those of steps 0-2 come from the consumers, which have nothing else to
    // do yet: 14 back-to-back pieces per producer were 5 200 clocks of set-up, tools/w8_trace.py); all of it has landed before the first barrier
#pragma unroll
    for (int k = 0; k < PK; ++k)
      if (k * 4 + pw < NPIECE) issue_patch_piece(0, k);
#pragma unroll
    for (int t = 3; t < AHEAD; ++t) issue_b(0, t, t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // Step I: the weights of step I + 5 (ring stage (I + 5) % 6 = that of step I - 1, whose fragments were fetched during steps I - 2 / I - 1),
    // during taps 0 .. NPT-1 PPS pieces of the next chunk's patch, then a COUNTED wait that leaves this step's and the previous step's
    // pieces in flight: everything issued two or more steps ago has landed.  (The first version waited for everything but this step's
    // pieces: a piece had ONE K step -- 450 clocks -- to arrive, less than a loaded L2 round trip and far less than the Infinity Cache / HBM
    // latency of a layer's first touch of its weights; the 6-deep ring always had room for this.)  Step s's weights are issued in step
    // s - 5, guaranteed by the wait of step s - 3 and first read in step s - 1; the patch pieces issued in taps <= 4 are guaranteed by
    // the wait of tap 7, one barrier before their first read in tap 8.
    constexpr int NPT = TH == 8 ? 5 : 4, PPS = TH == 8 ? 3 : 2;
    static_assert(NPT * PPS >= PK && NPT <= 5, "patch pieces of a chunk are issued in taps 0-4");
    int np_prev = 0;
    int pst = AHEAD;                           // (RT_RING) ring stage of the step whose weights are issued next
#define HD_M160_PSTEP(I)                                                                                                   \
    {                                                                                                                      \
      constexpr int E = (I) + AHEAD, TAP = (I) % 9;                                                                        \
      if constexpr (RT_RING) {                                                                                             \
        issue_b(2 * P + E / 9, E % 9, pst);                                                                                \
        pst = pst + 1 == NRING ? 0 : pst + 1;                                                                              \
      } else {                                                                                                             \
        issue_b(2 * P + E / 9, E % 9, E % 6);                                                                              \
      }                                                                                                                    \
      int np = 0;                                                                                                          \
      if (TAP < NPT) {                                                                                                     \
        _Pragma("unroll") for (int k = PPS * TAP; k < PPS * TAP + PPS; ++k)                                                \
          if (k < PK && k * 4 + pw < NPIECE) {                                                                             \
            issue_patch_piece(2 * P + (I) / 9 + 1, k);                                                                     \
            ++np;                                                                                                          \
          }                                                                                                                \
      }                                                                                                                    \
      switch (4 + np + np_prev) {                                                                                          \
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;                                                    \
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;                                                    \
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;                                                    \
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;                                                    \
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;                                                    \
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;                                                    \
        default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;                                                  \
      }                                                                                                                    \
      np_prev = np;                                                                                                        \
      __builtin_amdgcn_s_barrier();                                                                                        \
    }
    HD_M160_LOOP(HD_M160_PSTEP)
#undef HD_M160_PSTEP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    // =====================================================  CONSUMER  =====================================================
    // ---- fragment addresses (byte offsets from the start of `lds`).  v_mfma_f32_16x16x32_f16: lane l supplies row / column (l & 15) at
    //      k = 8 (l >> 4) .. + 7 of the 32-deep unit.  Pixel operand: block pixel (l & 15) = (tile row 2 wm + ((l & 15) >> 3), column
    //      8 b + (l & 7)), slot (4 q + (l >> 4)) ^ (x & 7): one register per tap COLUMN (the swizzle sees x only), tap row / block / patch stage
    //      are instruction offsets.  Weight operand: row 16 c + (l & 15) of the stage, slot (4 q + (l >> 4)) ^ ((row >> 1) & 7).
    const int fks = (lane >> 4) + 4 * wq;
    unsigned tB[3][2];           // [tap column][patch stage]
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int y = wm * 2 + (fp >> 3), x = (fp & 7) + kx;
      tB[kx][0] = (unsigned)((y * PW + x) * 128 + ((fks ^ x) & 7) * 16);
      tB[kx][1] = tB[kx][0] + (unsigned)(PSTAGE * 2);
    }
    const unsigned tA = (unsigned)(2 * PSTAGE * 2 + fp * 128 + ((fks ^ (fp >> 1)) & 7) * 16);
    const char* lb = reinterpret_cast<const char*>(lds);
    f16x8 af[2][4], bf[2][NB];        // two register sets: the unit being multiplied and the one being fetched
    int cst = 0;                      // (RT_RING) ring stage of the step whose fragments are fetched next
    // fragments of K step S (tap S % 9, patch stage (S / 9) & 1, ring stage S % 6), 32-deep half Q (TH = 8; TH = 4: the consumer's own) -> SET
#define HD_M160_FETCH(SET, S, Q)                                                                                           \
    {                                                                                                                      \
      constexpr int TAP_ = (S) % 9, KY_ = TAP_ / 3, KX_ = TAP_ % 3, PST_ = ((S) / 9) & 1, RST_ = RT_RING ? 0 : (S) % 6;     \
      const unsigned tb_ = tB[KX_][PST_] ^ (unsigned)((Q) << 6);                                                           \
      unsigned ta_ = tA ^ (unsigned)((Q) << 6);                                                                            \
      if constexpr (RT_RING) {                                                                                             \
        ta_ += (unsigned)(cst * BSTAGE * 2);                                                                               \
        cst = cst + 1 == NRING ? 0 : cst + 1;                                                                              \
      }                                                                                                                    \
      _Pragma("unroll") for (int c = 0; c < 4; ++c) af[SET][c] = *reinterpret_cast<const f16x8*>(lb + ta_ + RST_ * BSTAGE * 2 + c * 2048);   \
      _Pragma("unroll") for (int b = 0; b < NB; ++b) bf[SET][b] = *reinterpret_cast<const f16x8*>(lb + tb_ + KY_ * PW * 128 + b * 1024);      \
    }
#define HD_M160_MFMA(SET)                                                                                                  \
    _Pragma("unroll") for (int b = 0; b < NB; ++b)                                                                         \
      _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                                        \
        acc[b][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[SET][c], bf[SET][b], acc[b][c], 0, 0, 0);
    // instruction order of one unit (9 fragment reads of the next unit + 20 MFMAs of this one, one scheduling region): MFMA, read, MFMA,
    // read ... -- issued back to back in front of the MFMAs the nine reads held the wave for 166 clocks (four consumers push 36 KB through
    // the LDS pipe at once) while its matrix pipe idled: 591 clocks per step for 320 of MFMA work (profiles/r06_w8_trace_m160_v3.txt)
#define HD_M160_ORDER()                                                                                                    \
    _Pragma("unroll") for (int i_ = 0; i_ < 4 + NB; ++i_) {                                                                \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                   \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                                   \
    }                                                                                                                      \
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NB - (4 + NB), 0);
    {
      // prologue: the weights of K steps 0-2 (the one time a consumer issues vector-memory instructions; same piece / slot map as the
      // producers' issue_b with this wave in the place of producer `wave`)
      const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.w), 0, p.wbytes, 0x00020000);
      unsigned wb[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = (tid >> 3) + 32 * j;
        const int cg = (tid & 7) ^ ((row >> 1) & 7);
        const int co = n0 + row;
        wb[j] = co < p.Cout ? (unsigned)co * (unsigned)p.Ktot * 2u + (unsigned)cg * 16u : 0x80000000u;
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const unsigned koff = (unsigned)(t * p.Cin) * 2u;                       // chunk 0, tap t (ncc >= 1)
        dma16(rw, bst0 + t * BSTAGE + wave * 512, wb[0] + koff);
        dma16(rw, bst0 + t * BSTAGE + (wave + 4) * 512, wb[1] + koff);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                      // patch chunk 0 (producers) and weight steps 0-2 (consumers) have landed
    HD_TRACE(2, clock64());
    HD_M160_FETCH(0, 0, 0)
    __builtin_amdgcn_s_setprio(1);
    HD_TRACE(3, clock64());
#ifdef HD_CONV_TRACE
    long long tr_mem = 0, tr_b1 = 0, tr_mfma = 0, tr_b2 = 0, tr_t = clock64();
#define TR_MARK(accu) do { long long n_ = clock64(); accu += n_ - tr_t; tr_t = n_; } while (0)
#else
#define TR_MARK(accu) do {} while (0)
#endif
    // step I of a period: [TH = 8: fetch (I, half 1), multiply (I, half 0);] fetch (I + 1, half 0 / this consumer's half), multiply the
    // current unit; every outstanding LDS read retired; barrier.  (I + 1 = 18 wraps to step 0 of the next period: same tap, stages.)
#define HD_M160_CSTEP(I)                                                                                                   \
    {                                                                                                                      \
      if constexpr (NQ == 2) {                                                                                             \
        HD_M160_FETCH(1, (I), 1)                                                                                           \
        HD_M160_MFMA(0)                                                                                                    \
        HD_M160_ORDER()                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        TR_MARK(tr_mfma);                                                                                                  \
        HD_M160_FETCH(0, ((I) + 1) % 18, 0)                                                                                \
        HD_M160_MFMA(1)                                                                                                    \
        HD_M160_ORDER()                                                                                                    \
      } else {                                                                                                             \
        HD_M160_FETCH(((I) + 1) & 1, ((I) + 1) % 18, 0)                                                                    \
        HD_M160_MFMA((I) & 1)                                                                                              \
        HD_M160_ORDER()                                                                                                    \
      }                                                                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                                   \
      TR_MARK(tr_mfma);                                                                                                    \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                   \
      __builtin_amdgcn_s_barrier();                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                                   \
      TR_MARK(tr_b2);                                                                                                      \
    }
    HD_M160_LOOP(HD_M160_CSTEP)
#undef HD_M160_CSTEP
#undef HD_M160_FETCH
#undef HD_M160_MFMA
#undef HD_M160_ORDER
    __builtin_amdgcn_s_setprio(0);
#ifdef HD_CONV_TRACE
    HD_TRACE(8, (unsigned long long)tr_mem);
    HD_TRACE(9, (unsigned long long)tr_b1);
    HD_TRACE(10, (unsigned long long)tr_mfma);
    HD_TRACE(11, (unsigned long long)tr_b2);
#endif
  }
#undef HD_M160_LOOP
#undef HD_M160_STEPS9
  __syncthreads();            // every fragment read and every DMA of the K loop is done: the ring becomes the epilogue tile
  HD_TRACE(4, clock64());

  // ---------------- epilogue: consumer accumulators -> LDS fp32 [2][160][68], then conv_w8_epilogue.h ----------------
  // C/D map of v_mfma_f32_16x16x32: column = lane & 15 = pixel of the block, row = 4 (lane >> 4) + i = channel: a lane's four
  // registers are four CONSECUTIVE output channels of one pixel -> one 16-byte LDS write per accumulator block; tile row r = pixel
  // (ty0 + r / 40, tx0 + r % 40); the pitch of 68 floats keeps the 8-lane write groups (8 consecutive rows) on distinct banks.
  float* ct = reinterpret_cast<float*>(lds);
  const int ks4 = (lane >> 4) * 4;
  if constexpr (TH == 4) {
    if (consumer) {              // partial tile wq = this consumer's half of every K step
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int row = (wm * 2 + (fp >> 3)) * TW + b * 8 + (fp & 7);
          *reinterpret_cast<f32x4*>(ct + (wq * BMH + row) * CP + c * 16 + ks4) = acc[b][c];
        }
    }
    hd_w8_epilogue<BMH, BN, 2, TW>(p, lds, n_img, ty0, tx0, n0, tile_m);
  } else {
    // 320 pixels, no partial sums: consumer wm lays tile rows 2 wm, 2 wm + 1 down in ONE fp32 tile [320][68] (85 KiB) and the shared
    // epilogue runs once over it (five row passes per thread, one BatchNorm row per block)
    if (consumer) {
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int row = (wm * 2 + (fp >> 3)) * TW + b * 8 + (fp & 7);
          *reinterpret_cast<f32x4*>(ct + row * CP + c * 16 + ks4) = acc[b][c];
        }
    }
    hd_w8_epilogue<2 * BMH, BN, 1, TW>(p, lds, n_img, ty0, tx0, n0, tile_m);
  }
  HD_TRACE(5, clock64());
  HD_TRACE(6, wall_clock64());
  HD_TRACE(7, hw_ids());
}

template <int TH, bool DUAL, int TW = 40>
__global__ __launch_bounds__(512, 2) void conv3x3_m160_kernel(ConvP p) {
  __shared__ __attribute__((aligned(1024))) f16 lds[M160<TH, TW>::LDS_HALVES];
  conv3x3_m160_body<TH, TW, DUAL>(p, lds, blockIdx.x, gridDim.x);
}

// Several 96-pixel-tile problems in ONE grid (the pyramid levels of an FPN / RPN-head convolution: 19 x 19, 10 x 10, 5 x 5 maps are 64 - 480
// blocks each, a fraction of the chip's 512 slots): blocks [first[i], first[i + 1]) run problem i exactly as its own launch would.
__global__ __launch_bounds__(512, 2) void conv3x3_m96_multi_kernel(ConvMulti mp) {
  __shared__ __attribute__((aligned(1024))) f16 lds[M160<4, 24>::LDS_HALVES];
  int pi = 0;
  for (int i = 1; i < mp.n; ++i)
    if ((int)blockIdx.x >= mp.first[i]) pi = i;
  ConvP p = mp.p[pi];
  conv3x3_m160_body<4, 24, false>(p, lds, (int)blockIdx.x - mp.first[pi], mp.first[pi + 1] - mp.first[pi]);
}

}  // namespace

// 3x3 / s1 / p1, same extent in and out, Cin % 64 == 0 (both sources of a decoder concat), NHWC f16 out, Cout % 8 == 0; an upsampled
// source only together with a skip source (the decoder's conv1).  (Any extent is computed correctly -- ragged tiles are masked -- the
// dispatcher only picks these tiles for maps they cover exactly.)
bool hd_conv_m160_eligible(const ConvP& p) {
  if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.in_dil != 1 || p.in_scale) return false;
  if (p.out_mode != HD_OUT_NHWC_F16 || (p.Cout & 7) != 0 || p.Ho != p.Hin || p.Wo != p.Win) return false;
  if ((p.xbytes | p.x2bytes | p.wbytes) & 0x80000000u) return false;      // out-of-range lanes are marked by bit 31 of the offset
  if (p.x2) return p.up1 && (p.C1 % 64) == 0 && (p.C2 % 64) == 0;
  return !p.up1 && (p.C1 % 64) == 0;
}

// out_pool2 on these tiles: plain data gradient, even extent (a 4 x 40 epilogue tile holds whole 2 x 2 blocks), the pooled channel count
// a multiple of the 64-channel tile so that no tile straddles the two halves
bool hd_conv_m160_pool2_ok(const ConvP& p) {
  return hd_conv_m160_eligible(p) && !p.res && !p.mask && !p.bias && !p.stats && !p.bs_y && p.act == HD_ACT_NONE && (p.Ho % 2) == 0 && (p.Wo % 2) == 0 &&
         p.pool2 > 0 && (p.pool2 % 64) == 0 && p.pool2 <= p.Cout && ((p.Cout - p.pool2) % 8) == 0 && (p.pool2 == p.Cout || p.y2 != nullptr);
}

// BatchNorm partial-sum rows: one per block
int hd_conv_m160_tiles(const ConvP& p, int th, int tw) { return p.N * hd_cdiv(p.Ho, th) * hd_cdiv(p.Wo, tw); }

void hd_conv_launch_m96_multi(ConvMulti& mp, hipStream_t s) {
  int total = 0;
  for (int i = 0; i < mp.n; ++i) {
    ConvP& p = mp.p[i];
    p.gm = p.N * hd_cdiv(p.Ho, 4) * hd_cdiv(p.Wo, 24);
    p.gn = hd_cdiv(p.Cout, BN);
    mp.first[i] = total;
    total += p.gm * p.gn;
  }
  mp.first[mp.n] = total;
  hipLaunchKernelGGL(conv3x3_m96_multi_kernel, dim3(total), dim3(512), 0, s, mp);
}

// (th, tw) in {(4, 40), (8, 40), (4, 24)}; the 24-wide tile takes single-source problems only
void hd_conv_launch_m160(ConvP& p, int th, int tw, hipStream_t s) {
  p.gm = p.N * hd_cdiv(p.Ho, th) * hd_cdiv(p.Wo, tw);
  p.gn = hd_cdiv(p.Cout, BN);
  dim3 grid(p.gm * p.gn);
  if (tw == 24) {
    hipLaunchKernelGGL((conv3x3_m160_kernel<4, false, 24>), grid, dim3(512), 0, s, p);
  } else if (th == 8) {
    if (p.x2) hipLaunchKernelGGL((conv3x3_m160_kernel<8, true>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((conv3x3_m160_kernel<8, false>), grid, dim3(512), 0, s, p);
  } else {
    if (p.x2) hipLaunchKernelGGL((conv3x3_m160_kernel<4, true>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((conv3x3_m160_kernel<4, false>), grid, dim3(512), 0, s, p);
  }
}
