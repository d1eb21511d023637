// 3x3 / stride 1 / pad 1 convolution, 64 -> 64 channels, with the WEIGHTS RESIDENT IN REGISTERS (gfx950).
//
// The layers: ResNet-34 layer1 of the hallucination network at 128x160 (six forward convolutions with BatchNorm partial sums, their data
// gradients with the residual add; reference src/segmentation_models/encoders/resnet.py:47-65 via torchvision BasicBlock [EXT]) and the
// 3x3 of the detector's layer1 bottlenecks at 75x75 (bias + ReLU; data gradient with the ReLU mask).  M is huge (163 840 pixels), N = 64,
// K = 576: in the implicit-GEMM family a 128 x 64 tile lives for nine K steps, and its set-up + LDS epilogue cost more than its MFMAs
// (27-30 us per launch against a 5 us floor).  Two earlier attempts kept the weight matrix in LDS (49 us: 1.5 KiB of fragment reads per
// MFMA, LDS-bound) -- here the 64 x 576 matrix is the MFMA *A* operand and never leaves the register file:
//   * 4 waves per block, ONE wave per SIMD, one persistent block per CU walking a contiguous run of 8 x 16-pixel tiles;
//   * a wave owns 32 pixels (two tile rows) x all 64 output channels: 2 x 36 A fragments = 288 VGPRs of weights, loaded once per block
//     (LDS-DMA of the matrix as it lies in memory, source-side swizzle, 72 conflict-free ds_read_b128);
//   * per 16-deep K step ONE 1-KiB fragment read (the pixels, B operand) feeds TWO v_mfma_f32_32x32x16_f16: 144 KiB of LDS reads
//     per 2 304 MFMA clocks of a tile (the LDS port would carry 288);
//   * the (8+2) x (16+2) input patch of the NEXT tile arrives by LDS-DMA during the K loop (two stages, hardware zero-fill for the
//     padding), swizzle slot ^= (x >> 1) & 7 on the source side: conflict-free for all three kw under ds_read_b128's 16-lane groups
//     because the patch pitch (18 pixels) is even, so a pixel's 128-byte row alternates between the two halves of the 256-byte bank row
//     with x;
//   * C[cout][pixel]: a lane ends up with 4 consecutive output channels of one pixel per accumulator quad -> the epilogue runs in
//     registers (residual / bias / ReLU mask / activation, 8-byte stores), no LDS transpose, no barrier;
//   * BatchNorm partial sums accumulate in registers over all tiles of the block and are folded ONCE per block (LDS transpose, fixed
//     order: deterministic); one partial row per block.
// K order is tap-major, 16 channels at a time -- the order the implicit-GEMM kernels accumulate in.
#include "hd_common.h"
#include "conv_params.h"
#include <stdlib.h>

namespace {

constexpr int TH = 8, TW = 16, PH = TH + 2, PW = TW + 2;
constexpr int NPIX = PH * PW;                    // 180 patch pixels of 128 bytes
constexpr int PIECES = 6;                        // 1-KiB DMA pieces per wave per patch (4 x 6 = 24 >= 22.5)
constexpr int STAGE_BYTES = 4 * PIECES * 1024;   // 24 KiB
constexpr int KTOT = 576, KSTEPS = 36;
constexpr int W_OFF = 2 * STAGE_BYTES;           // weight staging (72 KiB), later the statistics transpose
constexpr int W_BYTES = 64 * KTOT * 2;
constexpr int LDS_BYTES = W_OFF + W_BYTES;       // 120 KiB: one block per CU
constexpr int CF_OFF = W_OFF + 4 * 64 * 65 * 4;  // per-channel coefficients of the bs_* modes: behind the statistics transpose, inside the weight staging
constexpr int BIAS_OFF = CF_OFF + 4 * 64 * 4;    // the bias vector (EPI & 2): read per tile from LDS, not from memory
static_assert(BIAS_OFF + 64 * 4 <= LDS_BYTES, "coefficient table and bias fit");
constexpr unsigned OOBB = 0x80000000u;

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds_dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds_dst, 16, voff, 0, 0, 0);
}

// The MFMAs are inline assembly so that the weight fragments can be PINNED in the accumulator file ("a" operands: 64 of the 72 fragments
// = all 256 AGPRs; the last 8 stay in VGPRs): left to the register allocator they were spilled to AGPRs and copied back with four
// v_accvgpr_read per MFMA (K loop 3 800 clocks per tile without any DMA, 8 000 in the statistics variant; MFMA work: 2 304).
// What the compiler therefore does not see and the code provides: the wait states between the last MFMA of a tile and the first VALU
// read of its accumulators (HD_C64_DRAIN); an accumulator that is both SrcC and vDst of back-to-back MFMAs needs none.
#define HD_C64_MFMA0_A(ACC, WF, BF) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(ACC) : "a"(WF), "v"(BF))
#define HD_C64_MFMA_A(ACC, WF, BF) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "a"(WF), "v"(BF))
#define HD_C64_MFMA_V(ACC, WF, BF) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(WF), "v"(BF))
#define HD_C64_DRAIN(A0, A1) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(A0), "+v"(A1))

constexpr int RING = 6;        // B fragments in flight ahead of the MFMAs
constexpr int NA = 32;         // K steps whose two weight fragments live in AGPRs (2 x 32 x 4 = 256 registers)

// EPI: 1 residual, 2 bias, 4 ReLU mask, 8 ReLU (template: as run-time flags the epilogue was 1 200 instructions with ~60 branches and
// took 3 100 clocks per tile without a single store)
// STATS: 0 none | 1 BatchNorm forward sums (sum y, sum y^2) | 2, 3 hd_conv_args.bs_*: the output is the gradient dz of a BatchNorm unit and
// its BACKWARD sums (sum dz*m, sum dz*m*xhat) leave through the same rows -- ReLU mask m recomputed from the unit's y (2) or read from
// its saved activation z (3); per-channel coefficients sit in the dead weight-staging area of LDS
template <int STATS, int EPI>
__global__ __launch_bounds__(256) void conv3x3_c64_kernel(ConvP p, int tiles_total) {
  __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, pl = lane & 31;
  HD_TRACE(0, wall_clock64());
  HD_TRACE(1, clock64());
#ifdef HD_CONV_TRACE
  long long tr_bar = 0, tr_k = 0, tr_wait = 0, tr_epi = 0, tr_t = 0;
#define TR_MARK(accu) do { long long n_ = clock64(); accu += n_ - tr_t; tr_t = n_; } while (0)
#else
#define TR_MARK(accu) do {} while (0)
#endif

  // blocks are dealt round-robin over the 8 XCDs: give each XCD a contiguous eighth of the tile list and each block a contiguous run
  // (neighbouring tiles share their halo columns in that XCD's L2)
  const int G = gridDim.x;
  int L;
  {
    const int b = blockIdx.x, xcd = b & 7, qq = G >> 3, rr = G & 7;
    L = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (b >> 3);
  }
  const int t_begin = (int)((long long)L * tiles_total / G), t_end = (int)((long long)(L + 1) * tiles_total / G);
  const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
  const int H = p.Hin, W = p.Win;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.x), 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.w), 0, p.wbytes, 0x00020000);

  // ---- patch fill tables: unit u = (k*4 + wave)*64 + lane is 16-byte slot (u & 7) of patch pixel u >> 3
  int rel[PIECES], pyx[PIECES];
#pragma unroll
  for (int k = 0; k < PIECES; ++k) {
    const int u = (k * 4 + wave) * 64 + lane;
    const int pp = u >> 3, slot = u & 7;
    const int py = (pp * 3641) >> 16, px = pp - py * PW;          // pp / 18 (exact for pp < 2 000)
    const int cg = slot ^ ((px >> 1) & 7);
    rel[k] = ((py - 1) * W + (px - 1)) * 128 + cg * 16;
    pyx[k] = pp < NPIX ? (py | (px << 8)) : 0x4000;                // bit 14: not a patch pixel
  }
  // tile t -> (image, tile row, tile column); uniform
  auto tile_pos = [&](int t, int& n, int& ty, int& tx) {
    const int r1 = t / tiles_x;
    tx = t - r1 * tiles_x;
    n = r1 / tiles_y;
    ty = r1 - n * tiles_y;
  };
  auto issue_patch = [&](int n, int ty, int tx, int stage, int k) {
    const int base = ((n * H + ty * TH) * W + tx * TW) * 128;
    const int py = pyx[k] & 0xff, px = (pyx[k] >> 8) & 0x3f;
    const int iy = ty * TH - 1 + py, ix = tx * TW - 1 + px;
    const bool ok = !(pyx[k] & 0x4000) && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    dma16(rx, lds + stage * STAGE_BYTES + (k * 4 + wave) * 1024, ok ? (unsigned)(base + rel[k]) : OOBB);
  };

  // ---- prologue: first patch + the weight matrix in flight together
  int cn = 0, cty = 0, ctx = 0;
  if (t_begin < t_end) {
    tile_pos(t_begin, cn, cty, ctx);
#pragma unroll
    for (int k = 0; k < PIECES; ++k) issue_patch(cn, cty, ctx, 0, k);
  }
#pragma unroll
  for (int k = 0; k < 18; ++k) {
    const int u = (k * 4 + wave) * 64 + lane;                    // 16-byte chunk index of the [64][576] matrix
    const int row = (u * 58255) >> 22, j = u - row * 72;          // u / 72 (exact for u < 4 608)
    const int src = (j & ~7) | ((j & 7) ^ ((row >> 1) & 7));
    dma16(rw, lds + W_OFF + (k * 4 + wave) * 1024, (unsigned)(row * (KTOT * 2) + src * 16));
  }
  HD_TRACE(2, clock64());
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  HD_TRACE(3, clock64());
  f16x8 wr[2][KSTEPS];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int row = 32 * b + pl, q = 2 * s + h;
      const int slot = (q & ~7) | ((q & 7) ^ ((row >> 1) & 7));
      wr[b][s] = *reinterpret_cast<const f16x8*>(lds + W_OFF + row * (KTOT * 2) + slot * 16);
    }

  if (EPI & 2) {
    __syncthreads();                                    // every wave has its weight fragments: the staging area is free
    if (tid < 64) reinterpret_cast<float*>(lds + BIAS_OFF)[tid] = p.bias[tid];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (ordered against the readers by the tile loop's first barrier)
  }
  if (STATS >= 2) {
    __syncthreads();                                    // every wave has its weight fragments: the staging area is free
    float* cf = reinterpret_cast<float*>(lds + CF_OFF);  // [mean | invstd | scale | shift][64]
    if (tid < 64) {
      const float mu = p.bs_mean[tid], is = p.bs_invstd[tid];
      const float g = p.bs_gamma ? p.bs_gamma[tid] : 1.f, b = p.bs_beta ? p.bs_beta[tid] : 0.f;
      cf[tid] = mu;
      cf[64 + tid] = is;
      cf[128 + tid] = g * is;
      cf[192 + tid] = b - mu * g * is;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (ordered against the readers by the tile loop's first barrier)
  }

  // ---- B fragment addresses: this lane's pixel (two tile rows per wave) at tap (kh, kw), 16-channel group c, half h
  const int y0l = 2 * wave + ((lane >> 4) & 1), x0l = lane & 15;
  int ab[3][4];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int xk = x0l + kw;
#pragma unroll
    for (int c = 0; c < 4; ++c) ab[kw][c] = (y0l * PW + xk) * 128 + (((2 * c + h) ^ ((xk >> 1) & 7)) & 7) * 16;
  }

  float s1[2][16], s2[2][16];
  if (STATS != 0) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) s1[b][r] = s2[b][r] = 0.f;
  }
  HD_TRACE(4, clock64());
#ifdef HD_CONV_TRACE
  tr_t = clock64();
#endif
  const f16* __restrict__ resp = p.res;
  const f16* __restrict__ maskp = p.mask;
  const float* __restrict__ biasp = p.bias;
  f16* __restrict__ yp = reinterpret_cast<f16*>(p.y);

  for (int t = t_begin; t < t_end; ++t) {
    const int stage = (t - t_begin) & 1;
    // every wave has retired its pieces of tile t (vmcnt(0) below / in the prologue) and is done reading the other stage
    __builtin_amdgcn_s_barrier();
    TR_MARK(tr_bar);
    const bool more = t + 1 < t_end;
    int nn = 0, nty = 0, ntx = 0;
    if (more) tile_pos(t + 1, nn, nty, ntx);
    const char* sb = lds + stage * STAGE_BYTES;
    // bs_* modes: the unit's y (and z) vectors of this tile are requested BEFORE the K loop -- after it they were an exposed HBM round trip
    // per tile on a kernel that runs one wave per SIMD (the 64-channel data gradients read 31.7 us with the sums against 17.7 without)
    f16x4 byv[8], bzv[8], rv[8], mv[8];
    if (EPI & 4) {          // the ReLU-mask operand likewise
      const int oy_ = cty * TH + y0l, ox_ = ctx * TW + x0l;
      const bool okp_ = oy_ < p.Ho && ox_ < p.Wo;
      const unsigned eoff_ = (unsigned)(((cn * p.Ho + oy_) * p.Wo + ox_) * 64);
#pragma unroll
      for (int j = 0; j < 8; ++j) mv[j] = okp_ ? *reinterpret_cast<const f16x4*>(maskp + eoff_ + 8 * j + 4 * h) : (f16x4){0, 0, 0, 0};
    }
    if (EPI & 1) {          // the residual operand too (requested after the K loop it cost the residual variants 7 us per launch)
      const int oy_ = cty * TH + y0l, ox_ = ctx * TW + x0l;
      const bool okp_ = oy_ < p.Ho && ox_ < p.Wo;
      const unsigned eoff_ = (unsigned)(((cn * p.Ho + oy_) * p.Wo + ox_) * 64);
#pragma unroll
      for (int j = 0; j < 8; ++j) rv[j] = okp_ ? *reinterpret_cast<const f16x4*>(resp + eoff_ + 8 * j + 4 * h) : (f16x4){0, 0, 0, 0};
    }
    if (STATS >= 2) {
      const int oy_ = cty * TH + y0l, ox_ = ctx * TW + x0l;
      const bool okp_ = oy_ < p.Ho && ox_ < p.Wo;
      const unsigned eoff_ = (unsigned)(((cn * p.Ho + oy_) * p.Wo + ox_) * 64);
#pragma unroll
      for (int j = 0; j < 8; ++j) byv[j] = okp_ ? *reinterpret_cast<const f16x4*>(p.bs_y + eoff_ + 8 * j + 4 * h) : (f16x4){0, 0, 0, 0};
      if (STATS == 3) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bzv[j] = okp_ ? *reinterpret_cast<const f16x4*>(p.bs_z + eoff_ + 8 * j + 4 * h) : (f16x4){0, 0, 0, 0};
      }
    }
    f32x16 acc0, acc1;
    f16x8 bf[RING];
#define HD_C64_B(S) (*reinterpret_cast<const f16x8*>(sb + ab[((S) >> 2) % 3][(S) & 3] + ((S) / 12) * (PW * 128)))
#pragma unroll
    for (int s = 0; s < RING; ++s) bf[s] = HD_C64_B(s);
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      const f16x8 cur = bf[s % RING];
      if (s == 0) {
        HD_C64_MFMA0_A(acc0, wr[0][s], cur);
        HD_C64_MFMA0_A(acc1, wr[1][s], cur);
      } else if (s < NA) {
        HD_C64_MFMA_A(acc0, wr[0][s], cur);
        HD_C64_MFMA_A(acc1, wr[1][s], cur);
      } else {
        HD_C64_MFMA_V(acc0, wr[0][s], cur);
        HD_C64_MFMA_V(acc1, wr[1][s], cur);
      }
      if (s + RING < KSTEPS) bf[s % RING] = HD_C64_B(s + RING);
      // the next tile's patch: one piece every six K steps (issued back to back a piece cost this wave ~150 clocks, spread out ~70)
      if (s % 6 == 1 && more) issue_patch(nn, nty, ntx, stage ^ 1, s / 6);
    }
#undef HD_C64_B
    HD_C64_DRAIN(acc0, acc1);
    __builtin_amdgcn_sched_barrier(0);
    TR_MARK(tr_k);
    // the next tile's pieces: the last one was issued five K steps ago, nothing younger than them is outstanding
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TR_MARK(tr_wait);

    // ---- epilogue in registers: acc{b}[4g + i] = channel 32b + 8g + 4h + i of this lane's pixel
    const int oy = cty * TH + y0l, ox = ctx * TW + x0l;
    const bool okp = oy < p.Ho && ox < p.Wo;
    const unsigned eoff = (unsigned)(((cn * p.Ho + oy) * p.Wo + ox) * 64);          // < 2^30 elements (eligibility)
    f32x4 bv[8];
    if (EPI & 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) bv[j] = *reinterpret_cast<const f32x4*>(lds + BIAS_OFF + (8 * j + 4 * h) * 4);
    }
    // rv / y / z vectors of the whole tile are requested up front; the packed output of a group pair is exchanged and stored as soon
    // as it exists (a separate store pass would keep all sixteen packed registers alive next to the resident weights)
    // STATS 2: (scale, shift) of this lane's channels for the ReLU mask, re-read per tile through an offset the compiler cannot see
    // through (hoisted out of the tile loop the 64 floats push weight fragments into scratch)
    int tile_zero = 0;
    if (STATS == 2) asm volatile("v_mov_b32 %0, 0" : "=v"(tile_zero));
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int gp = 0; gp < 4; gp += 2) {
        unsigned pk[2][2];                 // groups gp, gp + 1: four channels each as two packed f16 pairs
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int g = gp + gg, j = 4 * b + g;
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = b ? acc1[4 * g + i] : acc0[4 * g + i];
          if (EPI & 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += (float)rv[j][i];
          }
          if (EPI & 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] += bv[j][i];
          }
          if (EPI & 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ((float)mv[j][i] > 0.f) ? v[i] : 0.f;
          }
          if (STATS == 1) {
            const float keep = okp ? 1.f : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float vr = (float)(f16)v[i] * keep;
              s1[b][4 * g + i] += vr;
              s2[b][4 * g + i] += vr * vr;
            }
          }
          if (STATS >= 2) {
            // sum dz*m and sum dz*m*y of this lane's channels 8j + 4h .. + 3 on the fp16-rounded dz; sum dz*m*xhat =
            // invstd * (sum dz*m*y - mean * sum dz*m) is formed once per block below (no per-channel mean / invstd in this loop)
            f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc;
            if (STATS == 2) {
              const float* cf = reinterpret_cast<const float*>(lds + CF_OFF + tile_zero) + 8 * j + 4 * h;
              sc = *reinterpret_cast<const f32x4*>(cf + 128);
              sh = *reinterpret_cast<const f32x4*>(cf + 192);
            }
            const float keep = okp ? 1.f : 0.f;
            const bool relu = p.bs_relu != 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float gk = (float)(f16)v[i] * keep;
              const float yk = (float)byv[j][i];
              const bool on = STATS == 3 ? ((float)bzv[j][i] > 0.f) : ((float)(f16)hd_bn_affine(yk, sc[i], sh[i]) > 0.f);
              gk = (on || !relu) ? gk : 0.f;
              s1[b][4 * g + i] += gk;
              s2[b][4 * g + i] += gk * yk;
            }
          }
          if (EPI & 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
          }
          const f16x2 o01 = {(f16)v[0], (f16)v[1]}, o23 = {(f16)v[2], (f16)v[3]};
          pk[gg][0] = __builtin_bit_cast(unsigned, o01);
          pk[gg][1] = __builtin_bit_cast(unsigned, o23);
        }
        // 16-byte stores: lanes l and l + 32 hold the two 4-channel halves of the same pixel's 8-channel groups; one half exchange
        // (v_permlane32_swap) per register leaves lane l with all of group g, lane l + 32 with all of group g + 1
        const auto q0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
        const auto q1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
        const u32x4 o = {q0[0], q1[0], q0[1], q1[1]};
        if (okp) *reinterpret_cast<u32x4*>(yp + eoff + 32 * b + 8 * (gp + h)) = o;
      }
    cn = nn; cty = nty; ctx = ntx;
    __builtin_amdgcn_sched_barrier(0);
    TR_MARK(tr_epi);
  }
  HD_TRACE(5, clock64());
#ifdef HD_CONV_TRACE
  HD_TRACE(8, (unsigned long long)tr_bar);
  HD_TRACE(9, (unsigned long long)tr_k);
  HD_TRACE(10, (unsigned long long)tr_wait);
  HD_TRACE(11, (unsigned long long)tr_epi);
  HD_TRACE(12, (unsigned long long)(t_end - t_begin));
#endif

  if (STATS != 0) {
    // ---- one partial row per block: transpose the 64 per-lane sums of a wave through LDS (pitch 65 floats: conflict-free both ways),
    //      lane j adds value j over the 32 pixel lanes of each half, then 128 threads add the four waves -- fixed order throughout
    __syncthreads();                                   // patch stages and weight staging are dead
    float* red = reinterpret_cast<float*>(lds + W_OFF) + wave * (64 * 65);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        red[(b * 16 + r) * 65 + lane] = s1[b][r];
        red[(32 + b * 16 + r) * 65 + lane] = s2[b][r];
      }
    __syncthreads();
    float a0 = 0.f, a1 = 0.f;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
      a0 += red[lane * 65 + i];
      a1 += red[lane * 65 + 32 + i];
    }
    float* red2 = reinterpret_cast<float*>(lds);       // [wave][value][half]
    red2[(wave * 64 + lane) * 2 + 0] = a0;
    red2[(wave * 64 + lane) * 2 + 1] = a1;
    __syncthreads();
    if (tid < 128) {
      const int v = tid >> 1, hh = tid & 1;
      float s = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) s += red2[(w4 * 64 + v) * 2 + hh];
      const int which = v >> 5, b = (v >> 4) & 1, r = v & 15;
      const int c = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * hh;
      if (STATS >= 2 && which == 1) {
        // s = sum dz*m*y of channel c; its partner sum dz*m sits 32 values below in red2
        float s0 = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) s0 += red2[(w4 * 64 + v - 32) * 2 + hh];
        const float* cf = reinterpret_cast<const float*>(lds + CF_OFF);
        s = cf[64 + c] * (s - cf[c] * s0);
      }
      p.stats[((size_t)blockIdx.x * 2 + which) * 64 + c] = s;
    }
  }
  HD_TRACE(13, clock64());
  HD_TRACE(6, wall_clock64());
  HD_TRACE(7, hw_ids());
}

}  // namespace

// 3x3 / s1 / p1, one 64-channel source, 64 output channels, NHWC f16 out, same extent in and out, no sigmoid.  (Nothing here may depend
// on p.stats: hd_conv2d_stats_rows asks before the slab exists.)
bool hd_conv_c64_eligible(const ConvP& p) {
  if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.in_dil != 1 || p.up1) return false;
  if (p.C1 != 64 || p.C2 != 0 || p.x2 || p.Cout != 64 || p.out_mode != HD_OUT_NHWC_F16) return false;
  if (p.Ho != p.Hin || p.Wo != p.Win || p.Hsrc != p.Hin || p.Wsrc != p.Win || p.in_scale) return false;
  if (p.act != HD_ACT_NONE && p.act != HD_ACT_RELU) return false;
  if ((p.xbytes | p.wbytes) & 0xC0000000u) return false;      // out-of-range lanes are marked by bit 31 of the offset; element offsets fit 30 bits
  return true;
}

static int c64_tiles(const ConvP& p) { return p.N * hd_cdiv(p.Ho, TH) * hd_cdiv(p.Wo, TW); }

// partial-sum rows (= blocks) of this problem
int hd_conv_c64_rows(const ConvP& p) {
  const int t = c64_tiles(p);
  return t < 256 ? t : 256;
}

void hd_conv_launch_c64(ConvP& p, hipStream_t s) {
  const int tiles = c64_tiles(p);
  dim3 grid(hd_conv_c64_rows(p));
  const int epi = (p.res ? 1 : 0) | (p.bias ? 2 : 0) | (p.mask ? 4 : 0) | (p.act == HD_ACT_RELU ? 8 : 0);
  if (p.bs_y) {          // BatchNorm backward sums (data gradients: plain or + residual; fill_params excludes bias / mask / act)
    if (p.bs_z) {
      if (epi & 1) hipLaunchKernelGGL((conv3x3_c64_kernel<3, 1>), grid, dim3(256), 0, s, p, tiles);
      else hipLaunchKernelGGL((conv3x3_c64_kernel<3, 0>), grid, dim3(256), 0, s, p, tiles);
    } else {
      if (epi & 1) hipLaunchKernelGGL((conv3x3_c64_kernel<2, 1>), grid, dim3(256), 0, s, p, tiles);
      else hipLaunchKernelGGL((conv3x3_c64_kernel<2, 0>), grid, dim3(256), 0, s, p, tiles);
    }
    return;
  }
#define LAUNCH(E) case E: if (p.stats) hipLaunchKernelGGL((conv3x3_c64_kernel<1, E>), grid, dim3(256), 0, s, p, tiles); \
                          else hipLaunchKernelGGL((conv3x3_c64_kernel<0, E>), grid, dim3(256), 0, s, p, tiles); break
  switch (epi) {
    LAUNCH(0); LAUNCH(1); LAUNCH(2); LAUNCH(3); LAUNCH(4); LAUNCH(5); LAUNCH(6); LAUNCH(7);
    LAUNCH(8); LAUNCH(9); LAUNCH(10); LAUNCH(11); LAUNCH(12); LAUNCH(13); LAUNCH(14); LAUNCH(15);
  }
#undef LAUNCH
}
