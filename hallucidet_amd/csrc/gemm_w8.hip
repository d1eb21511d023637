// Large-tile GEMM for the box head's fully connected layers and the wide 1x1 convolutions (round 5):
//     y[M][N] = act( mask( x[M][K] . w[N][K]^T + bias[N] ) ),  fp16 in / fp32 accumulate / fp16 out,
// i.e. hd_conv2d problems whose im2col matrix IS the stored tensor (1x1 / stride 1, or a KHxKW window over a KHxKW map: fc6 =
// torchvision TwoMLPHead.fc6 over the flattened 7x7x256 RoI features [EXT], reached from src/utils/eval_forward_fasterrcnn.py:120-124).
//
// Why a second GEMM path.  Every operand of the implicit-GEMM family reaches the matrix cores through LDS, and a CU fills LDS at
// ~33 B/clk from L2 (16 from the Infinity Cache) however many waves issue the DMA (MI355X_MICROARCH "Indexed rows: gather into LDS";
// tools/probe_tile_order.py) while its MFMA pipes retire 4096 FLOP/clk: a BM x BN x 64 K step moves (BM + BN) * 128 bytes for
// BM * BN / 32 MFMA clocks, so 128x64 is fill-bound at 34 % MFMA utilisation, 128x128 at 52 %, 256x128 at 69 % and only 256x256
// balances the two.  The 4-wave kernels stop at 128x128 (fc6: 0.85 PFLOP/s forward, 0.44 data gradient = 6 272 tiles of 128x64, each
// re-filling 393 KB for 16 K steps).  This kernel: 256 x 128 tile on 8 waves (three 48-KB LDS stages) or 128 x 128 on 4 waves (two 32-KB
// stages, two blocks per CU), 64-deep K steps, filled by LDS-DMA exactly as conv_igemm_bk64.hip fills its (same piece shape, same source-side swizzle, zero-fill by
// out-of-range buffer offsets), one barrier per K step.
//   * The WEIGHTS are the MFMA A operand (rows of the 32x32 result = output channels), the activations the B operand (columns =
//     GEMM rows): a lane then owns 4 consecutive channels of one output row per register group, `v_permlane32_swap` pairs two groups
//     into 8 channels = one 16-byte store -- the epilogue stays in registers (no fp32 tile through LDS, no barrier), bias / ReLU
//     before the rounding, the ReLU-backward mask on the packed result (masking commutes with rounding).
//   * Products and their fp32 summation order over K are those of conv_igemm_bk64.hip (k = 64 kt + 16 ks + 8 h + j, one chain per
//     output), so the result is bit-identical to that family's -- which tile runs a problem may depend on the batch size
//     (tests/test_kernels_gpu.py::test_gemm_w8_matches_the_igemm_family_bit_for_bit).
//   * Tile list: XCD-contiguous runs in grouped order (conv_params.h: hd_conv_tile_order), the group sized from the operand bytes.
#include "hd_common.h"
#include "conv_params.h"

namespace {

constexpr int GK = 64;             // K step
constexpr int GROW = 64;           // halves per LDS row (128 bytes)
constexpr unsigned GOOB = 0xFFFFFFF0u;

typedef __attribute__((address_space(3))) void lds_void_g;

__device__ __forceinline__ void gdma16(__amdgpu_buffer_rsrc_t r, f16* lds_dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_g*)lds_dst, 16, voff, 0, 0, 0);
}

// BM: rows of x (GEMM M) per tile, BN: rows of w (output channels) per tile.  Waves = 2 (channels) x BM / 64 (rows): a wave owns
// BN / 2 channels x 64 rows -- 8 waves at BM = 256, 4 at BM = 128 (the 128 x 128 tile: two blocks per CU, for the short-K 1x1 layers
// whose time is the per-tile set-up and epilogue of the 4-wave family, not its K loop).
template <int BM, int BN, int NSTAGE>
__global__ __launch_bounds__(128 * (BM / 64)) void gemm_w8_kernel(ConvP p) {
  static_assert((BM == 256 || BM == 128) && BN == 128, "tile");
  constexpr int NWM = BM / 64, NT = 128 * NWM, RP = NT / 8;      // waves along the rows, threads, rows per DMA pass
  constexpr int NA = BN / 64;                    // 32-channel blocks per wave (MFMA A operand)
  constexpr int NB = 2;                          // 32-row blocks per wave (MFMA B operand)
  constexpr int STAGE = (BM + BN) * GROW;        // halves per stage: [BN weight rows][BM activation rows]
  constexpr int W_PASS = BN / RP, X_PASS = BM / RP;   // DMA instructions per wave per K step: RP rows (8 per wave) per pass
  constexpr int L_TILE = W_PASS + X_PASS;
  static_assert(NSTAGE * STAGE * 2 <= 160 * 1024, "LDS");
  static_assert(W_PASS >= 1 && X_PASS >= 1, "a DMA pass covers RP rows");
  __shared__ __attribute__((aligned(1024))) f16 lds[NSTAGE * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave / NWM, wm = wave - wn * NWM;

  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
    bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  }
  int tile_m, tile_n;
  hd_conv_tile_of(p, bid, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int K = p.Ktot;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.x), 0, p.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(p.w), 0, p.wbytes, 0x00020000);

  // ---- DMA addressing: thread -> (row tid >> 3 of a 64-row pass, 16-byte slot tid & 7); the lane that fills slot s of row r
  //      fetches logical chunk s ^ ((r >> 1) & 7) (source-side swizzle: the DMA writes LDS lane-linearly)
  const int j = (tid & 7) ^ ((tid >> 4) & 7);
  unsigned wbase[W_PASS], xbase[X_PASS];
  bool wok[W_PASS], xok[X_PASS];
#pragma unroll
  for (int i = 0; i < W_PASS; ++i) {
    const int row = n0 + i * RP + (tid >> 3);
    wok[i] = row < p.Cout;
    wbase[i] = (unsigned)(wok[i] ? row : 0) * (unsigned)K * 2u + (unsigned)j * 16u;
  }
#pragma unroll
  for (int i = 0; i < X_PASS; ++i) {
    const int row = m0 + i * RP + (tid >> 3);
    xok[i] = row < p.M;
    xbase[i] = (unsigned)(xok[i] ? row : 0) * (unsigned)K * 2u + (unsigned)j * 16u;
  }
  const int nk = (K + GK - 1) / GK;
  int kt_issue = 0;
  auto gload = [&](int stage) {
    f16* sw = lds + stage * STAGE + wave * (8 * GROW);
    f16* sx = sw + BN * GROW;
    const bool kv = kt_issue * GK + j * 8 < K;           // K % 8 == 0: a 16-byte chunk is inside or outside as a whole
    const unsigned ko = (unsigned)kt_issue * (GK * 2);
#pragma unroll
    for (int i = 0; i < W_PASS; ++i) gdma16(rw, sw + i * (RP * GROW), (wok[i] && kv) ? wbase[i] + ko : GOOB);
#pragma unroll
    for (int i = 0; i < X_PASS; ++i) gdma16(rx, sx + i * (RP * GROW), (xok[i] && kv) ? xbase[i] + ko : GOOB);
    ++kt_issue;
  };

  f32x16 acc[NA][NB];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int frow = lane & 31, fh = lane >> 5, swz = (frow >> 1) & 7;
  f16x8 wf[2][NA], xf[2][NB];
  auto frag = [&](int stage, int ks, int buf) {
    const f16* sw = lds + stage * STAGE;
    const f16* sx = sw + BN * GROW;
    const int slot = ((ks * 2 + fh) ^ swz) * 8;
#pragma unroll
    for (int a = 0; a < NA; ++a) wf[buf][a] = *reinterpret_cast<const f16x8*>(sw + (wn * (BN / 2) + a * 32 + frow) * GROW + slot);
#pragma unroll
    for (int b = 0; b < NB; ++b) xf[buf][b] = *reinterpret_cast<const f16x8*>(sx + (wm * 64 + b * 32 + frow) * GROW + slot);
  };

#pragma unroll
  for (int t = 0; t < NSTAGE - 1; ++t) gload(t);
  int rd = 0, wr = NSTAGE - 1;
  for (int kt = 0; kt < nk; ++kt) {
    // this wave's pieces of tile kt have landed once at most NSTAGE - 2 tiles' worth remain outstanding
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * L_TILE) : "memory");
    __builtin_amdgcn_s_barrier();                          // everyone's have; stage `wr` (read at kt - 1) is free
    __builtin_amdgcn_sched_barrier(0);
    frag(rd, 0, 0);
    frag(rd, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    // (the pieces issued one by one behind the MFMA groups instead: 220 -> 208 us on fc6 with the 256 x 128 tile, 286 -> 310 with
    //  256 x 256 -- inside the run-to-run spread of the first, a loss on the second; not kept)
    gload(wr);                                             // tile kt + NSTAGE - 1 (zero-fill past the end)
#pragma unroll
    for (int ks = 0; ks < GK / 16; ++ks) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks & 1][a], xf[ks & 1][b], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 2 < GK / 16) frag(rd, ks + 2, ks & 1);
    }
    rd = (rd + 1 == NSTAGE) ? 0 : rd + 1;
    wr = (wr + 1 == NSTAGE) ? 0 : wr + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the zero-fill tiles issued in the last iterations

  // ---------------- epilogue, registers only ----------------
  // acc[a][b][4 g + i] = channel n0 + wn * BN/2 + 32 a + 8 g + 4 h + i of row m0 + 64 wm + 32 b + (lane & 31)
  f16* __restrict__ yp = reinterpret_cast<f16*>(p.y);
  const f16* __restrict__ maskp = p.mask;
  const f16* __restrict__ resp = p.res;
  const float* __restrict__ biasp = p.bias;
  const bool relu = p.act == HD_ACT_RELU;
  const int N = p.Cout;
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    const int cb = n0 + wn * (BN / 2) + a * 32;
#pragma unroll
    for (int gp = 0; gp < 4; gp += 2) {
      f32x4 bv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      if (biasp) {
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int c = cb + 8 * (gp + gg) + 4 * fh;
          if (c < N) bv[gg] = *reinterpret_cast<const f32x4*>(biasp + c);       // N % 8 == 0: four channels in or out together
        }
      }
      const int co = cb + 8 * (gp + fh);                  // the 8 channels this lane stores after the exchange
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int m = m0 + wm * 64 + b * 32 + frow;
        const bool ok = m < p.M && co < N;
        unsigned pk[2][2];
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          float v[4];
          f16x4 rv = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
          if (resp) {                                   // the residual joins in fp32, before the bias (the igemm epilogue's order)
            const int c = cb + 8 * (gp + gg) + 4 * fh;
            if (m < p.M && c < N) rv = *reinterpret_cast<const f16x4*>(resp + (size_t)m * N + c);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = acc[a][b][4 * (gp + gg) + i];
            if (resp) v[i] += (float)rv[i];
            v[i] += bv[gg][i];
            if (relu) v[i] = fmaxf(v[i], 0.f);
          }
          const f16x2 o01 = {(f16)v[0], (f16)v[1]}, o23 = {(f16)v[2], (f16)v[3]};
          pk[gg][0] = __builtin_bit_cast(unsigned, o01);
          pk[gg][1] = __builtin_bit_cast(unsigned, o23);
        }
        const auto q0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
        const auto q1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
        u32x4 o = {q0[0], q1[0], q0[1], q1[1]};
        if (ok) {
          const size_t off = (size_t)m * N + co;
          if (maskp) {
            const f16x8 mv = *reinterpret_cast<const f16x8*>(maskp + off);
            f16x8 ov = __builtin_bit_cast(f16x8, o);
#pragma unroll
            for (int k = 0; k < 8; ++k) ov[k] = ((float)mv[k] > 0.f) ? ov[k] : (f16)0.f;
            o = __builtin_bit_cast(u32x4, ov);
          }
          *reinterpret_cast<u32x4*>(yp + off) = o;
        }
      }
    }
  }
}

}  // namespace

// Is this hd_conv2d problem a plain GEMM over stored tensors that the large-tile kernel implements?
bool hd_gemm_w8_eligible(const ConvP& p) {
  const bool flat = (p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0) ||
                    (p.KH == p.Hsrc && p.KW == p.Wsrc && p.pad == 0 && p.Ho == 1 && p.Wo == 1);
  return flat && !p.x2 && !p.up1 && p.in_dil == 1 && !p.stats && !p.in_scale && !p.bs_y && p.out_mode == HD_OUT_NHWC_F16 &&
         (p.act == HD_ACT_NONE || p.act == HD_ACT_RELU) && (p.Cout % 8) == 0 && (p.Ktot % 64) == 0 && (p.Hin == p.Hsrc && p.Win == p.Wsrc);
}

// tile: 128 = 256 x 128 (rows x channels, 8 waves), 1128 = 128 x 128 (4 waves, two blocks per CU)
void hd_gemm_w8_launch(ConvP& p, int tile, hipStream_t s) {
  const int bm = tile == 1128 ? 128 : 256;
  p.gm = hd_cdiv(p.M, bm);
  p.gn = hd_cdiv(p.Cout, 128);
  p.tgroup = hd_conv_tile_order(p);
  if (tile == 128) hipLaunchKernelGGL((gemm_w8_kernel<256, 128, 3>), dim3(p.gm * p.gn), dim3(512), 0, s, p);
  else hipLaunchKernelGGL((gemm_w8_kernel<128, 128, 2>), dim3(p.gm * p.gn), dim3(256), 0, s, p);
}
