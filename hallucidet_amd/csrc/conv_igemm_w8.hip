// Implicit-GEMM convolution, 8-wave family (512 threads, one block per CU, two waves per SIMD).
//
// Why a second family: the 4-wave kernels (conv_igemm_bk32/bk64.hip) fill LDS at the rate of the CU's vector-memory
// path, 64 B/clk (one 1-KiB LDS-DMA piece per 16 clocks at best).  A BM x BN x 64 K step needs (BM + BN) * 128 B of
// fill for BM*BN*64*2 / 4096 clocks of MFMA work:
//     64 x 64: 128 B/clk (2.0x the path)   128 x 128: 64 B/clk (1.0x)   256 x 128: 48 B/clk (0.75x)   256 x 256: 32 B/clk
// so the small tiles the 4-wave dispatcher picks to fill 256 CUs can never pass 35-50 % of the MFMA roof.  Here:
//   * every wave owns a 64 x 64 accumulator tile (2 x 2 MFMA 32x32x16 tiles: one fragment read per MFMA, 128 B/clk of
//     LDS reads per CU = half the LDS rate), eight waves form a WM x WN x WK grid:
//         WM x WN     = block tile in units of 64 rows / 64 columns (256x128, 128x128, 256x64, 128x64, 128x256, ...)
//         WK in 1,2,4 = INTRA-K-STEP split: the 64-deep K step has four 16-deep MFMA sub-steps, wave group wk takes
//                       4/WK of them.  All eight waves work on every K step (fill shared by twice / four times the
//                       MFMA waves), partial accumulators meet in the epilogue's LDS tile.  This is what lets a
//                       128 x 128 or 128 x 64 block tile run with 8 waves at one fragment read per MFMA;
//   * operands global -> LDS by LDS-DMA, NSTAGE-deep ring, counted vmcnt + one raw s_barrier per K step (as bk64);
//     each wave's (WM + WN) DMA pieces of a K step are issued BETWEEN its MFMA sub-steps, not in one burst in front
//     of them: a burst queues behind the other seven waves' bursts at the texture addresser (100-185 clocks per piece
//     measured), spread out a piece costs the issuing wave ~60;
//   * split-K over blocks (gridDim.y > 1) for launches whose M x N extent cannot give every CU a tile: each slice
//     accumulates its share of the K steps and stores a raw fp32 partial tile into a workspace slab; the LAST block to
//     arrive for a tile (agent-scope ticket, release/acquire as /opt/skills/guides/cdna_hip_programming.md 5 describes)
//     sums the slabs and runs the ordinary epilogue -- no second launch;
//   * epilogue through one fp32 LDS tile [WK][BM][BN] (128 KiB for every shape of the family), 16-byte stores.
// Same gather as bk64 (nearest-2x upsample + concat, zero-dilated data gradients, hardware zero fill for padding and
// ragged tails) and the same output contract (bias, residual, ReLU mask, activation, BN partial sums per M tile).
// NHWC f16 outputs with Cout % 8 == 0 only; everything else stays on the 4-wave family.
#include "hd_common.h"
#include "conv_params.h"

namespace {

constexpr int BK = 64;
constexpr int LDS_ROW = 64;   // halves per LDS row (128 B)
constexpr int CPT = 8;        // 16-byte chunks per row per K step
constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int NT_ = 512;      // threads

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, f16* lds_dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds_dst, 16, voff, 0, 0, 0);
}

// ADDR: 0 = SIMPLE (one source, no upsample / dilation, Cin % 64 == 0: a row's address at a tap is its tap-(0,0) address plus
//          a block-uniform tap offset, validity is one bit of a per-row tap mask: 3 VALU per row per K step),
//       1 = general uniform-tap walk (upsample / concat / zero-dilated sources, Cin % 64 == 0), 2 = per-lane taps (any Cin % 8)
template <int WM, int WN, int WK, bool DUAL, int ADDR, int NSTAGE>
__global__ __launch_bounds__(512, 2) void conv_w8_kernel(ConvP p) {
  static_assert(WM * WN * WK == 8, "eight waves");
  constexpr bool KGEN = ADDR == 2;
  constexpr bool SIMPLE = ADDR == 0;
  constexpr int BM = WM * 64, BN = WN * 64;
  constexpr int KSP = 4 / WK;                        // 16-deep MFMA sub-steps per wave per K step
  constexpr int A_LOADS = BM / 64;                   // DMA pieces per wave per K step (8 rows x 128 B each)
  constexpr int B_LOADS = BN / 64;
  constexpr int L_TILE = A_LOADS + B_LOADS;
  constexpr int STAGE = (BM + BN) * LDS_ROW;         // halves per stage
  constexpr int EPI_HALVES = WK * BM * BN * 2;       // fp32 [WK][BM][BN]
  constexpr int LDS_HALVES = NSTAGE * STAGE > EPI_HALVES ? NSTAGE * STAGE : EPI_HALVES;
  __shared__ __attribute__((aligned(1024))) f16 lds[LDS_HALVES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave / (WM * WN);
  const int wmn = wave - wk * (WM * WN);
  const int wm = wmn / WN, wn = wmn - wm * WN;
  HD_TRACE(0, wall_clock64());
  HD_TRACE(1, clock64());
  const int grp = wave >> 2;      // ping-pong group: waves w and w+4 share a SIMD (waves are dealt to the 4 SIMDs cyclically)

  // XCD-aware bijective tile map (blocks are dealt round-robin over the 8 XCDs): XCD x gets the x-th contiguous eighth
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

  // ---- per-thread A rows: row (tid>>3) + i*64
  int hb[A_LOADS], wb[A_LOADS];
  unsigned nb1[A_LOADS], nb2[A_LOADS];
  bool rvalid[A_LOADS];
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    int pix = m0 + (tid >> 3) + i * 64;
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
  unsigned wbase[B_LOADS];
  bool wvalid[B_LOADS];
#pragma unroll
  for (int i = 0; i < B_LOADS; ++i) {
    int co = n0 + (tid >> 3) + i * 64;
    wvalid[i] = co < p.Cout;
    wbase[i] = (unsigned)(wvalid[i] ? co : 0) * (unsigned)p.Ktot * 2u;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const bool dil2 = p.in_dil == 2;
  const bool up1 = p.up1 != 0;

  // this block's K steps (split-K over gridDim.y)
  const int nsl = gridDim.y, sl = blockIdx.y;
  const int kt_begin = (int)(((long long)p.nk * sl) / nsl), kt_end = (int)(((long long)p.nk * (sl + 1)) / nsl);
  const int nkt = kt_end - kt_begin;

  int kt_issue = kt_begin;
  int kh_u = 0, kw_u = 0, c8_u = 0;
  unsigned po1[A_LOADS], po2[A_LOADS];
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
  unsigned rb[A_LOADS];                 // SIMPLE: byte offset of (row's pixel at tap (0,0), channel 0); wraps when out of range
  unsigned long long vmask[A_LOADS];    // SIMPLE: bit t = tap t in range for this row (shifted as the walk advances)
  int tapoff = 0;                       // SIMPLE: block-uniform byte offset of the current tap relative to tap (0,0)
  if (!KGEN) {
    // uniform tap walk starts at K step kt_begin
    const int c8_total = kt_begin * CPT;
    const int tap0 = c8_total / p.cin8;
    c8_u = c8_total - tap0 * p.cin8;
    kh_u = tap0 / p.KW;
    kw_u = tap0 - kh_u * p.KW;
    if (SIMPLE) {
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) {
        rb[i] = nb1[i] + (unsigned)((hb[i] * p.Wsrc + wb[i]) * p.C1) * 2u;
        unsigned long long m = 0ull;
        const int ntap = p.KH * p.KW;
        for (int t = ntap - 1; t >= tap0; --t) {
          const int kh = t / p.KW, kw = t - kh * p.KW;
          const bool v = rvalid[i] && ((unsigned)(hb[i] + kh) < (unsigned)p.Hin) && ((unsigned)(wb[i] + kw) < (unsigned)p.Win);
          m = (m << 1) | (v ? 1ull : 0ull);
        }
        vmask[i] = m;
      }
      tapoff = (kh_u * p.Wsrc + kw_u) * p.C1 * 2;
    } else {
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) pixel_state(kh_u, kw_u, i, po1[i], po2[i], pv[i]);
    }
  }

  // addresses of the next K step's pieces (VALU only; the DMA instructions are issued later, between MFMA sub-steps)
  unsigned va[A_LOADS], vb[B_LOADS];
  bool use2 = false;
  auto next_addresses = [&]() {
    const int q = kt_issue * CPT + j;
    const bool kvalid = (q < p.nchunks) && (kt_issue < kt_end);
    if (SIMPLE) {
      const unsigned cofs = (unsigned)tapoff + (unsigned)(c8_u + j) * 16u;
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i) va[i] = ((vmask[i] & 1ull) && kvalid) ? rb[i] + cofs : OOB;
#pragma unroll
      for (int i = 0; i < B_LOADS; ++i) vb[i] = (wvalid[i] && kvalid) ? wbase[i] + (unsigned)q * 16u : OOB;
      ++kt_issue;
      c8_u += CPT;
      if (c8_u >= p.cin8) {            // uniform: next tap
        c8_u = 0;
        if (++kw_u == p.KW) {
          kw_u = 0;
          ++kh_u;
        }
        tapoff = (kh_u * p.Wsrc + kw_u) * p.C1 * 2;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) vmask[i] >>= 1;
      }
      return;
    }
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
    use2 = DUAL && (c8_u * 8 >= p.C1);          // uniform: a K step never straddles the concat boundary (C1 % 64 == 0)
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i)
      va[i] = (pv[i] && kvalid) ? (use2 ? po2[i] + (unsigned)(c - p.C1) * 2u : po1[i] + (unsigned)c * 2u) : OOB;
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) vb[i] = (wvalid[i] && kvalid) ? wbase[i] + (unsigned)q * 16u : OOB;
    ++kt_issue;
    if (!KGEN) {
      c8_u += CPT;
      if (c8_u >= p.cin8) {
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
  auto issue_piece = [&](int stage, int q) {   // q in [0, L_TILE): A pieces first, then B pieces
    f16* sa = lds + stage * STAGE + wave * (8 * LDS_ROW);
    f16* sb = sa + BM * LDS_ROW;
    if (q < A_LOADS) {
      if (DUAL && use2) dma16(rx2, sa + q * (64 * LDS_ROW), va[q]);
      else dma16(rx, sa + q * (64 * LDS_ROW), va[q]);
    } else {
      dma16(rw, sb + (q - A_LOADS) * (64 * LDS_ROW), vb[q - A_LOADS]);
    }
  };

  const int frow = lane & 31;
  const int fh = lane >> 5;
  const int swz = (frow >> 1) & 7;

  // ---- main loop: PING-PONG between the two wave groups (waves 0-3 / 4-7; one wave of each per SIMD).
  // Measured on the first version of this kernel (one barrier per K step, all 8 waves in lockstep; rocprofv3 PMC): the MFMA
  // pipe was 36 % busy and a wave spent ~2 170 clocks per K step -- both waves of a SIMD did their address VALU + DMA issue
  // together (pipe idle), then competed for the pipe.  Now a K step is two phases per group, the groups one phase apart:
  //     group 0:  MEM(0)  MFMA(0)  MEM(1)  MFMA(1) ...
  //     group 1:  (wait)  MEM(0)   MFMA(0) MEM(1)  ...         one s_barrier between phases
  // MEM(k)  = addresses + this wave's DMA pieces of K step k+1 into ring stage (k+1) % 3,
  // MFMA(k) = 16 fragment reads + 16 MFMAs on stage k % 3, then s_waitcnt vmcnt(0) (the pieces issued one phase ago).
  // Hazards: stage (k+1)%3 == (k-2)%3 was last read in MFMA(k-2), at least one barrier before any MEM(k); the first reader of
  // stage k+1 (group 0's MFMA(k+1)) starts two barriers after group 1's MEM(k) and one after its vmcnt(0).
  static_assert(NSTAGE >= 3, "ping-pong ring");
  HD_TRACE(2, clock64());
#ifdef HD_CONV_TRACE
  long long tr_mem = 0, tr_b1 = 0, tr_mfma = 0, tr_b2 = 0, tr_t = 0;
#define TR_MARK(accu) do { long long n_ = clock64(); accu += n_ - tr_t; tr_t = n_; } while (0)
#else
#define TR_MARK(accu) do {} while (0)
#endif
  next_addresses();
#pragma unroll
  for (int q = 0; q < L_TILE; ++q) issue_piece(0, q);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();
  int rd = 0, wr = 1;
  HD_TRACE(3, clock64());
#ifdef HD_CONV_TRACE
  tr_t = clock64();
#endif
  for (int kt = 0; kt < nkt; ++kt) {
    if (p.prio == 2) __builtin_amdgcn_s_setprio(1);
    // ---------------- MEM phase
    next_addresses();               // K step kt+1 (zeros past the end: out-of-range offsets)
#pragma unroll
    for (int q = 0; q < L_TILE; ++q) issue_piece(wr, q);
    if (p.prio == 2) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    TR_MARK(tr_mem);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    TR_MARK(tr_b1);
    // ---------------- MFMA phase
    const f16* sa = lds + rd * STAGE;
    const f16* sb = sa + BM * LDS_ROW;
    f16x8 af[2][2], bf[2][2];
    auto read_frags = [&](int s, f16x8 (&A)[2], f16x8 (&B)[2]) {
      const int ks = wk * KSP + s;
      const int slot = ((ks * 2 + fh) ^ swz) * 8;
#pragma unroll
      for (int a = 0; a < 2; ++a) A[a] = *reinterpret_cast<const f16x8*>(sa + (wm * 64 + a * 32 + frow) * LDS_ROW + slot);
#pragma unroll
      for (int b = 0; b < 2; ++b) B[b] = *reinterpret_cast<const f16x8*>(sb + (wn * 64 + b * 32 + frow) * LDS_ROW + slot);
    };
    read_frags(0, af[0], bf[0]);
    if (p.prio == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < KSP; ++s) {
      if (s + 1 < KSP) read_frags(s + 1, af[(s + 1) & 1], bf[(s + 1) & 1]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s & 1][a], bf[s & 1][b], acc[a][b], 0, 0, 0);
    }
    if (p.prio == 1) __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    TR_MARK(tr_mfma);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    TR_MARK(tr_b2);
    rd = (rd + 1 == NSTAGE) ? 0 : rd + 1;
    wr = (wr + 1 == NSTAGE) ? 0 : wr + 1;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  HD_TRACE(4, clock64());
#ifdef HD_CONV_TRACE
  HD_TRACE(8, (unsigned long long)tr_mem);
  HD_TRACE(9, (unsigned long long)tr_b1);
  HD_TRACE(10, (unsigned long long)tr_mfma);
  HD_TRACE(11, (unsigned long long)tr_b2);
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---------------- epilogue ----------------
  // (1) every wave parks its 64 x 64 fp32 partial in LDS: ct[wk][row][col]
  float* ct = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = wn * 64 + b * 32 + (lane & 31);
        ct[(wk * BM + row) * BN + col] = acc[a][b][r];
      }

  constexpr int CPR = BN / 8;            // 8-channel chunks per tile row
  constexpr int RPI = NT_ / CPR;         // rows covered per iteration
  constexpr int ITER = BM / RPI;         // rows per thread
  const int cch = tid % CPR, r0 = tid / CPR;
  const int co = n0 + cch * 8;
  const bool cvalid = co < p.Cout;       // Cout % 8 == 0 => whole chunk valid
  const int Cout = p.Cout;
  const size_t off0 = (size_t)(m0 + r0) * Cout + co;
  bool ok[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) ok[it] = cvalid && (m0 + r0 + it * RPI < p.M);

  bool last = true;
  if (nsl > 1) {
    // (2) split-K: park the raw partial tile in this slice's slab; the last slice to arrive for the tile reduces.
    __syncthreads();
    float* slab = p.ws + (size_t)sl * (size_t)p.M * Cout;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      if (ok[it]) {
        const int row = r0 + it * RPI;
        f32x4 c0 = *reinterpret_cast<const f32x4*>(ct + row * BN + cch * 8);
        f32x4 c1 = *reinterpret_cast<const f32x4*>(ct + row * BN + cch * 8 + 4);
#pragma unroll
        for (int g = 1; g < WK; ++g) {
          c0 += *reinterpret_cast<const f32x4*>(ct + (g * BM + row) * BN + cch * 8);
          c1 += *reinterpret_cast<const f32x4*>(ct + (g * BM + row) * BN + cch * 8 + 4);
        }
        float* d = slab + off0 + (size_t)it * RPI * Cout;
        *reinterpret_cast<f32x4*>(d) = c0;
        *reinterpret_cast<f32x4*>(d + 4) = c1;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(lds);     // the C tile is dead for this slice now
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int t = __hip_atomic_fetch_add(p.tickets + bid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int is_last = (t == nsl - 1);
      if (is_last) {
        __hip_atomic_store(p.tickets + bid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      *flag = is_last;
    }
    __syncthreads();
    last = *flag != 0;
    if (!last) return;
  }

  // residual / mask rows: all requested up front
  const f16* __restrict__ resp = p.res;
  const f16* __restrict__ maskp = p.mask;
  float* __restrict__ statsp = p.stats;
  const int act = p.act;
  f16x8 rv[ITER], mv[ITER];
  if (resp) {
#pragma unroll
    for (int it = 0; it < ITER; ++it)
      if (ok[it]) rv[it] = *reinterpret_cast<const f16x8*>(resp + off0 + (size_t)it * RPI * Cout);
  }
  if (maskp) {
#pragma unroll
    for (int it = 0; it < ITER; ++it)
      if (ok[it]) mv[it] = *reinterpret_cast<const f16x8*>(maskp + off0 + (size_t)it * RPI * Cout);
  }
  float bias8[8];
  {
    f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0;
    if (p.bias && cvalid) {
      q0 = *reinterpret_cast<const f32x4*>(p.bias + co);
      q1 = *reinterpret_cast<const f32x4*>(p.bias + co + 4);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) bias8[k] = k < 4 ? q0[k] : q1[k - 4];
  }
  if (nsl == 1) __syncthreads();

  float ssum8[8], ssq8[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) ssum8[k] = ssq8[k] = 0.f;
  f16* __restrict__ yp = reinterpret_cast<f16*>(p.y);
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    if (ok[it]) {
      const int row = r0 + it * RPI;
      f32x4 c0, c1;
      if (nsl == 1) {
        c0 = *reinterpret_cast<const f32x4*>(ct + row * BN + cch * 8);
        c1 = *reinterpret_cast<const f32x4*>(ct + row * BN + cch * 8 + 4);
#pragma unroll
        for (int g = 1; g < WK; ++g) {
          c0 += *reinterpret_cast<const f32x4*>(ct + (g * BM + row) * BN + cch * 8);
          c1 += *reinterpret_cast<const f32x4*>(ct + (g * BM + row) * BN + cch * 8 + 4);
        }
      } else {
        // slices in fixed order: the sum does not depend on which slice arrived last
        const float* s0 = p.ws + off0 + (size_t)it * RPI * Cout;
        c0 = *reinterpret_cast<const f32x4*>(s0);
        c1 = *reinterpret_cast<const f32x4*>(s0 + 4);
        for (int g = 1; g < nsl; ++g) {
          const float* sg = s0 + (size_t)g * (size_t)p.M * Cout;
          c0 += *reinterpret_cast<const f32x4*>(sg);
          c1 += *reinterpret_cast<const f32x4*>(sg + 4);
        }
      }
      float v[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
      if (resp) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += (float)rv[it][k];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += bias8[k];
      if (maskp) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = ((float)mv[it][k] > 0.f) ? v[k] : 0.f;
      }
      if (statsp) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float vr = (float)(f16)v[k];
          ssum8[k] += vr;
          ssq8[k] += vr * vr;
        }
      }
      if (act == HD_ACT_RELU) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
      } else if (act == HD_ACT_SIGMOID) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 1.f / (1.f + __expf(-v[k]));
      }
      f16x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (f16)v[k];
      *reinterpret_cast<f16x8*>(yp + off0 + (size_t)it * RPI * Cout) = o;
    }
  }
  if (statsp) {
    // lanes that share a channel chunk sit CPR apart: fold the 64/CPR rows of this wave with shuffles, then the 8 waves
    // through LDS in a fixed order (deterministic partial sums)
#pragma unroll
    for (int d = CPR; d < 64; d <<= 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        ssum8[k] += __shfl_xor(ssum8[k], d);
        ssq8[k] += __shfl_xor(ssq8[k], d);
      }
    }
    __syncthreads();                     // everyone is done reading the C tile
    float* red = reinterpret_cast<float*>(lds);   // [8 waves][BN][2]
    if (lane < CPR) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        red[((wave * BN) + lane * 8 + k) * 2 + 0] = ssum8[k];
        red[((wave * BN) + lane * 8 + k) * 2 + 1] = ssq8[k];
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < Cout) {
      float s = 0.f, s2 = 0.f;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        s += red[(m * BN + tid) * 2 + 0];
        s2 += red[(m * BN + tid) * 2 + 1];
      }
      statsp[((size_t)tile_m * 2 + 0) * Cout + n0 + tid] = s;
      statsp[((size_t)tile_m * 2 + 1) * Cout + n0 + tid] = s2;
    }
  }
  HD_TRACE(5, clock64());
  HD_TRACE(6, wall_clock64());
  HD_TRACE(7, hw_ids());
}

template <int WM, int WN, int WK, int NS>
void launch_w8(ConvP& p, int nslices, hipStream_t s) {
  constexpr int BM = WM * 64, BN = WN * 64;
  p.gm = hd_cdiv(p.M, BM);
  p.gn = hd_cdiv(p.Cout, BN);
  dim3 grid(p.gm * p.gn, nslices);
  const bool dual = p.x2 != nullptr;
  const bool kgen = (p.cin8 % 8) != 0;
  const bool simple = !dual && !kgen && !p.up1 && p.in_dil == 1 && p.KH * p.KW <= 64;
  if (dual) hipLaunchKernelGGL((conv_w8_kernel<WM, WN, WK, true, 1, NS>), grid, dim3(512), 0, s, p);
  else if (kgen) hipLaunchKernelGGL((conv_w8_kernel<WM, WN, WK, false, 2, NS>), grid, dim3(512), 0, s, p);
  else if (simple) hipLaunchKernelGGL((conv_w8_kernel<WM, WN, WK, false, 0, NS>), grid, dim3(512), 0, s, p);
  else hipLaunchKernelGGL((conv_w8_kernel<WM, WN, WK, false, 1, NS>), grid, dim3(512), 0, s, p);
}

}  // namespace

// Which problems the family takes at all (the dispatcher decides whether it should)
bool hd_conv_w8_eligible(const ConvP& p) {
  if (p.out_mode != HD_OUT_NHWC_F16 || (p.Cout & 7) != 0) return false;
  if (p.x2 && ((p.C1 % 64) != 0 || (p.C2 % 64) != 0)) return false;
  return true;
}

void hd_conv_w8_tile(int cfg, int* bm, int* bn) {
  static const int t[][2] = {{256, 128}, {128, 128}, {256, 64}, {128, 64}, {128, 256}, {64, 128}, {64, 256}};
  *bm = t[cfg][0];
  *bn = t[cfg][1];
}

// cfg: 0 256x128 | 1 128x128 (WK 2) | 2 256x64 (WK 2) | 3 128x64 (WK 4) | 4 128x256 | 5 64x128 (WK 4) | 6 64x256 (WK 2)
void hd_conv_launch_w8(ConvP& p, int cfg, int nslices, hipStream_t s) {
  p.nk = (p.nchunks + CPT - 1) / CPT;
  if (nslices > p.nk) nslices = p.nk;
  if (nslices < 1) nslices = 1;
  switch (cfg) {
    case 0: launch_w8<4, 2, 1, 3>(p, nslices, s); break;
    case 1: launch_w8<2, 2, 2, 3>(p, nslices, s); break;
    case 2: launch_w8<4, 1, 2, 3>(p, nslices, s); break;
    case 3: launch_w8<2, 1, 4, 3>(p, nslices, s); break;
    case 4: launch_w8<2, 4, 1, 3>(p, nslices, s); break;
    case 5: launch_w8<1, 2, 4, 3>(p, nslices, s); break;
    default: launch_w8<1, 4, 2, 3>(p, nslices, s); break;
  }
}
