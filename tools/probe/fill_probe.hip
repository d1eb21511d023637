// Per-CU global -> LDS fill-rate probe (gfx950): how fast can the workgroups of one CU pull 1-KiB pieces
// (8 rows x 128 B, the conv kernels' DMA piece) into LDS, by form (LDS-DMA / register staging / registers only),
// waves per CU, pieces in flight per wave and source footprint (L2 / Infinity Cache / HBM).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/fill_probe tools/probe/fill_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const char* src, unsigned footprint, char* dst, unsigned voff) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, footprint, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)dst, 16, voff, 0, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#define PROLOG \
  extern __shared__ __attribute__((aligned(1024))) char lds[]; \
  const int lane = threadIdx.x & 63; \
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); \
  const int nw = blockDim.x >> 6; \
  char* ring = lds + wave * DEPTH * 1024; \
  const unsigned cols = row_stride / 128; \
  unsigned rows_total = footprint / row_stride; \
  unsigned row0 = ((blockIdx.x * nw + wave) * 8u) % (rows_total - 8); \
  u32x4 acc = {0, 0, 0, 0}; \
  unsigned col = 0, rowadv = 0; \
  auto addr = [&]() { \
    unsigned rr = row0 + rowadv + (lane >> 3); \
    if (rr >= rows_total) rr -= rows_total; \
    return rr * row_stride + col * 128u + (lane & 7) * 16u; \
  }; \
  auto advance = [&]() { \
    if (++col == cols) { col = 0; rowadv += gridDim.x * nw * 8u; while (rowadv >= rows_total - 8) rowadv -= (rows_total - 8); } \
  };

// a block walks "K" (columns, then the next row group) over its own 8*nw rows like an im2col tile does
template <int DEPTH>
__global__ __launch_bounds__(1024) void fill_dma(const char* src, unsigned footprint, unsigned row_stride, int iters, unsigned* sink) {
  PROLOG
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d) { dma16(src, footprint, ring + d * 1024, addr()); advance(); }
  int slot = DEPTH - 1;
  for (int it = 0; it < iters; ++it) {
    dma16(src, footprint, ring + slot * 1024, addr());
    advance();
    slot = slot + 1 == DEPTH ? 0 : slot + 1;
    wait_vm<DEPTH - 1>();
  }
  wait_vm<0>();
  acc = *reinterpret_cast<u32x4*>(ring + lane * 16);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int DEPTH, int MODE>
__global__ __launch_bounds__(1024) void fill_reg(const char* src, unsigned footprint, unsigned row_stride, int iters, unsigned* sink) {
  PROLOG
  u32x4 regs[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) { regs[d] = *reinterpret_cast<const u32x4*>(src + addr()); advance(); }
  for (int it = 0; it < iters; it += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      u32x4 v = regs[d];
      regs[d] = *reinterpret_cast<const u32x4*>(src + addr());
      advance();
      if (MODE == 1) *reinterpret_cast<u32x4*>(ring + d * 1024 + lane * 16) = v;
      else acc ^= v;
    }
  }
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) acc ^= regs[d];
  if (MODE == 1) acc ^= *reinterpret_cast<u32x4*>(ring + lane * 16);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int DEPTH, int MODE>
static double run(const char* src, unsigned footprint, unsigned row_stride, int waves, int blocks_per_cu, unsigned* sink) {
  const int iters = 2048;
  dim3 grid(256 * blocks_per_cu), block(waves * 64);
  size_t shm = (size_t)waves * DEPTH * 1024;
  typedef void (*kern_t)(const char*, unsigned, unsigned, int, unsigned*);
  kern_t kern = MODE == 0 ? (kern_t)fill_dma<DEPTH> : (MODE == 1 ? (kern_t)fill_reg<DEPTH, 1> : (kern_t)fill_reg<DEPTH, 2>);
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  kern<<<grid, block, shm>>>(src, footprint, row_stride, iters, sink);
  hipEventRecord(a);
  kern<<<grid, block, shm>>>(src, footprint, row_stride, iters, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  if (hipGetLastError() != hipSuccess) return -1;
  float ms; hipEventElapsedTime(&ms, a, b);
  double bytes = (double)grid.x * waves * iters * 1024.0;
  return bytes / (ms * 1e-3) / 1e9 / 256.0;   // GB/s per CU
}

int main() {
  const size_t big = 2040u << 20;
  char* src; unsigned* sink;
  hipMalloc(&src, big); hipMemset(src, 1, big); hipMalloc(&sink, 4);
  const unsigned fps[3] = {2u << 20, 96u << 20, 2040u << 20};
  const char* fpn[3] = {"L2 2MB", "MALL 96MB", "HBM 2GB"};
  const char* mn[3] = {"lds-dma", "reg+ds_write", "reg only"};
  printf("GB/s per CU (x256 = chip); row_stride 512 B; pieces = 8 rows x 128 B\n");
  for (int f = 0; f < 3; ++f)
    for (int mode = 0; mode < 3; ++mode) {
      printf("%-10s %-13s", fpn[f], mn[mode]);
      for (int waves : {4, 8, 16})
        for (int bpc : {1, 2}) {
          double v2, v4, v8;
          if (mode == 0) { v2 = run<2, 0>(src, fps[f], 512, waves, bpc, sink); v4 = run<4, 0>(src, fps[f], 512, waves, bpc, sink); v8 = (waves * 8 * bpc <= 160) ? run<8, 0>(src, fps[f], 512, waves, bpc, sink) : -1; }
          else if (mode == 1) { v2 = run<2, 1>(src, fps[f], 512, waves, bpc, sink); v4 = run<4, 1>(src, fps[f], 512, waves, bpc, sink); v8 = (waves * 8 * bpc <= 160) ? run<8, 1>(src, fps[f], 512, waves, bpc, sink) : -1; }
          else { v2 = run<2, 2>(src, fps[f], 512, waves, bpc, sink); v4 = run<4, 2>(src, fps[f], 512, waves, bpc, sink); v8 = run<8, 2>(src, fps[f], 512, waves, bpc, sink); }
          printf(" | w%d b%d: %5.1f %5.1f %5.1f", waves, bpc, v2, v4, v8);
        }
      printf("\n");
    }
  printf("(columns per cell: 2 / 4 / 8 pieces in flight per wave)\n");
  return 0;
}
