#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const unsigned* src, int nbytes, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4 * 2];
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 0xdeadbeef;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, nbytes, 0x00020000);
  // lane L fetches source chunk (63-L) (per-lane gather), odd lanes out of range
  unsigned voff = (threadIdx.x & 1) ? 0xFFFFFFF0u : (63 - threadIdx.x) * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds, 16, voff, 0, 0, 0);
  // second instruction to the second KiB with an offset
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(lds + 256), 16, threadIdx.x * 16, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
  unsigned h[256]; for (int i = 0; i < 256; i++) h[i] = i;
  unsigned *d, *o; hipMalloc(&d, 1024); hipMalloc(&o, 2048);
  hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, 1024, o);
  unsigned r[512]; hipMemcpy(r, o, 2048, hipMemcpyDeviceToHost);
  printf("first KiB (lane L -> chunk 63-L, odd lanes OOB):\n");
  for (int L = 0; L < 8; L++) printf(" lane%d: %x %x %x %x\n", L, r[L*4], r[L*4+1], r[L*4+2], r[L*4+3]);
  printf("second KiB linear:\n");
  for (int L = 0; L < 4; L++) printf(" lane%d: %x %x %x %x\n", L, r[256+L*4], r[256+L*4+1], r[256+L*4+2], r[256+L*4+3]);
  printf("last lane63: %x (expect chunk0 -> 0)\n", r[63*4]);
  return 0;
}
