// What one in-kernel grid barrier costs on gfx950 (all blocks co-resident; release / acquire at agent scope across the eight XCDs'
// L2s): the number a cooperative "finalize + apply" BatchNorm kernel or an in-launch split-K has to beat is one launch floor (4-5 us).
//   hipcc --offload-arch=gfx950 -O2 tools/probe_grid_barrier.hip -o /tmp/probe_grid_barrier && /tmp/probe_grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void barrier_loop(unsigned* counter, float* data, int iters, float* out, int dirty_floats) {
  unsigned target = 0;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    // every block dirties `dirty_floats` floats of its own region (what a real kernel's phase 1 would have written)
    for (int i = threadIdx.x; i < dirty_floats; i += blockDim.x) data[(size_t)blockIdx.x * dirty_floats + i] = (float)(it + i);
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();                                   // release: this block's writes visible to the agent
      atomicAdd(counter, 1u);
      target += gridDim.x;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      __threadfence();                                   // acquire
    }
    __syncthreads();
    const int nb = (blockIdx.x + 37) % gridDim.x;        // read what another block (another XCD) wrote
    acc += __hip_atomic_load(&data[(size_t)nb * dirty_floats + (threadIdx.x % dirty_floats)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (acc == -1.f) out[0] = acc;
}

__global__ void empty_kernel(float* out) {
  if (out == nullptr) return;
}

int main() {
  unsigned* counter;
  float *data, *out;
  hipMalloc(&counter, 4);
  hipMalloc(&data, (size_t)1024 * 65536 * 4);
  hipMalloc(&out, 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int iters = 200;
  for (int blocks : {64, 256, 512, 1024}) {
    for (int dirty : {64, 4096, 65536}) {
      hipMemset(counter, 0, 4);
      hipLaunchKernelGGL(barrier_loop, dim3(blocks), dim3(256), 0, 0, counter, data, 3, out, dirty);   // warm-up
      hipDeviceSynchronize();
      hipMemset(counter, 0, 4);
      hipEventRecord(a, 0);
      hipLaunchKernelGGL(barrier_loop, dim3(blocks), dim3(256), 0, 0, counter, data, iters, out, dirty);
      hipEventRecord(b, 0);
      hipEventSynchronize(b);
      float ms = 0.f;
      hipEventElapsedTime(&ms, a, b);
      printf("blocks %4d x 256 threads, %6d floats dirtied per block and episode: %.2f us per write + barrier + read episode\n", blocks, dirty,
             ms * 1e3f / iters);
    }
  }
  // the launch floor for comparison: dependent empty kernels on one stream
  hipEventRecord(a, 0);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, out);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  printf("200 dependent empty launches (eager, one stream): %.2f us each\n", ms * 1e3f / 200);
  return 0;
}
