// Does "the last block to finish runs the finalize" beat a dependent finalize launch on gfx950 WITHOUT an agent-scope release fence?
// Round 3 measured 3-8 us for release + acquire per ticket episode (buffer_wbl2 writes back every dirty line of an XCD's L2).  Here the
// few KB the finalize needs (a block's row of partial sums) leave through WRITE-THROUGH stores (agent-scope relaxed atomic stores: sc1),
// the ticket is a relaxed agent-scope atomic add after s_waitcnt vmcnt(0) of the storing wave, and the last block reads all rows with
// agent-scope relaxed atomic loads (sc1: not served from a stale local L2 line).  No buffer_wbl2, no buffer_inv.
// Producer = stand-in for a convolution tile: `work` rounds of FMA on registers, a 64 KB output tile (plain stores: dirty L2 lines),
// one row of 2*C partial sums.  Timed inside one hipGraph of 20 (producer [+ finalize]) pairs, 160 blocks x 512 threads.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_last_block.hip -o /tmp/probe_last_block && /tmp/probe_last_block
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

__device__ __forceinline__ float busy(float x, int rounds) {
  float a = x, b = 1.0001f;
  for (int i = 0; i < rounds; ++i) {
    a = a * b + 0.5f;
    b = b * 0.9999f + 0.0001f;
  }
  return a + b;
}

template <bool TICKET>
__global__ __launch_bounds__(512) void producer(float* __restrict__ y, float* part, int C, int work, unsigned* counter, float* __restrict__ fin) {
  const int tid = threadIdx.x, bid = blockIdx.x, rows = gridDim.x;
  const float v = busy((float)(tid + bid), work);
  // the tile's output: 64 KB per block, ordinary stores
  for (int i = tid; i < 16384; i += 512) y[(size_t)bid * 16384 + i] = v + (float)i;
  // the row of partial sums: 2*C floats
  if (TICKET) {
    for (int c = tid; c < 2 * C; c += 512)
      __hip_atomic_store(&part[(size_t)bid * 2 * C + c], v * 1e-3f + (float)c + (float)bid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's stores (write-through rows included) are acknowledged
    __shared__ unsigned s_last;
    __syncthreads();
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (t == (unsigned)rows - 1u) ? 1u : 0u;
      if (s_last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // self-cleaning for the next launch
    }
    __syncthreads();
    if (!s_last) return;
    // finalize by the last block: fixed row order per column (the order of the separate kernel below)
    for (int c = tid; c < 2 * C; c += 512) {
      double s = 0.0;
      int r = 0;
      for (; r + 3 < rows; r += 4) {
        const float a0 = __hip_atomic_load(&part[(size_t)r * 2 * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float a1 = __hip_atomic_load(&part[(size_t)(r + 1) * 2 * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float a2 = __hip_atomic_load(&part[(size_t)(r + 2) * 2 * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float a3 = __hip_atomic_load(&part[(size_t)(r + 3) * 2 * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s += (double)a0;
        s += (double)a1;
        s += (double)a2;
        s += (double)a3;
      }
      for (; r < rows; ++r) s += (double)__hip_atomic_load(&part[(size_t)r * 2 * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      fin[c] = (float)s;
    }
  } else {
    for (int c = tid; c < 2 * C; c += 512) part[(size_t)bid * 2 * C + c] = v * 1e-3f + (float)c + (float)bid;
  }
}

__global__ __launch_bounds__(256) void finalize(const float* __restrict__ part, int rows, int C, float* __restrict__ fin) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= 2 * C) return;
  double s = 0.0;
  int r = 0;
  for (; r + 3 < rows; r += 4) {
    const float a0 = part[(size_t)r * 2 * C + c], a1 = part[(size_t)(r + 1) * 2 * C + c], a2 = part[(size_t)(r + 2) * 2 * C + c],
                a3 = part[(size_t)(r + 3) * 2 * C + c];
    s += (double)a0;
    s += (double)a1;
    s += (double)a2;
    s += (double)a3;
  }
  for (; r < rows; ++r) s += (double)part[(size_t)r * 2 * C + c];
  fin[c] = (float)s;
}

__global__ void consumer(const float* __restrict__ fin, int n, float* __restrict__ out) {      // the dependent next kernel reads the result
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fin[i] * 2.f;
}

int main() {
  const int blocks = 160, pairs = 20;
  float *y, *part, *fin_a, *fin_b, *out;
  unsigned* counter;
  hipMalloc(&y, (size_t)blocks * 16384 * 4 * pairs);
  hipMalloc(&part, (size_t)blocks * 2 * 512 * 4);
  hipMalloc(&fin_a, 1024 * 4 * pairs);
  hipMalloc(&fin_b, 1024 * 4 * pairs);
  hipMalloc(&out, 1024 * 4);
  hipMalloc(&counter, 4);
  hipMemset(counter, 0, 4);
  hipStream_t s;
  hipStreamCreate(&s);
  for (int C : {64, 256, 512}) {
    for (int work : {2000, 8000}) {
      float us[2];
      for (int mode = 0; mode < 2; ++mode) {
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < pairs; ++i) {
          float* yy = y + (size_t)i * blocks * 16384;
          if (mode == 0) {
            hipLaunchKernelGGL(producer<false>, dim3(blocks), dim3(512), 0, s, yy, part, C, work, counter, fin_a + i * 1024);
            hipLaunchKernelGGL(finalize, dim3((2 * C + 255) / 256), dim3(256), 0, s, part, blocks, C, fin_a + i * 1024);
            hipLaunchKernelGGL(consumer, dim3(4), dim3(256), 0, s, fin_a + i * 1024, 2 * C, out);
          } else {
            hipLaunchKernelGGL(producer<true>, dim3(blocks), dim3(512), 0, s, yy, part, C, work, counter, fin_b + i * 1024);
            hipLaunchKernelGGL(consumer, dim3(4), dim3(256), 0, s, fin_b + i * 1024, 2 * C, out);
          }
        }
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
          hipEventRecord(a, s);
          hipGraphLaunch(ge, s);
          hipEventRecord(b, s);
          hipEventSynchronize(b);
          float ms = 0.f;
          hipEventElapsedTime(&ms, a, b);
          if (ms < best) best = ms;
        }
        us[mode] = best * 1e3f / pairs;
        hipGraphExecDestroy(ge);
        hipGraphDestroy(g);
      }
      // equality of the two results (same fixed order) over all pairs, and a stress loop for stale rows
      std::vector<float> ha(1024 * pairs), hb(1024 * pairs);
      hipMemcpy(ha.data(), fin_a, ha.size() * 4, hipMemcpyDeviceToHost);
      hipMemcpy(hb.data(), fin_b, hb.size() * 4, hipMemcpyDeviceToHost);
      int bad = 0;
      for (int i = 0; i < pairs; ++i)
        for (int c = 0; c < 2 * C; ++c) bad += ha[i * 1024 + c] != hb[i * 1024 + c];
      printf("C %3d, producer work %5d: producer + finalize launch + consumer %.2f us | producer with last-block finalize + consumer %.2f us | %d of %d results differ\n",
             C, work, us[0], us[1], bad, pairs * 2 * C);
    }
  }
  // stress: 2 000 launches with changing contents, every result checked against the host sum of the rows read back afterwards
  {
    const int C = 256;
    std::vector<float> hp((size_t)blocks * 2 * C), hf(2 * C);
    int bad = 0;
    for (int it = 0; it < 2000; ++it) {
      hipLaunchKernelGGL(producer<true>, dim3(blocks), dim3(512), 0, s, y, part, C, 500 + (it % 7) * 300, counter, fin_b);
      hipStreamSynchronize(s);
      hipMemcpy(hp.data(), part, hp.size() * 4, hipMemcpyDeviceToHost);
      hipMemcpy(hf.data(), fin_b, hf.size() * 4, hipMemcpyDeviceToHost);
      for (int c = 0; c < 2 * C; ++c) {
        double sum = 0.0;
        for (int r = 0; r < blocks; ++r) sum += (double)hp[(size_t)r * 2 * C + c];
        bad += (float)sum != hf[c];
      }
    }
    printf("stress: 2000 launches x %d columns, %d results differ from the host sum of the rows\n", 2 * C, bad);
  }
  return 0;
}
