// 8-wave patch-staged weight gradient: kernel wrapper and launch (the kernel body lives in wgrad3x3_w8_body.h)
#include "wgrad3x3_w8_body.h"

namespace {

__global__ __launch_bounds__(512, 2) void wgrad3x3_w8_kernel(hd_wg8::Wg8P p) {
  __shared__ __attribute__((aligned(1024))) f16 lds[hd_wg8::LDS_HALVES];
  hd_wg8::wgrad3x3_w8_body(p, lds, blockIdx.x, blockIdx.y);
}

// Several layers' weight gradients as ONE grid (hd_wgrad_multi): block b of the 1-D grid belongs to the entry whose block range holds it
// and runs exactly the body a separate launch of that entry would run for (bx, by) -- bit-identical results.  Why: a block's fixed cost
// (set-up, the two pixel halves meeting in LDS, a 147 KB fp32 partial written to the slab and read back by the reduction) is ~40 % of a
// 5-tile block, and 256 blocks per LAYER is what filling the chip with one layer costs; a ResNet stage's layers together fill it with one
// block per (co, ci) tile and layer, each walking ALL of the layer's pixel tiles: 1 / nsplit of the slab bytes and of the fixed cost.
struct Wg8Tab {
  hd_wg8::Wg8P p[HD_WGRAD_MULTI_MAX];
  int first_block[HD_WGRAD_MULTI_MAX];
  int gx[HD_WGRAD_MULTI_MAX];
};

__global__ __launch_bounds__(512, 2) void wgrad3x3_w8_multi_kernel(const Wg8Tab tab, int n) {
  __shared__ __attribute__((aligned(1024))) f16 lds[hd_wg8::LDS_HALVES];
  int e = 0;
  for (int i = 1; i < n; ++i)
    if ((int)blockIdx.x >= tab.first_block[i]) e = i;
  const int b = (int)blockIdx.x - tab.first_block[e];
  const int gx = tab.gx[e];
  hd_wg8::wgrad3x3_w8_body(tab.p[e], lds, b % gx, b / gx);
}

}  // namespace

void hd_wgrad_w8_launch_multi(const hd_wgrad_args* a, int n, hipStream_t s) {
  Wg8Tab tab;
  int first = 0;
  for (int i = 0; i < n; ++i) {
    int gx, gy;
    hd_wg8::fill_params(a + i, tab.p[i], &gx, &gy);
    tab.first_block[i] = first;
    tab.gx[i] = gx;
    first += gx * gy;
  }
  for (int i = n; i < HD_WGRAD_MULTI_MAX; ++i) {
    tab.p[i] = tab.p[n - 1];
    tab.first_block[i] = 0x7fffffff;
    tab.gx[i] = 1;
  }
  hipLaunchKernelGGL(wgrad3x3_w8_multi_kernel, dim3(first), dim3(512), 0, s, tab, n);
}

bool hd_wgrad_w8_eligible(const hd_wgrad_args* a) {
  if (a->KH != 3 || a->KW != 3 || a->stride != 1 || a->pad != 1 || a->Ho != a->Hin || a->Wo != a->Win) return false;
  if ((a->Cout % 64) != 0 || a->nsplit < 1) return false;
  const int64_t xb = (int64_t)a->N * a->Hsrc * a->Wsrc * a->C1 * 2, x2b = a->x2 ? (int64_t)a->N * a->Hin * a->Win * a->C2 * 2 : 0;
  const int64_t db = (int64_t)a->N * a->Ho * a->Wo * a->Cout * 2;
  if (xb >= 0x80000000ll || x2b >= 0x80000000ll || db >= 0x80000000ll) return false;
  if (a->x2) return a->up1 && (a->C1 % 64) == 0 && (a->C2 % 64) == 0;
  return !a->up1 && (a->C1 % 64) == 0;
}

void hd_wgrad_w8_launch(const hd_wgrad_args* a, hipStream_t s) {
  hd_wg8::Wg8P p;
  int gx, gy;
  hd_wg8::fill_params(a, p, &gx, &gy);
  hipLaunchKernelGGL(wgrad3x3_w8_kernel, dim3(gx, gy), dim3(512), 0, s, p);
}
