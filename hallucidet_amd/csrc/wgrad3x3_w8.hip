// 8-wave patch-staged weight gradient: kernel wrapper and launch (the kernel body lives in wgrad3x3_w8_body.h)
#include "wgrad3x3_w8_body.h"

namespace {

__global__ __launch_bounds__(512, 2) void wgrad3x3_w8_kernel(hd_wg8::Wg8P p) {
  __shared__ __attribute__((aligned(1024))) f16 lds[hd_wg8::LDS_HALVES];
  hd_wg8::wgrad3x3_w8_body(p, lds, blockIdx.x, blockIdx.y);
}

}  // namespace

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
