// Fused optimizer step over a flat fp32 parameter arena: unscale -> clip_grad_value_ -> Adam.
// Reference semantics: torch.optim.Adam (config.py:204-245, train_hallucidet.py:431-435) with
// Lightning's gradient_clip_val=0.5 / algorithm "value" (train_hallucidet.py:498-499) applied first.
#include "hd_common.h"

namespace {

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            int64_t n, float lr, float b1, float b2, float eps, float wd, float clip, float inv_scale, float bc1,
                            float bc2_sqrt, const float* __restrict__ found_inf) {
  if (found_inf && found_inf[0] != 0.f) return;
  const float step = lr / bc1;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
    if (i + 3 < n) {
      f32x4 pp = *reinterpret_cast<f32x4*>(p + i), gg = *reinterpret_cast<const f32x4*>(g + i);
      f32x4 mm = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float gr = gg[k] * inv_scale;
        if (clip > 0.f) gr = fminf(fmaxf(gr, -clip), clip);
        if (wd != 0.f) gr += wd * pp[k];
        mm[k] = b1 * mm[k] + (1.f - b1) * gr;
        vv[k] = b2 * vv[k] + (1.f - b2) * gr * gr;
        float denom = sqrtf(vv[k]) / bc2_sqrt + eps;
        pp[k] -= step * (mm[k] / denom);
      }
      *reinterpret_cast<f32x4*>(p + i) = pp;
      *reinterpret_cast<f32x4*>(m + i) = mm;
      *reinterpret_cast<f32x4*>(v + i) = vv;
    } else {
      for (int64_t j = i; j < n; ++j) {
        float gr = g[j] * inv_scale;
        if (clip > 0.f) gr = fminf(fmaxf(gr, -clip), clip);
        if (wd != 0.f) gr += wd * p[j];
        m[j] = b1 * m[j] + (1.f - b1) * gr;
        v[j] = b2 * v[j] + (1.f - b2) * gr * gr;
        float denom = sqrtf(v[j]) / bc2_sqrt + eps;
        p[j] -= step * (m[j] / denom);
      }
    }
  }
}

__global__ void check_finite_kernel(const float* __restrict__ g, int64_t n, float* found) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float x = g[i];
    if (!(fabsf(x) <= 3.0e38f)) bad = true;
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) found[0] = 1.f;
}

}  // namespace

extern "C" int hd_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                            float weight_decay, float clip_value, float inv_scale, float bias_corr1, float bias_corr2,
                            const float* found_inf, void* stream) {
  HD_CHECK_ARG(p && g && m && v && n > 0 && bias_corr1 > 0.f && bias_corr2 > 0.f, "hd_adam_step: bad args");
  HD_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "hd_adam_step: buffers must be 16-byte aligned");
  int64_t nv = (n + 3) / 4;
  int grid = (int)((nv + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                     clip_value, inv_scale, bias_corr1, sqrtf(bias_corr2), found_inf);
  HD_CHECK_LAUNCH();
  return HD_OK;
}

extern "C" int hd_check_finite(const float* g, int64_t n, float* found_inf, void* stream) {
  HD_CHECK_ARG(g && found_inf && n > 0, "hd_check_finite: bad args");
  int grid = (int)((n + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(check_finite_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, n, found_inf);
  HD_CHECK_LAUNCH();
  return HD_OK;
}
