// Shared device/host helpers for libhallucidet_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/hallucidet_hip.h"

// Storage type of activations.  The bandwidth-bound sources that touch activations without matrix cores (elementwise.hip,
// roi_align.hip, the GroupNorm kernels of fcos.hip) are built TWICE: as they stand (`f16` = IEEE half: the product's storage type),
// and with -DHD_STORE_F32, where the same code stores activations as fp32 and every entry point carries the suffix _f32 -- the
// element-wise half of `--precision 32` (the reference's default, src/config/config.py:149; convolutions: conv_f32.hip).  In that build
// the names f16 / f16x8 mean "the storage type" (a vector of eight is then two 16-byte accesses), nothing else changes.
#ifdef HD_STORE_F32
typedef float f16;
typedef float f16x2 __attribute__((ext_vector_type(2)));
typedef float f16x4 __attribute__((ext_vector_type(4)));
typedef float f16x8 __attribute__((ext_vector_type(8)));
#define HD_API(name) name##_f32
#else
typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define HD_API(name) name
#endif
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

void hd_set_error(const char* fmt, ...);

#define HD_CHECK_ARG(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      hd_set_error(__VA_ARGS__);       \
      return HD_E_ARG;                 \
    }                                  \
  } while (0)

#define HD_CHECK_LAUNCH()                                                   \
  do {                                                                      \
    hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) {                                                 \
      hd_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,           \
                   hipGetErrorString(e_));                                  \
      return HD_E_LAUNCH;                                                   \
    }                                                                       \
  } while (0)

static inline int hd_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// BatchNorm's affine map as ONE fused multiply-add, spelled out so that every kernel that evaluates it (hd_bn_apply, the ReLU-mask
// recomputation of the BatchNorm backward, the consumer-side BatchNorm of the small-channel kernels) rounds identically.
__device__ __forceinline__ float hd_bn_affine(float y, float scale, float shift) { return __builtin_fmaf(y, scale, shift); }

