#include "hd_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void hd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int hd_abi_version(void) { return 6; }      // 6: the struct growth that round 5 shipped under version 5 -- hd_conv_args.y2, hd_wgrad_args.dw_oihw / dw_scale / reserved1, + hd_wgrad_multi, hd_wgrad_direct_ok -- gets its own number, and the Python binding now refuses any other (hallucidet_amd/_abi.py: ABI_VERSION); 5: + hd_gemm_w8_mode (large-tile GEMM path behind hd_conv2d), hd_pad_cast_f32_f16_multi, hd_conv_args.out_pool2 (was reserved0) + hd_conv2d_pool2_ok; 4: hd_conv_args carries bs_* (BatchNorm backward sums from the data-gradient epilogue), + hd_conv2d_bstat_ok, hd_maxpool3x3s2_bwd_idx_add, hd_concat_up_bwd, the _f32 twins; 2: hd_conv_args / hd_wgrad_args carry in_scale / in_shift / in_relu; 3: + hd_conv2d_multi, hd_conv2d_wgrad, hd_wgrad_reduce_multi; - hd_conv2d_patch, hd_conv_set_workspace
extern "C" const char* hd_last_error(void) { return g_err; }
extern "C" const char* hd_arch(void) { return "gfx950"; }
