"""ctypes binding of libhallucidet_hip.so (the C ABI in include/hallucidet_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails the
product path raises.  torch is used here only to obtain device pointers and the
current HIP stream.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HD_HIP_LIB: load another build of the same ABI (A/B runs of kernel changes, the profiling build); default = the in-tree library
LIB_PATH = os.environ.get("HD_HIP_LIB") or os.path.join(_HERE, "lib", "libhallucidet_hip.so")

# hd_abi_version() this binding's struct mirrors and prototypes belong to (include/hallucidet_hip.h); load() refuses any other build
ABI_VERSION = 6

HD_ACT_NONE, HD_ACT_RELU, HD_ACT_SIGMOID = 0, 1, 2
HD_OUT_NHWC_F16, HD_OUT_NCHW_F32, HD_OUT_NHWC_F32 = 0, 1, 2

c_i32 = C.c_int32
c_i64 = C.c_int64
c_f = C.c_float
c_d = C.c_double
vp = C.c_void_p


class HipLibraryMissing(RuntimeError):
    pass


class HipCallError(RuntimeError):
    pass


class ConvArgs(C.Structure):
    _fields_ = [
        ("x", vp), ("x2", vp), ("w", vp), ("bias", vp), ("res", vp), ("mask", vp), ("y", vp), ("stats", vp),
        ("N", c_i32), ("Hsrc", c_i32), ("Wsrc", c_i32), ("Hin", c_i32), ("Win", c_i32),
        ("C1", c_i32), ("C2", c_i32), ("Ho", c_i32), ("Wo", c_i32), ("Cout", c_i32),
        ("KH", c_i32), ("KW", c_i32), ("stride", c_i32), ("pad", c_i32),
        ("up1", c_i32), ("in_dil", c_i32), ("act", c_i32), ("out_mode", c_i32),
        ("in_scale", vp), ("in_shift", vp), ("in_relu", c_i32), ("out_pool2", c_i32),
        ("bs_y", vp), ("bs_z", vp), ("bs_mean", vp), ("bs_invstd", vp), ("bs_gamma", vp), ("bs_beta", vp), ("bs_relu", c_i32), ("reserved1", c_i32),
        ("y2", vp),
    ]


class WprepDesc(C.Structure):
    _fields_ = [("w_oihw", C.c_void_p), ("w_fwd", C.c_void_p), ("w_dgrad", C.c_void_p)] + \
               [(n, C.c_int32) for n in ("Cout", "Cin", "KH", "KW", "Cin_pad", "Cout_pad")]


class WredDesc(C.Structure):
    _fields_ = [("slab", C.c_void_p), ("dw_oihw", C.c_void_p)] + \
               [(n, C.c_int32) for n in ("nsplit", "Cout_slab", "Cout", "KH", "KW", "Cin", "Cin_real", "accumulate")] + \
               [("scale", C.c_float)] + [(n, C.c_int32) for n in ("mode", "first_block", "reserved")]


class WgradArgs(C.Structure):
    _fields_ = [
        ("x", vp), ("x2", vp), ("dy", vp), ("slab", vp),
        ("N", c_i32), ("Hsrc", c_i32), ("Wsrc", c_i32), ("Hin", c_i32), ("Win", c_i32),
        ("C1", c_i32), ("C2", c_i32), ("Ho", c_i32), ("Wo", c_i32), ("Cout", c_i32),
        ("KH", c_i32), ("KW", c_i32), ("stride", c_i32), ("pad", c_i32), ("up1", c_i32),
        ("nsplit", c_i32),
        ("in_scale", vp), ("in_shift", vp), ("in_relu", c_i32), ("reserved0", c_i32),
        ("dw_oihw", vp), ("dw_scale", c_f), ("reserved1", c_i32),
    ]


# name -> (restype, argtypes).  Must list every symbol include/hallucidet_hip.h declares
# (tests/test_abi.py parses the header and checks this table and the .so against it).
PROTOTYPES = {
    "hd_abi_version": (C.c_int, []),
    "hd_last_error": (C.c_char_p, []),
    "hd_arch": (C.c_char_p, []),
    "hd_conv2d": (C.c_int, [C.POINTER(ConvArgs), vp]),
    "hd_conv2d_stats_rows": (C.c_int, [C.POINTER(ConvArgs)]),
    "hd_conv2d_pool2_ok": (C.c_int, [C.POINTER(ConvArgs)]),
    "hd_conv2d_bstat_ok": (C.c_int, [C.POINTER(ConvArgs)]),
    "hd_conv7x7s2_dgrad_thin": (C.c_int, [vp, vp, vp, vp] + [C.c_int] * 5 + [vp]),
    "hd_conv_tune_override": (C.c_int, [C.c_int] * 4),
    "hd_conv_tune_w8": (C.c_int, [C.c_int, C.c_int]),
    "hd_conv_nominal_batch": (C.c_int, [C.c_int]),
    "hd_gemm_w8_mode": (C.c_int, [C.c_int]),
    "hd_wgrad_w8_blocks": (C.c_int, [C.POINTER(WgradArgs)]),
    "hd_wgrad": (C.c_int, [C.POINTER(WgradArgs), vp]),
    "hd_wgrad_multi": (C.c_int, [C.POINTER(WgradArgs), C.c_int, vp]),
    "hd_wgrad_direct_ok": (C.c_int, [C.POINTER(WgradArgs)]),
    "hd_conv2d_wgrad": (C.c_int, [C.POINTER(ConvArgs), C.POINTER(WgradArgs), vp]),
    "hd_conv2d_multi": (C.c_int, [C.POINTER(ConvArgs), C.c_int, vp]),
    "hd_wgrad_tune_override": (C.c_int, [C.c_int]),
    "hd_wgrad_reduce": (C.c_int, [vp, vp] + [C.c_int] * 7 + [c_f, C.c_int, vp]),
    "hd_wgrad_reduce_plan": (C.c_int, [vp, C.c_int]),
    "hd_wgrad_reduce_multi": (C.c_int, [vp, C.c_int, C.c_int, vp]),
    "hd_weight_prep": (C.c_int, [vp, vp, vp, vp] + [C.c_int] * 6 + [vp]),
    "hd_weight_prep_multi": (C.c_int, [vp, C.c_int, C.c_int, vp]),
    "hd_colsum": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp]),
    "hd_rowsum": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, vp]),
    "hd_bn_finalize": (C.c_int, [vp, C.c_int, C.c_int, c_d, vp, vp, vp, vp, c_f, c_f, vp, vp, vp, vp, vp]),
    "hd_bn_eval_scale_shift": (C.c_int, [vp, vp, vp, vp, c_f, C.c_int, vp, vp, vp]),
    "hd_bn_apply": (C.c_int, [vp, vp, vp, vp, vp, c_i64, C.c_int, C.c_int, vp]),
    "hd_bn_bwd_reduce": (C.c_int, [vp] * 8 + [C.c_int, c_i64, C.c_int, C.c_int, vp]),
    "hd_bn_bwd_apply": (C.c_int, [vp] * 8 + [C.c_int] + [vp] * 5 + [c_f, C.c_int, c_i64, C.c_int, C.c_int, vp]),
    "hd_maxpool3x3s2": (C.c_int, [vp, vp] + [C.c_int] * 6 + [vp]),
    "hd_maxpool3x3s2_bwd": (C.c_int, [vp, vp, vp] + [C.c_int] * 6 + [vp]),
    "hd_maxpool3x3s2_idx": (C.c_int, [vp, vp, vp] + [C.c_int] * 6 + [vp]),
    "hd_maxpool3x3s2_bwd_idx": (C.c_int, [vp, vp, vp] + [C.c_int] * 6 + [vp]),
    "hd_maxpool3x3s2_bwd_idx_add": (C.c_int, [vp, vp, vp, vp] + [C.c_int] * 6 + [vp]),
    "hd_concat_up_bwd": (C.c_int, [vp, vp, vp] + [C.c_int] * 5 + [vp]),
    "hd_subsample2": (C.c_int, [vp, vp] + [C.c_int] * 6 + [vp]),
    "hd_subsample2_bwd": (C.c_int, [vp, vp] + [C.c_int] * 7 + [vp]),
    "hd_nchw_to_nhwc_resize": (C.c_int, [vp, vp] + [C.c_int] * 7 + [vp]),
    "hd_nchw_to_nhwc_resize_strided": (C.c_int, [vp, c_i64, c_i64, vp] + [C.c_int] * 7 + [vp]),
    "hd_nchw_to_nhwc_resize_bwd": (C.c_int, [vp, vp] + [C.c_int] * 7 + [c_f, vp]),
    "hd_nhwc_to_nchw": (C.c_int, [vp, vp] + [C.c_int] * 5 + [vp]),
    "hd_upsample_add": (C.c_int, [vp, vp, vp] + [C.c_int] * 6 + [vp]),
    "hd_upsample_add_bwd": (C.c_int, [vp, vp] + [C.c_int] * 7 + [vp]),
    "hd_upsample2_bwd": (C.c_int, [vp, vp] + [C.c_int] * 7 + [vp]),
    "hd_add_f16": (C.c_int, [vp, vp, vp, c_i64, vp]),
    "hd_slice_channels": (C.c_int, [vp, vp, c_i64, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "hd_sigmoid_bwd_nchw_to_nhwc": (C.c_int, [vp, vp, vp] + [C.c_int] * 5 + [c_f, vp]),
    "hd_relu_bwd": (C.c_int, [vp, vp, vp, c_i64, vp]),
    "hd_f32_to_f16": (C.c_int, [vp, vp, c_i64, c_f, vp]),
    "hd_f16_to_f32": (C.c_int, [vp, vp, c_i64, c_f, vp]),
    "hd_pad_cast_f32_f16": (C.c_int, [vp, vp, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int64, vp]),
    "hd_pad_cast_f32_f16_multi": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, vp]),
    "hd_channel_sum_f16": (C.c_int, [vp, c_i64, C.c_int, vp, C.c_int, vp]),
    "hd_scale_store": (C.c_int, [vp, vp, C.c_int, c_f, C.c_int, vp]),
    "hd_nms_sorted_batched": (C.c_int, [vp, vp, C.c_int, C.c_int, c_f, vp, vp, vp]),
    "hd_nms_sorted_batched_topk": (C.c_int, [vp, vp, C.c_int, C.c_int, c_f, vp, vp, C.c_int, vp]),
    "hd_roi_align": (C.c_int, [vp, vp, vp] + [C.c_int] * 7 + [c_f, C.c_int, vp]),
    "hd_roi_align_ml": (C.c_int, [vp, vp, vp, vp, C.c_int, vp, vp, vp] + [C.c_int] * 5 + [vp]),
    "hd_roi_align_ml_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp] + [C.c_int] * 6 + [vp]),
    "hd_roi_align_ml_bwd_gather": (C.c_int, [vp, vp, vp, vp, vp, vp, vp] + [C.c_int] * 7 + [vp]),
    "hd_roi_align_bwd": (C.c_int, [vp, vp, vp] + [C.c_int] * 7 + [c_f, C.c_int, vp]),
    "hd_box_iou": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp]),
    "hd_box_iou_batched": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "hd_rpn_loss": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int64, C.c_float, vp, C.c_float, vp, vp, vp]),
    "hd_rpn_loss_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int64, C.c_float, vp, vp, vp, C.c_float, vp, vp, vp]),
    "hd_retinanet_loss": (C.c_int, [vp] * 6 + [C.c_int] * 4 + [C.c_float] * 3 + [vp, vp, vp, vp, vp]),
    "hd_retinanet_loss_bwd": (C.c_int, [vp] * 6 + [C.c_int] * 4 + [C.c_float] * 3 + [vp] * 7),
    "hd_sigmoid_focal_loss": (C.c_int, [vp, vp, C.c_int64, C.c_float, C.c_float, vp, vp, vp]),
    "hd_groupnorm8_relu": (C.c_int, [vp] * 5 + [C.c_int] * 3 + [C.c_float, C.c_int, vp]),
    "hd_groupnorm8_relu_bwd": (C.c_int, [vp] * 6 + [C.c_int] * 4 + [vp]),
    "hd_groupnorm8_param_grad": (C.c_int, [vp] * 6 + [C.c_int] * 4 + [C.c_float, C.c_int, vp]),
    "hd_fcos_match": (C.c_int, [vp] * 3 + [C.c_int] * 5 + [C.c_float, vp, vp]),
    "hd_fcos_loss": (C.c_int, [vp] * 7 + [C.c_int] * 4 + [C.c_float] * 2 + [vp] * 4),
    "hd_fcos_loss_bwd": (C.c_int, [vp] * 7 + [C.c_int] * 4 + [C.c_float] * 2 + [vp] * 6),
    "hd_fastrcnn_loss": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, vp, vp, vp]),
    "hd_fastrcnn_loss_bwd": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, vp, vp, vp, vp, vp]),
    "hd_fastrcnn_loss_masked": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, vp, vp, vp, vp]),
    "hd_fastrcnn_loss_masked_bwd": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, vp, vp, vp, vp, vp, vp]),
    "hd_sample_pos_neg": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    "hd_roi_postprocess": (C.c_int, [vp, vp, vp, C.c_long, vp, C.c_int, C.c_int, C.c_int, vp] + [C.c_float] * 5 + [vp] * 4),
    "hd_roi_samples_padded": (C.c_int, [vp] * 7 + [C.c_int] * 4 + [vp] * 6),
    "hd_roi_samples_finish": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    "hd_roi_levels": (C.c_int, [vp, C.c_long, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, vp, vp]),
    "hd_batched_nms_pick": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "hd_batched_nms_pick_segments": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, C.c_float, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "hd_rpn_decode_filter": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, vp, vp, vp, vp]),
    "hd_roi_decode_clip": (C.c_int, [vp, vp, C.c_long, C.c_int, C.c_int, vp, C.c_float, C.c_float, C.c_float, vp, vp]),
    "hd_topk_select_rows": (C.c_int, [vp, C.c_int, C.c_long, vp, C.c_int, C.c_int, vp, C.c_long, vp]),
    "hd_match_targets": (C.c_int, [vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, vp, vp, vp, vp, vp, vp]),
    "hd_adam_step": (C.c_int, [vp, vp, vp, vp, c_i64] + [c_f] * 9 + [vp, vp]),
    "hd_check_finite": (C.c_int, [vp, c_i64, vp, vp]),
}

# fp32-storage twins (include/hallucidet_hip.h, last section): same signatures under the suffix _f32
F32_TWINS = ["hd_conv2d", "hd_conv2d_stats_rows", "hd_wgrad", "hd_weight_prep", "hd_bn_apply", "hd_bn_bwd_reduce", "hd_bn_bwd_apply",
             "hd_maxpool3x3s2", "hd_maxpool3x3s2_bwd", "hd_maxpool3x3s2_idx", "hd_maxpool3x3s2_bwd_idx", "hd_maxpool3x3s2_bwd_idx_add", "hd_concat_up_bwd", "hd_subsample2", "hd_subsample2_bwd",
             "hd_nchw_to_nhwc_resize", "hd_nchw_to_nhwc_resize_strided", "hd_nchw_to_nhwc_resize_bwd", "hd_nhwc_to_nchw", "hd_upsample_add",
             "hd_upsample_add_bwd", "hd_upsample2_bwd", "hd_add_f16", "hd_slice_channels", "hd_sigmoid_bwd_nchw_to_nhwc", "hd_relu_bwd",
             "hd_f32_to_f16", "hd_f16_to_f32", "hd_pad_cast_f32_f16", "hd_pad_cast_f32_f16_multi", "hd_channel_sum_f16", "hd_roi_align", "hd_roi_align_bwd",
             "hd_roi_align_ml", "hd_roi_align_ml_bwd", "hd_roi_align_ml_bwd_gather", "hd_groupnorm8_relu", "hd_groupnorm8_relu_bwd",
             "hd_groupnorm8_param_grad"]
for _n in F32_TWINS:
    PROTOTYPES[_n + "_f32"] = PROTOTYPES[_n]

_lib = None


def fn(name, t):
    """The entry point `name` for the storage type of tensor `t`: fp16 -> `name`, fp32 -> `name_f32`."""
    import torch
    lib = load()
    if t.dtype == torch.float32:
        return getattr(lib, name + "_f32")
    if t.dtype != torch.float16:
        raise TypeError("hallucidet_amd: activations are float16 or float32 (got %s)" % t.dtype)
    return getattr(lib, name)


def load():
    """Load the shared library (once).  Raises HipLibraryMissing when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            "libhallucidet_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `python hallucidet_amd/build.py`. There is no CPU fallback." % LIB_PATH)
    # PyTorch ships its own libamdhip64: it must be the HIP runtime of the process.  Loading this library BEFORE torch would pull in
    # the system copy, and torch's copy would then find no device ("no ROCm-capable device is detected" on the first launch).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    lib.hd_abi_version.restype = C.c_int
    got = lib.hd_abi_version()
    if got != ABI_VERSION:
        # a library built from other sources reads structs of another size: fail before the first call passes it one
        raise HipLibraryMissing(
            "%s reports hd_abi_version() == %d, this binding is written for %d: rebuild it "
            "(`python hallucidet_amd/build.py --force`)" % (LIB_PATH, got, ABI_VERSION))
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().hd_last_error().decode("utf-8", "replace")
        raise HipCallError("%s failed with status %d: %s" % (what, rc, msg))


def ptr(t):
    """Device (or host) pointer of a torch tensor, None -> NULL."""
    return None if t is None else t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
