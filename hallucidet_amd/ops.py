"""Thin tensor-level wrappers over the C ABI (one function per entry point family).

Inputs/outputs are torch tensors living on the GPU; activations are NHWC float16 -- or NHWC float32 (`--precision 32`, the reference's
default): every wrapper takes the storage type from its input tensor and calls the `_f32` twin of the entry point (hallucidet_amd._abi.fn;
include/hallucidet_hip.h, last section).  A parity mode, not a fast path.
Nothing here computes on the host and nothing falls back to ATen: a missing library
or a failing launch raises (``_abi.HipLibraryMissing`` / ``_abi.HipCallError``).
"""
import ctypes as C
import os
import threading

import torch

from . import _abi
from ._abi import ConvArgs, WgradArgs, check, ptr

ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2

# Storage type of NEW activation / packed-weight tensors where no input tensor decides it (layout conversions from fp32 images, weight
# re-packs, head-gradient casts): float16, or float32 inside `with ops.storage(torch.float32):` -- entered by the modules that were
# built with precision=32 (EncoderDecoderLit / DetectorLit / Detector.calculate_loss) around their forward and backward passes.
_tls = threading.local()


def act_dtype():
    return getattr(_tls, "dtype", torch.float16)


class storage:
    def __init__(self, dtype):
        assert dtype in (torch.float16, torch.float32)
        self.dtype = dtype

    def __enter__(self):
        self.prev = getattr(_tls, "dtype", torch.float16)
        _tls.dtype = self.dtype
        return self

    def __exit__(self, *exc):
        _tls.dtype = self.prev
        return False


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("hallucidet_amd.ops: tensors must live on the GPU (got %s); there is no CPU path" % t.device)


def conv_out_size(h, k, stride, pad):
    return (h + 2 * pad - k) // stride + 1


def conv2d(x, w, KH, KW, *, x2=None, bias=None, res=None, mask=None, stride=1, pad=0, up1=False, in_dil=1, act=ACT_NONE,
           out_nchw_f32=False, out_nhwc_f32=False, want_stats=False, out_hw=None, cout=None, out=None,
           in_scale=None, in_shift=None, in_relu=True, bstat=None, pool2=None, _defer=None):
    """Implicit-GEMM convolution.  x: [N,Hs,Ws,C1] f16, w: [Cout, KH*KW*(C1+C2)] f16.

    ``bstat`` (dict y, z|None, mean, invstd, gamma, beta, relu): the output of this call is the incoming gradient of a BatchNorm unit
    with raw conv output ``y``; where the kernel supports it (hd_conv2d_bstat_ok) the unit's backward sums leave with the call and
    ``bstat["part"]`` receives the [rows, 2*Cout] tensor `bn_backward(part=...)` takes; otherwise ``bstat["part"]`` is None.

    ``pool2`` (dict, optional key c_up = Cout): ask for the first c_up output channels 2 x 2 SUM-POOLED, [N, Ho/2, Wo/2, c_up] (returned),
    and the rest unpooled in ``pool2["skip"]`` (hd_conv_args.out_pool2 / y2: the data gradient of a decoder convolution over
    cat([nearest_2x(a), skip])); ``pool2["done"]`` says whether the kernel did it (else the plain output is returned and the caller
    runs concat_up_bwd).

    ``out_hw`` overrides the output extent (required with in_dil>1: data-gradient of strided convs).
    ``in_scale`` / ``in_shift`` ([C1] fp32): consumer-side BatchNorm -- x holds the RAW output of the producing conv and the kernel
    reads relu(fp16(x * scale + shift)) in its place (small-channel 3x3 kernel only; bit-identical to bn_apply + the plain call).
    Returns y or (y, stats_slab[rows,2,Cout]).
    """
    _need_cuda(x, w, x2, bias, res, mask)
    lib = _abi.load()
    N, Hs, Ws, C1 = x.shape
    C2 = 0 if x2 is None else x2.shape[3]
    Cout = w.shape[0] if cout is None else cout
    assert x.dtype in (torch.float16, torch.float32) and w.dtype == x.dtype and x.is_contiguous() and w.is_contiguous()
    assert w.numel() >= Cout * KH * KW * (C1 + C2), "weight tensor too small"
    if in_dil > 1:
        Hin, Win = 0, 0
        assert out_hw is not None
        Ho, Wo = out_hw
    else:
        Hin, Win = (Hs * 2, Ws * 2) if up1 else (Hs, Ws)
        if x2 is not None:
            assert x2.shape[1] == Hin and x2.shape[2] == Win and x2.is_contiguous()
        if out_hw is None:
            Ho, Wo = conv_out_size(Hin, KH, stride, pad), conv_out_size(Win, KW, stride, pad)
        else:
            Ho, Wo = out_hw
    def alloc_y():
        if out is not None:
            return out
        if out_nchw_f32:
            return torch.empty((N, Cout, Ho, Wo), dtype=torch.float32, device=x.device)
        if out_nhwc_f32:
            return torch.empty((N, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
        return torch.empty((N, Ho, Wo, Cout), dtype=x.dtype, device=x.device)

    a = ConvArgs(ptr(x), ptr(x2), ptr(w), ptr(bias), ptr(res), ptr(mask), None, None,
                 N, Hs, Ws, Hin, Win, C1, C2, Ho, Wo, Cout, KH, KW, stride, pad,
                 1 if up1 else 0, in_dil, act, 1 if out_nchw_f32 else (2 if out_nhwc_f32 else 0),
                 ptr(in_scale), ptr(in_shift), 1 if in_relu else 0, 0)
    y, y2 = None, None
    if pool2 is not None:
        pool2["done"] = False
        c_up = int(pool2.get("c_up", Cout))
        if x.dtype == torch.float16 and out is None and not want_stats and bstat is None and Ho % 2 == 0 and Wo % 2 == 0:
            # the pooled pair is probed BEFORE the full-resolution output exists: where the kernel takes the request that tensor (168 MB
            # for the last decoder block at 8 x 512 x 640) is exactly what the feature avoids, and under graph capture even an unused
            # allocation raises the private pool's high-water mark
            yp = torch.empty((N, Ho // 2, Wo // 2, c_up), dtype=x.dtype, device=x.device)
            y2 = torch.empty((N, Ho, Wo, Cout - c_up), dtype=x.dtype, device=x.device) if Cout > c_up else None
            a.out_pool2, a.y, a.y2 = c_up, ptr(yp), ptr(y2)
            if lib.hd_conv2d_pool2_ok(C.byref(a)) == 1:
                y = yp
                pool2["done"], pool2["skip"] = True, y2
            else:
                a.out_pool2, a.y2, y2 = 0, None, None
                del yp
    if y is None:
        y = alloc_y()
        a.y = ptr(y)
    stats = None
    if bstat is not None:
        bstat["part"] = None
        if x.dtype == torch.float16 and not want_stats and lib.hd_conv2d_bstat_ok(C.byref(a)) == 1:
            rows = lib.hd_conv2d_stats_rows(C.byref(a))
            check(0 if rows > 0 else rows, "hd_conv2d_stats_rows")
            part = torch.empty((rows, 2 * Cout), dtype=torch.float32, device=x.device)
            by, bz = bstat["y"], bstat.get("z")
            assert by.shape == y.shape and by.dtype == y.dtype and by.is_contiguous() and (bz is None or (bz.shape == y.shape and bz.is_contiguous()))
            a.stats, a.bs_y, a.bs_z = ptr(part), ptr(by), ptr(bz)
            a.bs_mean, a.bs_invstd, a.bs_gamma, a.bs_beta = ptr(bstat["mean"]), ptr(bstat["invstd"]), ptr(bstat.get("gamma")), ptr(bstat.get("beta"))
            a.bs_relu = 1 if bstat.get("relu", True) else 0
            bstat["part"] = part
    if want_stats:
        rows = _abi.fn("hd_conv2d_stats_rows", x)(C.byref(a))
        check(0 if rows > 0 else rows, "hd_conv2d_stats_rows")
        stats = torch.empty((rows, 2, Cout), dtype=torch.float32, device=x.device)
        a.stats = ptr(stats)
    if _defer is not None:          # wgrad_dgrad: the caller launches (the argument block and its tensors are kept by the list)
        _defer.append((a, (x, x2, w, bias, res, mask, y, y2, stats, in_scale, in_shift, bstat)))
    else:
        check(_abi.fn("hd_conv2d", x)(C.byref(a), _stream()), "hd_conv2d")
    return (y, stats) if want_stats else y


def wgrad(x, dy, KH, KW, *, x2=None, stride=1, pad=0, up1=False, nsplit=None, in_scale=None, in_shift=None, in_relu=True, _defer=None,
          dw=None, dw_scale=1.0):
    """Returns fp32 slab [nsplit, Cout, KH*KW*(C1+C2)] of partial weight gradients.  ``in_scale`` / ``in_shift``: as in conv2d
    (x is the raw output of the producing conv; small-channel 3x3 kernel only).  ``dw`` (fp32 OIHW, with nsplit == 1 where
    hd_wgrad_direct_ok): the kernel writes dw = dw_scale * gradient itself (== wgrad_reduce of the one slab, bit for bit) and None is
    returned; where the kernel cannot, the slab is returned and the caller reduces as usual."""
    _need_cuda(x, dy, x2)
    lib = _abi.load()
    N, Hs, Ws, C1 = x.shape
    C2 = 0 if x2 is None else x2.shape[3]
    _, Ho, Wo, Cout = dy.shape
    Hin, Win = (Hs * 2, Ws * 2) if up1 else (Hs, Ws)
    K = KH * KW * (C1 + C2)
    M = N * Ho * Wo
    if nsplit is None and x.dtype == torch.float32:
        nsplit = pick_nsplit(M, Cout, K)
    if nsplit is None:
        probe = WgradArgs(ptr(x), ptr(x2), ptr(dy), None, N, Hs, Ws, Hin, Win, C1, C2, Ho, Wo, Cout, KH, KW, stride, pad, 1 if up1 else 0, 1)
        blocks8 = lib.hd_wgrad_w8_blocks(C.byref(probe))
        if blocks8 > 0:
            # 8-wave patch kernel: one 512-thread block per CU, at least four 16x8-pixel tiles per block
            tiles = N * ((Ho + 15) // 16) * ((Wo + 7) // 8)
            nsplit = max(1, min(_WGRAD8_BLOCKS // blocks8, tiles // 4))
            while nsplit > 1 and nsplit * Cout * K * 4 > (64 << 20):
                nsplit //= 2
        elif x2 is not None and up1 and KH == 3 and C1 == 64 and C2 == 64 and Cout <= 32 and stride == 1 and pad == 1:
            nsplit = 256          # thin-output kernel on the decoder concat (wgrad3x3_small.hip): 125 KB of LDS, one persistent block per CU
        else:
            nsplit = pick_nsplit(M, Cout, K)
    direct = False
    if dw is not None and nsplit == 1 and x.dtype == torch.float16:
        probe = WgradArgs(ptr(x), ptr(x2), ptr(dy), None, N, Hs, Ws, Hin, Win, C1, C2, Ho, Wo, Cout, KH, KW, stride, pad, 1 if up1 else 0, 1,
                          ptr(in_scale), ptr(in_shift), 1 if in_relu else 0, 0)
        direct = lib.hd_wgrad_direct_ok(C.byref(probe)) == 1
        if direct:
            assert dw.dtype == torch.float32 and dw.is_contiguous() and dw.numel() == Cout * K
    slab = None if direct else torch.empty((nsplit, Cout, K), dtype=torch.float32, device=x.device)
    a = WgradArgs(ptr(x), ptr(x2), ptr(dy), ptr(slab), N, Hs, Ws, Hin, Win, C1, C2, Ho, Wo, Cout, KH, KW, stride, pad,
                  1 if up1 else 0, nsplit, ptr(in_scale), ptr(in_shift), 1 if in_relu else 0, 0, ptr(dw) if direct else None, float(dw_scale), 0)
    if _defer is not None:
        _defer.append((a, (x, x2, dy, slab, in_scale, in_shift, dw)))
    else:
        check(_abi.fn("hd_wgrad", x)(C.byref(a), _stream()), "hd_wgrad")
    return slab


def stem_dgrad_weights(wf, cin_p=8):
    """The 16 x 1024 sub-pixel weight matrix of hd_conv7x7s2_dgrad_thin from the forward GEMM layout wf [64, 7*7*cin_p] (f16) of a
    7x7 / stride-2 / pad-3 convolution with <= 4 real input channels."""
    w = wf.view(64, 7, 7, cin_p).float()
    out = torch.zeros(16, 16, 64, dtype=torch.float32, device=wf.device)             # [row][tap][ch]
    for a in range(2):
        for b in range(2):
            for di in range(-1, 3):
                for dj in range(-1, 3):
                    kh, kw = a + 3 - 2 * di, b + 3 - 2 * dj
                    if 0 <= kh <= 6 and 0 <= kw <= 6:
                        for c in range(min(3, cin_p)):
                            out[(2 * a + b) * 4 + c, (di + 1) * 4 + dj + 1] = w[:, kh, kw, c]
    return out.view(16, 1024).half().contiguous()


def conv7x7s2_dgrad_thin(dy, w16, in_hw, mask_z=None, out=None):
    """Data gradient of the ResNet stem convolution in sub-pixel form (hd_conv7x7s2_dgrad_thin): dy [N,Hl,Wl,64] f16 -> [N,H,W,8] f16
    (`out`: written in place when given)."""
    _need_cuda(dy, w16, mask_z)
    N, Hl, Wl, C_ = dy.shape
    H, W = in_hw
    assert C_ == 64 and dy.dtype == torch.float16 and dy.is_contiguous() and w16.shape == (16, 1024) and w16.dtype == torch.float16
    assert mask_z is None or (mask_z.shape == dy.shape and mask_z.dtype == dy.dtype and mask_z.is_contiguous())
    if out is not None:
        assert out.shape == (N, H, W, 8) and out.dtype == torch.float16 and out.is_contiguous() and out.device == dy.device
    dx = out if out is not None else torch.empty((N, H, W, 8), dtype=torch.float16, device=dy.device)
    check(_abi.load().hd_conv7x7s2_dgrad_thin(ptr(dy), ptr(mask_z), ptr(w16), ptr(dx), N, Hl, Wl, H, W, _stream()), "hd_conv7x7s2_dgrad_thin")
    return dx


def conv2d_multi(calls):
    """calls: [(x, w, KH, KW, kwargs)] -- the arguments of independent `conv2d` calls -> their outputs, issued as ONE grid when all
    of them run in the same 4-wave implicit-GEMM variant (hd_conv2d_multi: the per-level convolutions of an FPN / a detection head),
    else one after the other.  Put the largest problem first (its tile serves all).  Bit-identical to the separate calls."""
    if len(calls) == 1 or calls[0][0].dtype == torch.float32:          # fp32 storage: the parity mode has no multi-problem grid
        return [conv2d(x, w, KH, KW, **kw) for x, w, KH, KW, kw in calls]
    hold, outs = [], []
    for x, w, KH, KW, kw in calls:
        outs.append(conv2d(x, w, KH, KW, _defer=hold, **kw))
    arr = (ConvArgs * len(hold))(*[h[0] for h in hold])
    check(_abi.load().hd_conv2d_multi(arr, len(hold), _stream()), "hd_conv2d_multi")
    return outs


WGRAD_MULTI_MAX = 24      # HD_WGRAD_MULTI_MAX


def wgrad_w8_blocks(x, dy, KH, KW, *, x2=None, stride=1, pad=0, up1=False, **_):
    """(Cin / 64) * (Cout / 64) if hd_wgrad routes this problem to the 8-wave patch-staged 3x3 kernel, else 0 (hd_wgrad_w8_blocks)."""
    if x.dtype != torch.float16:
        return 0
    N, Hs, Ws, C1 = x.shape
    C2 = 0 if x2 is None else x2.shape[3]
    _, Ho, Wo, Cout = dy.shape
    Hin, Win = (Hs * 2, Ws * 2) if up1 else (Hs, Ws)
    probe = WgradArgs(ptr(x), ptr(x2), ptr(dy), None, N, Hs, Ws, Hin, Win, C1, C2, Ho, Wo, Cout, KH, KW, stride, pad, 1 if up1 else 0, 1)
    return int(_abi.load().hd_wgrad_w8_blocks(C.byref(probe)))


def wgrad_takes_w8(x, dy, KH, KW, **kw):
    return wgrad_w8_blocks(x, dy, KH, KW, **kw) > 0


def wgrad_multi(calls):
    """calls: [(x, dy, KH, KW, kwargs)] -- the arguments of independent `wgrad` calls -> their slabs; ONE grid per <= 24 calls (HD_WGRAD_MULTI_MAX) when all of
    them run in the 8-wave patch-staged 3x3 kernel (hd_wgrad_multi), else one launch each.  Bit-identical to the separate calls."""
    if not calls:
        return []
    if calls[0][0].dtype == torch.float32:
        return [wgrad(x, dy, KH, KW, **kw) for x, dy, KH, KW, kw in calls]
    slabs = []
    for i in range(0, len(calls), WGRAD_MULTI_MAX):
        hold = []
        for x, dy, KH, KW, kw in calls[i:i + WGRAD_MULTI_MAX]:
            slabs.append(wgrad(x, dy, KH, KW, _defer=hold, **kw))
        arr = (WgradArgs * len(hold))(*[h[0] for h in hold])
        check(_abi.load().hd_wgrad_multi(arr, len(hold), _stream()), "hd_wgrad_multi")
    return slabs


def wgrad_dgrad(x, dy, KH, KW, wd, *, x2=None, stride=1, pad=0, up1=False, in_scale=None, in_shift=None, in_relu=True, dgrad=None):
    """The two consumers of a layer's dY in one call (hd_conv2d_wgrad): -> (slab as `wgrad(x, dy, ...)`, dx as `conv2d(dy, wd, KH, KW,
    **dgrad)`).  One grid when both run in the 8-wave kernels, two launches otherwise; bit-identical to the separate calls."""
    if x.dtype == torch.float32:
        slab = wgrad(x, dy, KH, KW, x2=x2, stride=stride, pad=pad, up1=up1, in_scale=in_scale, in_shift=in_shift, in_relu=in_relu)
        return slab, conv2d(dy, wd, KH, KW, **(dgrad or {}))
    hold = []
    slab = wgrad(x, dy, KH, KW, x2=x2, stride=stride, pad=pad, up1=up1, in_scale=in_scale, in_shift=in_shift, in_relu=in_relu, _defer=hold)
    dx = conv2d(dy, wd, KH, KW, _defer=hold, **(dgrad or {}))
    check(_abi.load().hd_conv2d_wgrad(C.byref(hold[1][0]), C.byref(hold[0][0]), _stream()), "hd_conv2d_wgrad")
    return slab, dx


_WGRAD_BLOCKS = int(os.environ.get("HD_WGRAD_BLOCKS", "0"))   # 0: per-class targets below
_WGRAD8_BLOCKS = int(os.environ.get("HD_WGRAD8_BLOCKS", "256"))   # one 512-thread block per CU: every block writes a 147 KB fp32 partial, so more blocks = more slab traffic (swept 128 / 256 / 512)


def pick_nsplit(M, Cout, K, target_blocks=None):
    """Split count of the pixel reduction.  Per-layer sweep (tools/tune_wgrad.py; hd_wgrad + hd_wgrad_reduce timed
    together): the 32-row kernel of the thin decoder layers wants ~1500 blocks, the 128-row kernel ~500 (slab traffic grows
    with the split), the 256/512-channel layers ~750."""
    if target_blocks is None:
        target_blocks = _WGRAD_BLOCKS or (1536 if Cout <= 32 else (768 if Cout >= 256 else 512))
    tm = 128 if Cout > 64 else (64 if Cout > 32 else 32)
    tiles = ((K + 127) // 128) * ((Cout + tm - 1) // tm)
    ns = max(1, min(target_blocks // max(tiles, 1), M // 256))
    # bound slab size to 64 MiB
    while ns > 1 and ns * Cout * K * 4 > (64 << 20):
        ns //= 2
    return max(1, ns)


def wgrad_reduce(slab, dw, KH, KW, Cin, Cin_real=None, Cout=None, scale=1.0, accumulate=False):
    _need_cuda(slab, dw)
    nsplit, Cout_slab, K = slab.shape
    Cout = Cout_slab if Cout is None else Cout
    Cin_real = Cin if Cin_real is None else Cin_real
    assert dw.numel() == Cout * Cin_real * KH * KW and dw.dtype == torch.float32 and dw.is_contiguous()
    check(_abi.load().hd_wgrad_reduce(ptr(slab), ptr(dw), nsplit, Cout_slab, Cout, KH, KW, Cin, Cin_real, scale,
                                      1 if accumulate else 0, _stream()), "hd_wgrad_reduce")
    return dw


def weight_prep(w_oihw, *, out_scale=None, cin_pad=None, cout_pad=None, want_fwd=True, want_dgrad=False, dtype=None):
    """fp32 OIHW -> ([Cout, KH*KW*Cin_pad], [Cin_pad, KH*KW*Cout_pad] flipped) in the storage type `dtype` (float16 / float32)."""
    _need_cuda(w_oihw)
    Cout, Cin, KH, KW = w_oihw.shape
    cin_pad = cin_pad or ((Cin + 7) // 8 * 8)
    cout_pad = cout_pad or ((Cout + 7) // 8 * 8)
    dtype = dtype or act_dtype()
    wf = torch.empty((Cout, KH * KW * cin_pad), dtype=dtype, device=w_oihw.device) if want_fwd else None
    wd = torch.empty((cin_pad, KH * KW * cout_pad), dtype=dtype, device=w_oihw.device) if want_dgrad else None
    check(_abi.fn("hd_weight_prep", wf if wf is not None else wd)(ptr(w_oihw.contiguous()), ptr(out_scale), ptr(wf), ptr(wd), Cout, Cin, KH, KW,
                                     cin_pad, cout_pad, _stream()), "hd_weight_prep")
    return wf, wd


# A batch is launched as soon as it holds this many bytes of slabs (not only at the end of a backward segment): slabs that were written
# by the last two or three weight-gradient launches are still in the 256 MB MALL when the reduction reads them.  Swept 30 / 40 / 80 /
# 110 / 150 MB and "segment end only" on two boxes: 80 - 110 MB is best, -0.08 ms per training step against segment end only (eleven more
# launches per step included); one launch per tensor (30) is slower than segment end only.  0 = segment end only.
_WRED_FLUSH_BYTES = int(float(os.environ.get("HD_WRED_FLUSH_MB", "100")) * 1e6)


class WgradReduceBatch:
    """Deferred hd_wgrad_reduce calls of one backward segment, issued as ONE launch per 16 tensors (hd_wgrad_reduce_multi): `add()`
    has the signature of `wgrad_reduce` and keeps the slab alive until `flush()`, which plans the grid on the host and launches
    (the descriptor table travels in the kernel arguments: capture-safe, nothing to keep alive afterwards).  Per-tensor summation
    order = hd_wgrad_reduce's: bit-identical results."""
    MAX = 16      # HD_WRED_MAX

    def __init__(self):
        self.items = []

    def add(self, slab, dw, KH, KW, Cin, Cin_real=None, Cout=None, scale=1.0, accumulate=False):
        _need_cuda(slab, dw)
        nsplit, Cout_slab, K = slab.shape
        Cout = Cout_slab if Cout is None else Cout
        Cin_real = Cin if Cin_real is None else Cin_real
        assert dw.numel() == Cout * Cin_real * KH * KW and dw.dtype == torch.float32 and dw.is_contiguous()
        self.items.append((slab, dw, (nsplit, Cout_slab, Cout, KH, KW, Cin, Cin_real, 1 if accumulate else 0, float(scale))))
        if _WRED_FLUSH_BYTES and sum(s.numel() for s, _, _ in self.items) * 4 >= _WRED_FLUSH_BYTES:
            self.flush()

    def flush(self):
        from ._abi import WredDesc
        lib = _abi.load()
        items, self.items = self.items, []
        for lo in range(0, len(items), self.MAX):
            part = items[lo:lo + self.MAX]
            n = len(part)
            arr = (WredDesc * n)(*[WredDesc(slab.data_ptr(), dw.data_ptr(), *a[:8], a[8], 0, 0, 0) for slab, dw, a in part])
            blocks = lib.hd_wgrad_reduce_plan(C.cast(arr, C.c_void_p), n)
            if blocks <= 0:
                check(blocks if blocks < 0 else -1, "hd_wgrad_reduce_plan")
            check(lib.hd_wgrad_reduce_multi(C.cast(arr, C.c_void_p), n, blocks, _stream()), "hd_wgrad_reduce_multi")


class WeightPrepPlan:
    """All (fp32 master -> fp16 GEMM layouts) conversions of a network as ONE launch: persistent output buffers and a
    device-resident descriptor table built once; `run()` re-packs every layer (hd_weight_prep_multi)."""

    def __init__(self, items, dtype=None):
        """items: list of (w_oihw fp32 parameter, cin_pad, cout_pad, want_dgrad).  dtype float32: the same persistent buffers, re-packed
        by one hd_weight_prep_f32 launch per layer (the parity mode has no multi-layer kernel)."""
        from ._abi import WprepDesc
        dev = items[0][0].device
        dtype = dtype or act_dtype()
        self.dtype, self.items = dtype, list(items)
        self.outputs, descs, biggest = [], [], 1
        for w, cin_pad, cout_pad, want_d in items:
            Cout, Cin, KH, KW = w.shape
            wf = torch.empty((Cout, KH * KW * cin_pad), dtype=dtype, device=dev)
            wd = torch.empty((cin_pad, KH * KW * cout_pad), dtype=dtype, device=dev) if want_d else None
            self.outputs.append((wf, wd))
            descs.append(WprepDesc(w.data_ptr(), wf.data_ptr(), wd.data_ptr() if want_d else None, Cout, Cin, KH, KW, cin_pad, cout_pad))
            # blocks a layer can use: one per 32(co) x 32(ci) tile (element-wise form for > 9 taps: 1024 elements per block)
            if KH * KW <= 9:
                biggest = max(biggest, ((max(Cout, cout_pad if want_d else 0) + 31) // 32) * ((cin_pad + 31) // 32))
            else:
                biggest = max(biggest, (Cout * KH * KW * cin_pad + 1023) // 1024)
        self._ptrs = [w.data_ptr() for w, *_ in items]
        raw = bytes(bytearray().join(bytes(d) for d in descs))
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        self.n = len(descs)
        self.blocks = int(min(2048, biggest))          # grid.x of the one launch: the largest layer's tile count (layers with fewer tiles leave blocks idle)

    def valid_for(self, params):
        return len(params) == self.n and all(p.data_ptr() == q for p, q in zip(params, self._ptrs))

    def run(self):
        if self.dtype == torch.float32:
            lib = _abi.load()
            for (w, cin_pad, cout_pad, want_d), (wf, wd) in zip(self.items, self.outputs):
                Cout, Cin, KH, KW = w.shape
                check(lib.hd_weight_prep_f32(ptr(w), None, ptr(wf), ptr(wd), Cout, Cin, KH, KW, cin_pad, cout_pad, _stream()), "hd_weight_prep_f32")
            return self.outputs
        check(_abi.load().hd_weight_prep_multi(ptr(self.table), self.n, self.blocks, _stream()), "hd_weight_prep_multi")
        return self.outputs


def colsum(slab2d):
    """[rows, W] fp32 -> [W] (deterministic)."""
    _need_cuda(slab2d)
    rows, W = slab2d.shape
    out = torch.empty((W,), dtype=torch.float32, device=slab2d.device)
    ws = torch.empty((128 * W,), dtype=torch.float32, device=slab2d.device) if rows > 32 else None
    check(_abi.load().hd_colsum(ptr(slab2d), rows, W, ptr(out), ptr(ws), _stream()), "hd_colsum")
    return out


def rowsum(slab2d, out_rows):
    """[rows, W] fp32 -> [out_rows, W]: one deterministic reduction stage."""
    rows, W = slab2d.shape
    out = torch.empty((out_rows, W), dtype=torch.float32, device=slab2d.device)
    check(_abi.load().hd_rowsum(ptr(slab2d), rows, W, ptr(out), out_rows, _stream()), "hd_rowsum")
    return out


def bn_finalize(part, count, gamma, beta, running_mean, running_var, momentum, eps):
    """part: [rows, 2, C] (or [2C]) partial sums from the conv epilogue."""
    if part.dim() == 3:
        part = part.reshape(part.shape[0], -1)
    elif part.dim() == 1:
        part = part.reshape(1, -1)
    if part.shape[0] > 4096:       # (never on the hot path) the finalize kernel sums any number of rows; keep its loop short
        part = rowsum(part, 1024)
    rows, W = part.shape
    C_ = W // 2
    dev = part.device
    mean = torch.empty(C_, dtype=torch.float32, device=dev)
    invstd = torch.empty_like(mean)
    scale = torch.empty_like(mean)
    shift = torch.empty_like(mean)
    check(_abi.load().hd_bn_finalize(ptr(part), rows, C_, float(count), ptr(gamma), ptr(beta), ptr(running_mean),
                                     ptr(running_var), momentum, eps, ptr(mean), ptr(invstd), ptr(scale), ptr(shift),
                                     _stream()), "hd_bn_finalize")
    return mean, invstd, scale, shift


def bn_eval_scale_shift(gamma, beta, running_mean, running_var, eps):
    C_ = running_mean.numel()
    scale = torch.empty(C_, dtype=torch.float32, device=running_mean.device)
    shift = torch.empty_like(scale)
    check(_abi.load().hd_bn_eval_scale_shift(ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), eps, C_,
                                             ptr(scale), ptr(shift), _stream()), "hd_bn_eval_scale_shift")
    return scale, shift


def bn_apply(y, scale, shift, *, res=None, relu=True, out=None):
    _need_cuda(y, scale, shift, res)
    z = torch.empty_like(y) if out is None else out
    check(_abi.fn("hd_bn_apply", y)(ptr(y), ptr(res), ptr(scale), ptr(shift), ptr(z), y.numel(), y.shape[-1],
                                  1 if relu else 0, _stream()), "hd_bn_apply")
    return z


def bn_backward(dz, z, y, mean, invstd, gamma, beta=None, *, relu=True, want_dres=False, gscale=1.0, dgamma=None, dbeta=None,
                accumulate=False, rows=None, part=None):
    """Backward of z = relu(bn_train(y) (+res)).  `z=None` (no residual): the ReLU mask is recomputed from y.  `part` [rows, 2C]: the
    reduction's partial rows when the kernel that produced dz already emitted them (conv2d(bstat=...)): the reduction pass is skipped.
    Returns (dy, dres|None, dgamma, dbeta)."""
    _need_cuda(dz, y)
    C_ = y.shape[-1]
    npix = y.numel() // C_
    lib = _abi.load()
    if part is None:
        if rows is None:
            rows = int(max(1, min(512, npix // 64)))      # swept 128..4096 (tools/tune_bn.py): 512 is at or within 1 % of the best everywhere; npix // 16 and // 8 for the small tensors: no change (those launches are latency chains, not bandwidth)
        part = torch.empty((rows, 2 * C_), dtype=torch.float32, device=y.device)
        check(_abi.fn("hd_bn_bwd_reduce", y)(ptr(dz), ptr(z), ptr(y), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), ptr(part), rows, npix, C_,
                                   1 if relu else 0, _stream()), "hd_bn_bwd_reduce")
    else:
        assert part.dtype == torch.float32 and part.dim() == 2 and part.shape[1] == 2 * C_ and part.is_contiguous()
    coef = torch.empty((5, C_), dtype=torch.float32, device=y.device)      # A, B, D, scale, shift: written by the coefficient launch
    dy = torch.empty_like(y)
    dres = torch.empty_like(y) if want_dres else None
    if dgamma is None:
        dgamma = torch.empty(C_, dtype=torch.float32, device=y.device)
    if dbeta is None:
        dbeta = torch.empty(C_, dtype=torch.float32, device=y.device)
    check(_abi.fn("hd_bn_bwd_apply", y)(ptr(dz), ptr(z), ptr(y), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), ptr(part), part.shape[0],
                              ptr(coef), ptr(dy), ptr(dres), ptr(dgamma), ptr(dbeta), gscale, 1 if accumulate else 0, npix, C_,
                              1 if relu else 0, _stream()), "hd_bn_bwd_apply")
    return dy, dres, dgamma, dbeta


def groupnorm8_relu(x, gamma, beta, eps=1e-5, *, relu=True):
    """GroupNorm with 8 channels per group (+ReLU) over NHWC f16 [N, H, W, C] -> (y, mean_rstd [N, C/8, 2])."""
    _need_cuda(x, gamma, beta)
    N, H, W, C_ = x.shape
    y = torch.empty_like(x)
    stat = torch.empty((N, C_ // 8, 2), dtype=torch.float32, device=x.device)
    check(_abi.fn("hd_groupnorm8_relu", x)(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(stat), N, H * W, C_, float(eps), 1 if relu else 0, _stream()),
          "hd_groupnorm8_relu")
    return y, stat


def groupnorm8_relu_bwd(dy, x, y, gamma, stat, *, relu=True):
    """Data gradient of groupnorm8_relu (y = the forward output, used as the ReLU mask)."""
    _need_cuda(dy, x, gamma, stat)
    N, H, W, C_ = x.shape
    dx = torch.empty_like(x)
    check(_abi.fn("hd_groupnorm8_relu_bwd", x)(ptr(dy), ptr(x), ptr(y), ptr(gamma), ptr(stat), ptr(dx), N, H * W, C_, 1 if relu else 0, _stream()),
          "hd_groupnorm8_relu_bwd")
    return dx


def groupnorm8_param_grad(dy, x, y, stat, dgamma, dbeta, scale=1.0, *, relu=True, accumulate=True):
    """dgamma / dbeta (fp32 [C], accumulated in place) of groupnorm8_relu over the whole batch."""
    _need_cuda(dy, x, stat, dgamma, dbeta)
    N, H, W, C_ = x.shape
    check(_abi.fn("hd_groupnorm8_param_grad", x)(ptr(dy), ptr(x), ptr(y), ptr(stat), ptr(dgamma), ptr(dbeta), N, H * W, C_, 1 if relu else 0,
                                               float(scale), 1 if accumulate else 0, _stream()), "hd_groupnorm8_param_grad")


def pad_cast_f32_f16(x, cp, dtype=None):
    """x [n, H, W, C] fp32 whose images are dense (strides (*, W*C, C, 1); the image stride may be larger: a slice of a bigger
    buffer) -> [n, H, W, cp] fp16, extra channels zero (one launch)."""
    _need_cuda(x)
    n, H, W, C_ = x.shape
    assert x.stride()[1:] == (W * C_, C_, 1) or n * H * W == 0
    y = torch.empty((n, H, W, cp), dtype=dtype or act_dtype(), device=x.device)
    img_stride = x.stride(0) if n > 1 else H * W * C_
    check(_abi.fn("hd_pad_cast_f32_f16", y)(ptr(x), ptr(y), n * H * W, C_, cp, H * W, max(img_stride, H * W * C_), _stream()), "hd_pad_cast_f32_f16")
    return y


def pad_cast_f32_f16_many(xs, cps, dtype=None):
    """`pad_cast_f32_f16` for several tensors in ONE launch (hd_pad_cast_f32_f16_multi; <= 16 per launch) -> list of outputs."""
    if len(xs) == 1:
        return [pad_cast_f32_f16(xs[0], cps[0], dtype=dtype)]
    _need_cuda(*xs)
    outs = []
    for lo in range(0, len(xs), 16):
        part = xs[lo:lo + 16]
        n = len(part)
        ys, Ps, Cs, Cps, rpis, strides = [], [], [], [], [], []
        for x, cp in zip(part, cps[lo:lo + 16]):
            b, H, W, C_ = x.shape
            assert x.dtype == torch.float32 and (x.stride()[1:] == (W * C_, C_, 1) or b * H * W == 0)
            y = torch.empty((b, H, W, cp), dtype=dtype or act_dtype(), device=x.device)
            ys.append(y); Ps.append(b * H * W); Cs.append(C_); Cps.append(cp); rpis.append(max(H * W, 1))
            strides.append(max(x.stride(0) if b > 1 else H * W * C_, H * W * C_))
        arr = lambda ctype, vals: (ctype * n)(*vals)
        check(_abi.fn("hd_pad_cast_f32_f16_multi", ys[0])(arr(C.c_void_p, [x.data_ptr() for x in part]), arr(C.c_void_p, [y.data_ptr() for y in ys]),
                                                          arr(C.c_int64, Ps), arr(C.c_int, Cs), arr(C.c_int, Cps), arr(C.c_int64, rpis),
                                                          arr(C.c_int64, strides), n, _stream()), "hd_pad_cast_f32_f16_multi")
        outs += ys
    return outs


def maxpool3x3s2(x):
    N, H, W, C_ = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((N, Ho, Wo, C_), dtype=x.dtype, device=x.device)
    check(_abi.fn("hd_maxpool3x3s2", x)(ptr(x), ptr(y), N, H, W, C_, Ho, Wo, _stream()), "hd_maxpool3x3s2")
    return y


def maxpool3x3s2_idx(x):
    """-> (y, idx uint8 [N,Ho,Wo,C]) ; idx feeds maxpool3x3s2_bwd_idx."""
    N, H, W, C_ = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((N, Ho, Wo, C_), dtype=x.dtype, device=x.device)
    idx = torch.empty((N, Ho, Wo, C_), dtype=torch.uint8, device=x.device)
    check(_abi.fn("hd_maxpool3x3s2_idx", x)(ptr(x), ptr(y), ptr(idx), N, H, W, C_, Ho, Wo, _stream()), "hd_maxpool3x3s2_idx")
    return y, idx


def maxpool3x3s2_bwd_idx(idx, dy, in_hw, add=None):
    """Gradient of the 3x3 / stride-2 max-pool routed by the recorded winners; `add` [N,H,W,C]: a second gradient of the pooled
    tensor's input summed in the same pass (== maxpool3x3s2_bwd_idx followed by add_f16, bit for bit)."""
    N, Ho, Wo, C_ = dy.shape
    H, W = in_hw
    dx = torch.empty((N, H, W, C_), dtype=dy.dtype, device=dy.device)
    if add is None:
        check(_abi.fn("hd_maxpool3x3s2_bwd_idx", dy)(ptr(idx), ptr(dy.contiguous()), ptr(dx), N, H, W, C_, Ho, Wo, _stream()), "hd_maxpool3x3s2_bwd_idx")
    else:
        assert add.shape == dx.shape and add.dtype == dy.dtype and add.is_contiguous()
        check(_abi.fn("hd_maxpool3x3s2_bwd_idx_add", dy)(ptr(idx), ptr(dy.contiguous()), ptr(add), ptr(dx), N, H, W, C_, Ho, Wo, _stream()),
              "hd_maxpool3x3s2_bwd_idx_add")
    return dx


def concat_up_bwd(dcat, c_up):
    """Gradient of cat([nearest_2x(a), skip], channel): -> (da [N,H/2,W/2,c_up], dskip [N,H,W,C-c_up] or None), one launch
    (== upsample2_bwd + slice_channels, bit for bit)."""
    N, H, W, Ct = dcat.shape
    assert dcat.is_contiguous() and H % 2 == 0 and W % 2 == 0
    da = torch.empty((N, H // 2, W // 2, c_up), dtype=dcat.dtype, device=dcat.device)
    ds = torch.empty((N, H, W, Ct - c_up), dtype=dcat.dtype, device=dcat.device) if Ct > c_up else None
    check(_abi.fn("hd_concat_up_bwd", dcat)(ptr(dcat), ptr(da), ptr(ds) if ds is not None else None, N, H // 2, W // 2, c_up, Ct - c_up, _stream()),
          "hd_concat_up_bwd")
    return da, ds


def maxpool3x3s2_bwd(x, dy):
    N, H, W, C_ = x.shape
    _, Ho, Wo, _ = dy.shape
    dx = torch.empty_like(x)
    check(_abi.fn("hd_maxpool3x3s2_bwd", x)(ptr(x), ptr(dy), ptr(dx), N, H, W, C_, Ho, Wo, _stream()), "hd_maxpool3x3s2_bwd")
    return dx


def subsample2(x):
    N, H, W, C_ = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((N, Ho, Wo, C_), dtype=x.dtype, device=x.device)
    check(_abi.fn("hd_subsample2", x)(ptr(x), ptr(y), N, H, W, C_, Ho, Wo, _stream()), "hd_subsample2")
    return y


def subsample2_bwd(dy, dx, accumulate=True):
    N, H, W, C_ = dx.shape
    _, Ho, Wo, _ = dy.shape
    check(_abi.fn("hd_subsample2_bwd", dy)(ptr(dy), ptr(dx), N, H, W, C_, Ho, Wo, 1 if accumulate else 0, _stream()),
          "hd_subsample2_bwd")
    return dx


def nchw_to_nhwc_resize(x, Ho, Wo, Cp=8, out=None, dtype=None):
    """`out`: write into this [N, Ho, Wo, Cp] tensor (e.g. a slice along dim 0 of a larger batch buffer) instead of a new one; `dtype`:
    storage type of a new one (float16 / float32)."""
    _need_cuda(x)
    N, Cr, H, W = x.shape
    assert x.dtype == torch.float32
    if out is None:
        y = torch.empty((N, Ho, Wo, Cp), dtype=dtype or act_dtype(), device=x.device)
    else:
        y = out
        assert y.shape == (N, Ho, Wo, Cp) and y.dtype in (torch.float16, torch.float32) and y.is_contiguous()
    if x.is_contiguous():
        check(_abi.fn("hd_nchw_to_nhwc_resize", y)(ptr(x), ptr(y), N, Cr, H, W, Ho, Wo, Cp, _stream()), "hd_nchw_to_nhwc_resize")
    else:
        # dense planes with arbitrary image / channel strides: the stride-0 channel view of a 1 -> 3 channel `expand`
        assert dense_planes(x), "nchw_to_nhwc_resize: rows of x must be dense (stride (.., .., W, 1))"
        check(_abi.fn("hd_nchw_to_nhwc_resize_strided", y)(ptr(x), x.stride(0), x.stride(1), ptr(y), N, Cr, H, W, Ho, Wo, Cp, _stream()),
              "hd_nchw_to_nhwc_resize_strided")
    return y


def dense_planes(x):
    """[N, C, H, W] tensor whose H x W planes are dense (any image / channel stride, e.g. `t.expand(-1, 3, -1, -1)`)."""
    return x.dim() == 4 and x.stride(3) == 1 and x.stride(2) == x.shape[3]


def as_dense_planes_f32(x):
    """fp32 view / copy of an NCHW batch that hd_nchw_to_nhwc_resize can read: contiguous tensors and stride-0 channel views of a
    dense single-channel batch pass through untouched; anything else is made contiguous."""
    x = x if x.dtype == torch.float32 else x.float()
    return x if (x.is_contiguous() or dense_planes(x)) else x.contiguous()


def nchw_to_nhwc_resize_bwd(dy, N, Cr, H, W, gscale=1.0):
    _, Ho, Wo, Cp = dy.shape
    dx = torch.empty((N, Cr, H, W), dtype=torch.float32, device=dy.device)
    check(_abi.fn("hd_nchw_to_nhwc_resize_bwd", dy)(ptr(dy), ptr(dx), N, Cr, H, W, Ho, Wo, Cp, gscale, _stream()),
          "hd_nchw_to_nhwc_resize_bwd")
    return dx


def nhwc_to_nchw(x, Cr=None):
    N, H, W, Cp = x.shape
    Cr = Cp if Cr is None else Cr
    y = torch.empty((N, Cr, H, W), dtype=torch.float32, device=x.device)
    check(_abi.fn("hd_nhwc_to_nchw", x)(ptr(x), ptr(y), N, Cr, H, W, Cp, _stream()), "hd_nhwc_to_nchw")
    return y


def upsample_add(a, b):
    N, H, W, C_ = a.shape
    _, Hb, Wb, _ = b.shape
    y = torch.empty_like(a)
    check(_abi.fn("hd_upsample_add", a)(ptr(a), ptr(b), ptr(y), N, H, W, C_, Hb, Wb, _stream()), "hd_upsample_add")
    return y


def upsample_add_bwd(dy, db, accumulate):
    N, H, W, C_ = dy.shape
    _, Hb, Wb, _ = db.shape
    check(_abi.fn("hd_upsample_add_bwd", dy)(ptr(dy), ptr(db), N, H, W, C_, Hb, Wb, 1 if accumulate else 0, _stream()),
          "hd_upsample_add_bwd")
    return db


def upsample2_bwd(dy_up, dx_low, c_off, accumulate):
    N, Hl, Wl, C_ = dx_low.shape
    Ctot = dy_up.shape[3]
    check(_abi.fn("hd_upsample2_bwd", dy_up)(ptr(dy_up), ptr(dx_low), N, Hl, Wl, C_, Ctot, c_off, 1 if accumulate else 0,
                                       _stream()), "hd_upsample2_bwd")
    return dx_low


def add_f16(a, b, out=None):
    out = torch.empty_like(a) if out is None else out
    check(_abi.fn("hd_add_f16", a)(ptr(a), ptr(b), ptr(out), a.numel(), _stream()), "hd_add_f16")
    return out


def slice_channels(x, y, c_off, accumulate):
    Ctot = x.shape[-1]
    C_ = y.shape[-1]
    npix = y.numel() // C_
    check(_abi.fn("hd_slice_channels", x)(ptr(x), ptr(y), npix, Ctot, c_off, C_, 1 if accumulate else 0, _stream()),
          "hd_slice_channels")
    return y


def sigmoid_bwd_nchw_to_nhwc(dy, s, Cp=8, gscale=1.0, dtype=None):
    N, Cr, H, W = s.shape
    dl = torch.empty((N, H, W, Cp), dtype=dtype or act_dtype(), device=s.device)
    check(_abi.fn("hd_sigmoid_bwd_nchw_to_nhwc", dl)(ptr(dy.contiguous()), ptr(s), ptr(dl), N, Cr, H, W, Cp, gscale,
                                                  _stream()), "hd_sigmoid_bwd_nchw_to_nhwc")
    return dl


def relu_bwd(dy, z):
    dx = torch.empty_like(dy)
    check(_abi.fn("hd_relu_bwd", dy)(ptr(dy), ptr(z), ptr(dx), dy.numel(), _stream()), "hd_relu_bwd")
    return dx


def f32_to_f16(x, scale=1.0, dtype=None):
    """fp32 -> the storage type (a scaled copy when that is float32)."""
    y = torch.empty(x.shape, dtype=dtype or act_dtype(), device=x.device)
    check(_abi.fn("hd_f32_to_f16", y)(ptr(x.contiguous()), ptr(y), x.numel(), scale, _stream()), "hd_f32_to_f16")
    return y


def f16_to_f32(x, scale=1.0):
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(_abi.fn("hd_f16_to_f32", x)(ptr(x.contiguous()), ptr(y), x.numel(), scale, _stream()), "hd_f16_to_f32")
    return y


def channel_sum(x, rows=None):
    C_ = x.shape[-1]
    npix = x.numel() // C_
    if rows is None:
        rows = int(max(1, min(512, npix // 64)))
    part = torch.empty((rows, C_), dtype=torch.float32, device=x.device)
    check(_abi.fn("hd_channel_sum_f16", x)(ptr(x), npix, C_, ptr(part), rows, _stream()), "hd_channel_sum_f16")
    return colsum(part)


def scale_store(src, dst, scale=1.0, accumulate=False):
    check(_abi.load().hd_scale_store(ptr(src), ptr(dst), dst.numel(), scale, 1 if accumulate else 0, _stream()),
          "hd_scale_store")
    return dst


def nms_sorted_batched(boxes, counts, iou_thr, max_keep=None):
    """boxes [B, nmax, 4] fp32 sorted by descending score per image; counts [B] int32.  Returns keep [B, nmax] bool.
    `max_keep`: the caller only uses the first max_keep survivors per image -> the serial scan stops once they are found
    (entries after that 64-box chunk read False)."""
    _need_cuda(boxes, counts)
    B, nmax, _ = boxes.shape
    cb = (nmax + 63) // 64
    ws = torch.empty((B, nmax, cb), dtype=torch.int64, device=boxes.device)
    keep = torch.zeros((B, nmax), dtype=torch.uint8, device=boxes.device)
    check(_abi.load().hd_nms_sorted_batched_topk(ptr(boxes.contiguous()), ptr(counts), B, nmax, iou_thr, ptr(ws), ptr(keep),
                                                 0x7fffffff if max_keep is None else int(max_keep), _stream()), "hd_nms_sorted_batched")
    return keep.bool()


def sample_pos_neg(labels, keys, batch_size, cap_pos):
    """labels [N,A] int64, keys [N,A] int32 (random, >= 0) -> pos_sel, neg_sel [N,A] bool, counts [N,2] int64 (num_pos, num_neg)."""
    _need_cuda(labels, keys)
    N, A = labels.shape
    dev = labels.device
    pos = torch.empty((N, A), dtype=torch.bool, device=dev)
    neg = torch.empty((N, A), dtype=torch.bool, device=dev)
    counts = torch.empty((N, 2), dtype=torch.int64, device=dev)
    check(_abi.load().hd_sample_pos_neg(ptr(labels.contiguous()), ptr(keys.contiguous()), N, A, int(batch_size), int(cap_pos), ptr(pos), ptr(neg),
                                        ptr(counts), _stream()), "hd_sample_pos_neg")
    return pos, neg, counts


def roi_samples_finish(sel, comb, lab, matched, gt, gvalid, coder_weights):
    """sel [R] int64 (flat indices into the [N,T] candidates), comb [N,T,4] f32, lab / matched [N,T] int64, gt [N,G,4],
    gvalid [N,G] bool -> rois [R,5] f32, labels [R] int64, regression targets [R,4] f32 (one launch)."""
    _need_cuda(sel, comb, lab, matched, gt, gvalid)
    N, T, _ = comb.shape
    G = gt.shape[1]
    R = sel.shape[0]
    dev = comb.device
    rois = torch.empty((R, 5), dtype=torch.float32, device=dev)
    labels = torch.empty((R,), dtype=torch.int64, device=dev)
    reg_t = torch.empty((R, 4), dtype=torch.float32, device=dev)
    gv = gvalid.contiguous().view(torch.uint8) if gvalid.dtype == torch.bool else gvalid.to(torch.uint8).contiguous()
    w = (C.c_float * 4)(*[float(x) for x in coder_weights])
    check(_abi.load().hd_roi_samples_finish(ptr(sel.contiguous()), R, ptr(comb.contiguous().float()), ptr(lab.contiguous()), ptr(matched.contiguous()),
                                            ptr(gt.contiguous().float()), ptr(gv), T, G, C.cast(w, C.c_void_p), ptr(rois), ptr(labels), ptr(reg_t),
                                            _stream()), "hd_roi_samples_finish")
    return rois, labels, reg_t


def roi_samples_padded(pos_sel, neg_sel, comb, lab, matched, gt, gvalid, S, coder_weights):
    """Fixed-size RoI sampling tail in one launch: (rois [N*S,5], labels [N*S] (-1 = padding), targets [N*S,4], counts [N] int64)."""
    _need_cuda(pos_sel, neg_sel, comb, lab, matched, gt, gvalid)
    N, T, _ = comb.shape
    G = gt.shape[1]
    dev = comb.device
    rois = torch.empty((N * S, 5), dtype=torch.float32, device=dev)
    labels = torch.empty((N * S,), dtype=torch.int64, device=dev)
    reg_t = torch.empty((N * S, 4), dtype=torch.float32, device=dev)
    counts = torch.empty((N,), dtype=torch.int64, device=dev)
    u8 = lambda t: t.contiguous().view(torch.uint8) if t.dtype == torch.bool else t.to(torch.uint8).contiguous()
    w = (C.c_float * 4)(*[float(x) for x in coder_weights])
    check(_abi.load().hd_roi_samples_padded(ptr(u8(pos_sel)), ptr(u8(neg_sel)), ptr(comb.contiguous().float()), ptr(lab.contiguous()),
                                            ptr(matched.contiguous()), ptr(gt.contiguous().float()), ptr(u8(gvalid)), N, T, G, int(S),
                                            C.cast(w, C.c_void_p), ptr(rois), ptr(labels), ptr(reg_t), ptr(counts), _stream()),
          "hd_roi_samples_padded")
    return rois, labels, reg_t, counts


def roi_postprocess(class_logits, box_regression, rois, counts, S, coder_weights, bbox_xform_clip, image_shape, score_thresh, min_size=1e-2):
    """Fixed-size RoI list -> foreground boxes [R, C-1, 4], scores [R, C-1], valid [R, C-1] bool (softmax + decode + clip + tests)."""
    _need_cuda(class_logits, box_regression, rois, counts)
    R, C_ = class_logits.shape
    dev = class_logits.device
    lg, br, ro = class_logits.contiguous().float(), box_regression.contiguous().float(), rois.contiguous().float()
    boxes = torch.empty((R, C_ - 1, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((R, C_ - 1), dtype=torch.float32, device=dev)
    valid = torch.empty((R, C_ - 1), dtype=torch.bool, device=dev)
    w = (C.c_float * 4)(*[float(x) for x in coder_weights])
    check(_abi.load().hd_roi_postprocess(ptr(lg), ptr(br), ro.data_ptr() + (ro.shape[1] - 4) * 4, ro.shape[1], ptr(counts.contiguous().to(torch.int64)),
                                         R, C_, int(S), C.cast(w, C.c_void_p), float(bbox_xform_clip), float(image_shape[0]), float(image_shape[1]),
                                         float(score_thresh), float(min_size), ptr(boxes), ptr(scores), ptr(valid), _stream()), "hd_roi_postprocess")
    return boxes, scores, valid


def roi_levels(rois, canonical_scale, canonical_level, eps, k_min, k_max):
    """rois [R, 4 or 5] f32 (box in the last four columns) -> FPN level index [R] int32 (LevelMapper)."""
    _need_cuda(rois)
    rois = rois.contiguous().float()
    R = rois.shape[0]
    levels = torch.empty((R,), dtype=torch.int32, device=rois.device)
    off = rois.shape[1] - 4
    check(_abi.load().hd_roi_levels(rois.data_ptr() + off * 4 if R else None, rois.shape[1], R, float(canonical_scale), float(canonical_level),
                                    float(eps), int(k_min), int(k_max), ptr(levels), _stream()), "hd_roi_levels")
    return levels


def batched_nms_pick(boxes, idxs, valid, order, iou_thr, top_n):
    """Coordinate-offset batched NMS from unsorted candidates: boxes [B,n,4] f32, idxs [B,n] int64, valid [B,n] bool,
    order [B,n] int64 (descending score, invalid last).  -> pick [B, min(top_n, n)] int64 (candidate index of the k-th
    survivor in score order, order[b,0] beyond the count), picked [B] int64."""
    _need_cuda(boxes, idxs, valid, order)
    B, n, _ = boxes.shape
    dev = boxes.device
    cb = (n + 63) // 64
    boxes, idxs, order = boxes.contiguous().float(), idxs.contiguous(), order.contiguous()
    v8 = valid.contiguous().view(torch.uint8) if valid.dtype == torch.bool else valid.to(torch.uint8).contiguous()
    sorted_ws = torch.empty((B, n, 4), dtype=torch.float32, device=dev)
    counts_ws = torch.empty((B,), dtype=torch.int32, device=dev)
    mask_ws = torch.empty((B, n, cb), dtype=torch.int64, device=dev)
    keep_ws = torch.empty((B, n), dtype=torch.uint8, device=dev)
    k = min(int(top_n), n)
    pick = torch.empty((B, k), dtype=torch.int64, device=dev)
    picked = torch.empty((B,), dtype=torch.int64, device=dev)
    check(_abi.load().hd_batched_nms_pick(ptr(boxes), ptr(idxs), ptr(v8), ptr(order), B, n, float(iou_thr), int(top_n), ptr(sorted_ws),
                                          ptr(counts_ws), ptr(mask_ws), ptr(keep_ws), ptr(pick), ptr(picked), _stream()), "hd_batched_nms_pick")
    return pick, picked


def batched_nms_pick_segments(boxes, scores, valid, seg_sizes, iou_thr, top_n):
    """Coordinate-offset batched NMS whose categories are SEGMENTS of the candidate list (the RPN's feature levels), each already in
    descending score order: boxes [B,n,4] f32, scores [B,n] f32, valid [B,n] bool, seg_sizes (python ints, sum n, <= 8).
    -> pick [B, min(top_n, n)] int64, picked [B] int64 -- what `batched_nms_pick` returns for the same candidates."""
    _need_cuda(boxes, scores, valid)
    B, n, _ = boxes.shape
    dev = boxes.device
    seg_sizes = [int(v) for v in seg_sizes]
    L, S = len(seg_sizes), max(seg_sizes)
    assert sum(seg_sizes) == n and scores.dtype == torch.float32
    boxes, scores = boxes.contiguous().float(), scores.contiguous()
    v8 = valid.contiguous().view(torch.uint8) if valid.dtype == torch.bool else valid.to(torch.uint8).contiguous()
    cb = (S + 63) // 64
    sorted_ws = torch.empty((B * L, S, 4), dtype=torch.float32, device=dev)
    order_ws = torch.empty((B * L, S), dtype=torch.int64, device=dev)
    counts_ws = torch.empty((B * L,), dtype=torch.int32, device=dev)
    mask_ws = torch.empty((B * L, S, cb), dtype=torch.int64, device=dev)
    keep_ws = torch.empty((B * L, S), dtype=torch.uint8, device=dev)
    pick_ws = torch.empty((B * L, min(int(top_n), S)), dtype=torch.int64, device=dev)
    picked_seg = torch.empty((B, L), dtype=torch.int64, device=dev)
    pick = torch.empty((B, min(int(top_n), n)), dtype=torch.int64, device=dev)
    picked = torch.empty((B,), dtype=torch.int64, device=dev)
    segs = (C.c_int * L)(*seg_sizes)
    check(_abi.load().hd_batched_nms_pick_segments(ptr(boxes), ptr(scores), ptr(v8), B, n, C.cast(segs, C.c_void_p), L, float(iou_thr), int(top_n),
                                                   ptr(sorted_ws), ptr(order_ws), ptr(counts_ws), ptr(mask_ws), ptr(keep_ws), ptr(pick_ws), ptr(picked_seg),
                                                   ptr(pick), ptr(picked), _stream()), "hd_batched_nms_pick_segments")
    return pick, picked


def roi_align(feat, rois, PH, PW, spatial_scale, sampling_ratio):
    N, H, W, C_ = feat.shape
    R = rois.shape[0]
    out = torch.empty((R, PH, PW, C_), dtype=feat.dtype, device=feat.device)
    check(_abi.fn("hd_roi_align", feat)(ptr(feat), ptr(rois.contiguous()), ptr(out), R, N, H, W, C_, PH, PW, spatial_scale,
                                   sampling_ratio, _stream()), "hd_roi_align")
    return out


def roi_align_bwd(dout, rois, feat_shape, spatial_scale, sampling_ratio):
    N, H, W, C_ = feat_shape
    R, PH, PW, _ = dout.shape
    dfeat = torch.zeros((N, H, W, C_), dtype=torch.float32, device=dout.device)
    check(_abi.fn("hd_roi_align_bwd", dout)(ptr(dout.contiguous()), ptr(rois.contiguous()), ptr(dfeat), R, N, H, W, C_, PH, PW,
                                       spatial_scale, sampling_ratio, _stream()), "hd_roi_align_bwd")
    return dfeat


def roi_align_ml(feats, scales, rois, levels, PH, PW, sampling_ratio):
    """Multi-level RoIAlign.  feats: list of [N,H,W,C] f16; rois [R,5] fp32 (batch,x1,y1,x2,y2); levels [R] int32."""
    L = len(feats)
    _need_cuda(rois, levels, *feats)
    R = rois.shape[0]
    C_ = feats[0].shape[3]
    out = torch.empty((R, PH, PW, C_), dtype=feats[0].dtype, device=rois.device)
    fp = (C.c_void_p * L)(*[f.data_ptr() for f in feats])
    Hs = (C.c_int * L)(*[f.shape[1] for f in feats])
    Ws = (C.c_int * L)(*[f.shape[2] for f in feats])
    sc = (C.c_float * L)(*scales)
    check(_abi.fn("hd_roi_align_ml", feats[0])(fp, Hs, Ws, sc, L, ptr(rois.contiguous()), ptr(levels), ptr(out), R, C_, PH, PW,
                                      sampling_ratio, _stream()), "hd_roi_align_ml")
    return out


def roi_align_ml_bwd(dout, rois, levels, feat_shapes, scales, sampling_ratio):
    """Returns list of fp32 gradient maps (zero-initialised, atomically accumulated)."""
    L = len(feat_shapes)
    R, PH, PW, C_ = dout.shape
    dfs = [torch.zeros(s, dtype=torch.float32, device=dout.device) for s in feat_shapes]
    fp = (C.c_void_p * L)(*[f.data_ptr() for f in dfs])
    Hs = (C.c_int * L)(*[s[1] for s in feat_shapes])
    Ws = (C.c_int * L)(*[s[2] for s in feat_shapes])
    sc = (C.c_float * L)(*scales)
    check(_abi.fn("hd_roi_align_ml_bwd", dout)(ptr(dout.contiguous()), ptr(rois.contiguous()), ptr(levels), fp, Hs, Ws, sc, L, R, C_,
                                          PH, PW, sampling_ratio, _stream()), "hd_roi_align_ml_bwd")
    return dfs


def roi_align_ml_bwd_gather(dout, rois, levels, feat_shapes, scales, sampling_ratio, n_images=None):
    """Gather-form backward (7x7 bins, sampling_ratio 2): returns fp16 gradient maps shaped like the features; images
    >= n_images (no RoI refers to them) stay zero.  No atomics: deterministic, every element written once."""
    L = len(feat_shapes)
    R, PH, PW, C_ = dout.shape
    N = feat_shapes[0][0]
    n_images = N if n_images is None else min(int(n_images), N)
    alloc = torch.empty if n_images == N else torch.zeros
    dfs = [alloc(s, dtype=dout.dtype, device=dout.device) for s in feat_shapes]
    fp = (C.c_void_p * L)(*[f.data_ptr() for f in dfs])
    Hs = (C.c_int * L)(*[s[1] for s in feat_shapes])
    Ws = (C.c_int * L)(*[s[2] for s in feat_shapes])
    sc = (C.c_float * L)(*scales)
    check(_abi.fn("hd_roi_align_ml_bwd_gather", dout)(ptr(dout.contiguous()), ptr(rois.contiguous()), ptr(levels), fp, Hs, Ws, sc, L, R,
                                                 n_images, C_, PH, PW, sampling_ratio, _stream()), "hd_roi_align_ml_bwd_gather")
    return dfs


def box_iou(gt, boxes):
    G, A = gt.shape[0], boxes.shape[0]
    iou = torch.empty((G, A), dtype=torch.float32, device=boxes.device)
    check(_abi.load().hd_box_iou(ptr(gt.contiguous()), G, ptr(boxes.contiguous()), A, ptr(iou), _stream()), "hd_box_iou")
    return iou


def box_iou_batched(gt, boxes):
    """gt [N,G,4]; boxes [A,4] (shared) or [N,A,4].  Returns [N,G,A] fp32."""
    N, G, _ = gt.shape
    shared = boxes.dim() == 2
    A = boxes.shape[-2]
    iou = torch.empty((N, G, A), dtype=torch.float32, device=gt.device)
    check(_abi.load().hd_box_iou_batched(ptr(gt.contiguous()), G, ptr(boxes.contiguous()), A, N, 1 if shared else 0, ptr(iou),
                                         _stream()), "hd_box_iou_batched")
    return iou


def rpn_decode_filter(deltas, objectness, anchors, top, bbox_xform_clip, image_shape, min_size, score_thresh):
    """deltas [N,A,4] f32, objectness [N,A] f32, anchors [A,4] f32, top [N,K] int64 -> (boxes [N,K,4], prob [N,K], valid [N,K] bool):
    decode + clip + sigmoid + min-size / score tests of the selected anchors in one launch."""
    _need_cuda(deltas, objectness, anchors, top)
    N, A = objectness.shape
    K = top.shape[1]
    dev = deltas.device
    deltas, objectness, anchors, top = deltas.contiguous().float(), objectness.contiguous().float(), anchors.contiguous().float(), top.contiguous()
    boxes = torch.empty((N, K, 4), dtype=torch.float32, device=dev)
    prob = torch.empty((N, K), dtype=torch.float32, device=dev)
    valid = torch.empty((N, K), dtype=torch.bool, device=dev)
    check(_abi.load().hd_rpn_decode_filter(ptr(deltas), ptr(objectness), ptr(anchors), ptr(top), N, A, K, float(bbox_xform_clip),
                                           float(image_shape[0]), float(image_shape[1]), float(min_size), float(score_thresh), ptr(boxes),
                                           ptr(prob), ptr(valid), _stream()), "hd_rpn_decode_filter")
    return boxes, prob, valid


def roi_decode_clip(codes, rois, coder_weights, bbox_xform_clip, image_shape):
    """codes [R, K*4] f32, rois [R, 4 or 5] f32 (box in the last four columns) -> boxes [R, K, 4] decoded and clipped."""
    _need_cuda(codes, rois)
    R = codes.shape[0]
    K = codes.shape[1] // 4
    codes, rois = codes.contiguous().float(), rois.contiguous().float()
    boxes = torch.empty((R, K, 4), dtype=torch.float32, device=codes.device)
    w = (C.c_float * 4)(*[float(x) for x in coder_weights])
    off = rois.shape[1] - 4
    check(_abi.load().hd_roi_decode_clip(ptr(codes), rois.data_ptr() + off * 4, rois.shape[1], R, K, C.cast(w, C.c_void_p), float(bbox_xform_clip),
                                         float(image_shape[0]), float(image_shape[1]), ptr(boxes), _stream()), "hd_roi_decode_clip")
    return boxes


def topk_rows_segments(scores, seg_sizes, k):
    """Per row of scores [B, sum(seg_sizes)] and per segment: the indices (into the row) of the min(k, n) largest entries
    of the segment in descending score order, equal scores by ascending index -- torch.sort(seg, descending=True,
    stable=True)[1][:, :k] + offset for every segment, concatenated.  -> int64 [B, sum(min(k, n))]."""
    _need_cuda(scores)
    assert scores.dim() == 2 and scores.dtype == torch.float32 and scores.stride(1) == 1
    B = scores.shape[0]
    seg_sizes = [int(n) for n in seg_sizes]
    assert sum(seg_sizes) == scores.shape[1]
    out = torch.empty((B, sum(min(k, n) for n in seg_sizes)), dtype=torch.int64, device=scores.device)
    segs = (C.c_int * len(seg_sizes))(*seg_sizes)
    check(_abi.load().hd_topk_select_rows(ptr(scores), B, scores.stride(0), C.cast(segs, C.c_void_p), len(seg_sizes), int(k), ptr(out),
                                          out.stride(0), _stream()), "hd_topk_select_rows")
    return out


def match_targets(gt, gvalid, glabels, boxes, high, low, allow_low_quality, coder_weights=None, want_labels=True):
    """Fused box_iou + Matcher + label lookup (+ BoxCoder.encode when `coder_weights` is given) for N images.
    gt [N,G,4] f32, gvalid [N,G] bool, glabels [N,G] int64 or None, boxes [A,4] (shared) or [N,A,4].
    Returns (matched [N,A] int64 with -1/-2 codes, labels [N,A] int64 or None, reg_t [N,A,4] f32 or None)."""
    _need_cuda(gt, gvalid, boxes, glabels)
    N, G, _ = gt.shape
    shared = boxes.dim() == 2
    A = boxes.shape[-2]
    dev = gt.device
    gt, boxes = gt.contiguous().float(), boxes.contiguous().float()
    gv = gvalid.contiguous().view(torch.uint8) if gvalid.dtype == torch.bool else gvalid.to(torch.uint8).contiguous()
    gl = None if glabels is None else glabels.contiguous().to(torch.int64)
    matched = torch.empty((N, A), dtype=torch.int64, device=dev)
    labels = torch.empty((N, A), dtype=torch.int64, device=dev) if want_labels else None
    reg_t = w = None
    if coder_weights is not None:
        reg_t = torch.empty((N, A, 4), dtype=torch.float32, device=dev)
        w = (C.c_float * 4)(*[float(x) for x in coder_weights])
    ws = torch.empty((N * G,), dtype=torch.float32, device=dev) if allow_low_quality else None
    check(_abi.load().hd_match_targets(ptr(gt), ptr(gv), ptr(gl), G, ptr(boxes), A, N, 1 if shared else 0, float(high), float(low),
                                       1 if allow_low_quality else 0, C.cast(w, C.c_void_p) if w is not None else None, ptr(ws), ptr(matched),
                                       ptr(labels), ptr(reg_t), _stream()), "hd_match_targets")
    return matched, labels, reg_t


def adam_step(p, g, m, v, *, lr, beta1, beta2, eps, weight_decay, clip_value, inv_scale, step, found_inf=None):
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    check(_abi.load().hd_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, weight_decay,
                                   clip_value, inv_scale, bc1, bc2, ptr(found_inf), _stream()), "hd_adam_step")


def check_finite(g, found_inf):
    check(_abi.load().hd_check_finite(ptr(g), g.numel(), ptr(found_inf), _stream()), "hd_check_finite")
