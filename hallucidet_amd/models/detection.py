"""Faster R-CNN ResNet-50-FPN on hand-written HIP kernels, with torchvision 0.12's attribute tree.

The reference takes this model from the un-vendored torchvision (src/models/detector.py:130) and reaches into it from
src/utils/eval_forward_fasterrcnn.py:38-136: `.transform`, `.backbone(x) -> OrderedDict('0','1','2','3','pool')`,
`.rpn.{head, anchor_generator, box_coder, filter_proposals, assign_targets_to_anchors, compute_loss}`,
`.roi_heads.{select_training_samples, box_roi_pool, box_head, box_predictor, postprocess_detections, has_keypoint,
keypoint_*}`.  This module provides exactly that surface and the same state_dict keys (`backbone.body.layer1.0.conv1
.weight`, `backbone.fpn.inner_blocks.0.weight`, `rpn.head.conv.weight`, `roi_heads.box_head.fc6.weight`, ...), so
torchvision-0.12 checkpoints load unchanged (0.13+ FPN/RPN key renames are mapped in `load_state_dict`).

Execution: the detector is FROZEN on the hot path (train_hallucidet.py:100-105).  FrozenBatchNorm is folded
into the fp16 GEMM-layout weights once; forward = implicit-GEMM convs with bias/residual/ReLU epilogues; backward = data
gradients only (the same kernel on flipped weights, ReLU masks fused in the epilogue).  Box arithmetic, matching,
sampling and the four losses are small fp32 tensor ops on the GPU; NMS, RoIAlign and IoU are HIP kernels.
Sampling uses `torch.randperm` in the reference's call order (injectable for parity tests).
"""
import math
import os
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..ops import ACT_NONE, ACT_RELU
from .custom_generalized_transform import CustomGeneralizedRCNNTransform


# ======================================================================================================================
# parameter containers
# ======================================================================================================================
class FrozenBatchNorm2d(nn.Module):
    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))

    def scale_shift(self):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        return scale, self.bias - self.running_mean * scale


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, width, stride):
        super().__init__()
        cout = width * 4
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = FrozenBatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, bias=False)
        self.bn2 = FrozenBatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, cout, 1, bias=False)
        self.bn3 = FrozenBatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), FrozenBatchNorm2d(cout))
        self.stride = stride


class ResNet50Body(nn.Module):
    """IntermediateLayerGetter(resnet50, layer1..4 -> '0'..'3')"""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = FrozenBatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for i, (n, wdt) in enumerate(zip((3, 4, 6, 3), (64, 128, 256, 512))):
            blocks = []
            for b in range(n):
                blocks.append(Bottleneck(cin, wdt, 2 if (b == 0 and i > 0) else 1))
                cin = wdt * 4
            setattr(self, "layer%d" % (i + 1), nn.Sequential(*blocks))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


class LastLevelMaxPool(nn.Module):
    pass


class LastLevelP6P7(nn.Module):
    """RetinaNet's two extra levels [EXT torchvision 0.12 ops/feature_pyramid_network.py]: P6 = conv3x3/s2(P5) when
    in_channels == out_channels (as retinanet_resnet50_fpn builds it), P7 = conv3x3/s2(ReLU(P6))."""

    def __init__(self, in_channels=256, out_channels=256):
        super().__init__()
        self.p6 = nn.Conv2d(in_channels, out_channels, 3, 2, 1)
        self.p7 = nn.Conv2d(out_channels, out_channels, 3, 2, 1)
        self.use_P5 = in_channels == out_channels


class FeaturePyramidNetwork(nn.Module):
    def __init__(self, in_channels_list=(256, 512, 1024, 2048), out_channels=256, extra_blocks=None):
        super().__init__()
        self.inner_blocks = nn.ModuleList(nn.Conv2d(c, out_channels, 1) for c in in_channels_list)
        self.layer_blocks = nn.ModuleList(nn.Conv2d(out_channels, out_channels, 3, padding=1) for _ in in_channels_list)
        self.extra_blocks = LastLevelMaxPool() if extra_blocks is None else extra_blocks
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1)
                nn.init.constant_(m.bias, 0)


_STEM_SUBPIXEL = os.environ.get("HD_STEM_SUBPIXEL", "1") != "0"      # 0: the stem's data gradient through hd_conv2d's in_dil = 2 route (A/B)


def _conv_entry(conv, bn=None, cin_pad=None):
    """Fold FrozenBN into the conv and build the fp16 GEMM layouts (+ data-gradient layout)."""
    w = conv.weight.detach().float()
    cout, cin = w.shape[:2]
    cin_p = cin_pad or (cin + 7) // 8 * 8
    cout_p = (cout + 7) // 8 * 8
    if bn is not None:
        scale, shift = bn.scale_shift()
        scale, bias = scale.float().contiguous(), shift.float().contiguous()
    else:
        scale = None
        bias = conv.bias.detach().float().contiguous() if conv.bias is not None else None
    wf, wd = ops.weight_prep(w, out_scale=scale, cin_pad=cin_p, cout_pad=cout_p, want_fwd=True, want_dgrad=True)
    return dict(wf=wf, wd=wd, bias=bias, k=conv.kernel_size[0], stride=conv.stride[0], pad=conv.padding[0], cin=cin, cout=cout,
                cin_p=cin_p, cout_p=cout_p)


def _fwd(e, x, *, act=ACT_NONE, res=None, f32=False):
    """f32: False -> NHWC fp16; True -> NCHW fp32; "nhwc" -> NHWC fp32 (returned as its NCHW VIEW: torchvision's
    `permute_and_flatten` of such a tensor is a free view instead of a copy)."""
    if f32 == "nhwc":
        return ops.conv2d(x, e["wf"], e["k"], e["k"], bias=e["bias"], res=res, stride=e["stride"], pad=e["pad"], act=act,
                          out_nhwc_f32=True, cout=e["cout"]).permute(0, 3, 1, 2)
    return ops.conv2d(x, e["wf"], e["k"], e["k"], bias=e["bias"], res=res, stride=e["stride"], pad=e["pad"], act=act,
                      out_nchw_f32=f32, cout=e["cout"])


def _fwd_many(es, xs, *, act=ACT_NONE, f32=False):
    """`_fwd(e, x, ...)` for several independent (entry, input) pairs -- the same layer type on every feature level -- as ONE grid
    where the kernels allow it (ops.conv2d_multi; put the largest level first).  Same outputs as the per-level calls, bit for bit."""
    calls = []
    for e, x in zip(es, xs):
        kw = dict(bias=e["bias"], stride=e["stride"], pad=e["pad"], act=act, cout=e["cout"])
        if f32 == "nhwc":
            kw["out_nhwc_f32"] = True
        elif f32:
            kw["out_nchw_f32"] = True
        calls.append((x, e["wf"], e["k"], e["k"], kw))
    outs = ops.conv2d_multi(calls)
    if f32 == "nhwc":
        return [o.permute(0, 3, 1, 2) for o in outs]
    return outs


def _dgrad_many(es, dys, hws, *, ress=None, masks=None):
    """`_dgrad` for several independent (entry, gradient) pairs as one grid (see _fwd_many)."""
    calls = []
    for i, (e, dy, hw) in enumerate(zip(es, dys, hws)):
        calls.append((dy, e["wd"], e["k"], e["k"], dict(stride=1, pad=e["k"] - 1 - e["pad"], in_dil=e["stride"], out_hw=hw, cout=e["cin_p"],
                                                        res=None if ress is None else ress[i], mask=None if masks is None else masks[i])))
    return ops.conv2d_multi(calls)


def _head_grad_nhwc16(g, H, W, cout_p, dtype=None):
    """Gradient of a head output [n, C, H, W] fp32 -> NHWC fp16 with cout_p channels.  When the gradient is the NCHW view of NHWC
    memory (what autograd hands back for `_fwd(..., f32="nhwc")` outputs) this is one pad-and-cast launch."""
    v = g.permute(0, 2, 3, 1)
    if g.dtype == torch.float32 and v.stride()[1:] == (v.shape[2] * v.shape[3], v.shape[3], 1):     # dense images (slices of the flat [N, HWA, C] gradient)
        return ops.pad_cast_f32_f16(v, cout_p, dtype=dtype)
    return ops.nchw_to_nhwc_resize(g.contiguous().float(), H, W, cout_p, dtype=dtype)


_PAD_CAST_MULTI = os.environ.get("HD_PAD_CAST_MULTI", "1") != "0"      # A/B knob: 0 = one launch per tensor


def _head_grads_nhwc16_many(gs, hws, cout_ps, dtype=None):
    """`_head_grad_nhwc16` for a list of head-output gradients: the dense ones (the training step's: slices of the flat [N, HWA, C]
    gradient) leave in ONE pad-and-cast launch (ops.pad_cast_f32_f16_many), anything else one by one."""
    outs = [None] * len(gs)
    dense, where = [], []
    for i, (g, (H, W), cp) in enumerate(zip(gs, hws, cout_ps)):
        v = g.permute(0, 2, 3, 1)
        if _PAD_CAST_MULTI and g.dtype == torch.float32 and v.stride()[1:] == (v.shape[2] * v.shape[3], v.shape[3], 1):
            dense.append(v)
            where.append(i)
        else:
            outs[i] = _head_grad_nhwc16(g, H, W, cp, dtype)
    if dense:
        for i, y in zip(where, ops.pad_cast_f32_f16_many(dense, [cout_ps[i] for i in where], dtype=dtype)):
            outs[i] = y
    return outs


def _dgrad(e, dy, in_hw, *, res=None, mask=None):
    return ops.conv2d(dy, e["wd"], e["k"], e["k"], stride=1, pad=e["k"] - 1 - e["pad"], in_dil=e["stride"], out_hw=in_hw,
                      cout=e["cin_p"], res=res, mask=mask)


# ---- parameter gradients (detector fine-tuning, train_detector.py:147-203 / `train_det=True`) ---------------------------
# Modules carry `train_params` (bool) and `grad_scale` (loss scale to remove); parameter gradients are ACCUMULATED into
# `param.grad`, which `ParamArena` (optim.py) makes a view of one flat fp32 arena that is zeroed once per step.
def _wgrad_into(param, e, x, dy, inv_scale, bn_scale=None):
    """param.grad += inv_scale * dL/dW for conv entry `e` (x: NHWC f16 input, dy: NHWC f16 gradient of the conv output
    before bias/activation).  FrozenBN-folded convs: W_eff = W * s[cout]  =>  dL/dW = s[cout] * dL/dW_eff."""
    k = e["k"]
    slab = ops.wgrad(x, dy, k, k, stride=e["stride"], pad=e["pad"])
    g = param.grad
    if bn_scale is None:
        ops.wgrad_reduce(slab, g, k, k, Cin=e["cin_p"], Cin_real=e["cin"], Cout=e["cout"], scale=inv_scale, accumulate=True)
    else:
        tmp = torch.empty_like(g)
        ops.wgrad_reduce(slab, tmp, k, k, Cin=e["cin_p"], Cin_real=e["cin"], Cout=e["cout"], scale=inv_scale, accumulate=False)
        g.addcmul_(tmp, bn_scale.view(-1, 1, 1, 1))


def _bgrad_into(param, dy, inv_scale):
    c8 = dy.shape[-1] // 8
    if dy.shape[-1] % 8 == 0 and c8 & (c8 - 1) == 0:
        s = ops.channel_sum(dy)
    else:                       # 24- / 40-channel head outputs (RetinaNet cls / box convs): hd_channel_sum_f16 wants C/8 = 2^k
        s = dy.float().sum(dim=(0, 1, 2))
    param.grad.add_(s[: param.numel()], alpha=inv_scale)


_ACTIVE_VIEWS = os.environ.get("HD_ACTIVE_VIEWS", "1") != "0"
_DEFER_DETS = os.environ.get("HD_DEFER_DETS", "1") != "0"      # A/B knob of LazyDetections.deferred


def _active_views(feats, n_active):
    """When only the first `n_active` images of a batched multi-pass evaluation carry a gradient, the backbone hands out, next
    to every full feature map, its leading [n_active] slice as the tensor autograd differentiates (`_hd_active`): the consumers
    compute on the full map and return [n_active, ...] gradients for the slice -- no zero-padded 24-image gradient maps, no copies
    into them, and autograd's accumulation of the branches adds a third of the bytes."""
    acts = [getattr(f, "_hd_active", None) for f in feats]
    if all(a is not None and a.shape[0] == n_active and a.requires_grad for a in acts):
        return acts
    return None


class _BackboneFn(torch.autograd.Function):
    """`n_active`: only the first n_active images of the batch need a data gradient (the hallucinated images when the
    RGB / IR detector passes of a training step are batched with them); the rest is forward-only."""

    @staticmethod
    def forward(ctx, x, hook, bb, n_active):
        outs, saved = bb._forward(x, save=True, n_active=n_active)
        ctx.bb, ctx.saved, ctx.n_active, ctx.n = bb, saved, n_active, x.shape[0]
        ctx.need_dx = x.requires_grad
        ctx.xmeta = (tuple(x.shape), x.dtype, x.device)
        ctx.nout = len(outs)
        ctx.set_materialize_grads(False)          # unused levels arrive as None, not as zero-filled maps
        if n_active < x.shape[0] and _ACTIVE_VIEWS:
            ctx.mark_non_differentiable(*outs)
            return tuple(outs) + tuple(o[:n_active] for o in outs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        na = ctx.n_active
        if len(grads) > ctx.nout:                      # gradients arrive for the [n_active] views
            grads = list(grads[ctx.nout:])
        else:
            grads = [None if g is None else g[:na] for g in grads]
        # The gradient of the whole [n] batch with rows >= n_active zero: a buffer kept per (shape, n_active) and zeroed ONCE -- the
        # stem's data gradient writes rows [0, n_active) in place every pass, nothing ever writes the others (was: a zero fill of the whole
        # batch + a copy per pass).  Created outside stream capture only (a graph's warm-up passes do that).
        full = None
        if na < ctx.n and ctx.need_dx:
            shape, dtype, dev = ctx.xmeta
            pads = ctx.bb.__dict__.setdefault("_dx_pads", {})
            key = (shape, dtype, str(dev), na)
            full = pads.get(key)
            if full is None and dtype == torch.float16 and not torch.cuda.is_current_stream_capturing():
                if len(pads) >= 4:
                    pads.pop(next(iter(pads)))
                full = pads[key] = torch.zeros(shape, dtype=dtype, device=dev)
        dx = ctx.bb._backward(ctx.saved, grads, need_dx=ctx.need_dx, dx_out=None if full is None else full[:na])
        ctx.saved = None
        if dx is None:
            return None, None, None, None
        if na < ctx.n:
            if full is not None and dx.data_ptr() == full.data_ptr():
                return full, None, None, None
            pad = torch.zeros((ctx.n,) + tuple(dx.shape[1:]), dtype=dx.dtype, device=dx.device)
            pad[:na] = dx
            dx = pad
        return dx, None, None, None


class BackboneWithFPN(nn.Module):
    out_channels = 256

    def __init__(self, returned_layers=(1, 2, 3, 4), extra_blocks=None):
        """Faster R-CNN: layers 1-4 + LastLevelMaxPool ('0','1','2','3','pool'); RetinaNet: returned_layers=(2,3,4) +
        LastLevelP6P7 ('0','1','2','p6','p7')."""
        super().__init__()
        self.body = ResNet50Body()
        self.returned_layers = tuple(returned_layers)
        assert self.returned_layers[-1] == 4 and list(self.returned_layers) == sorted(self.returned_layers)
        chans = [256 * 2 ** (l - 1) for l in self.returned_layers]
        self.fpn = FeaturePyramidNetwork(chans, 256, extra_blocks)
        self.p6p7 = isinstance(self.fpn.extra_blocks, LastLevelP6P7)
        self.out_names = tuple(str(i) for i in range(len(chans))) + (("p6", "p7") if self.p6p7 else ("pool",))
        self._pack = None
        self._hook = None
        self.train_params, self.grad_scale = False, 1.0     # set by FasterRCNN.set_trainable (detector fine-tuning)

    # -------------------------------------------------------------- frozen weight pack
    def invalidate(self):
        self._pack = None

    def pack(self):
        if self._pack is not None:
            return self._pack
        b = self.body
        P = {"stem": _conv_entry(b.conv1, b.bn1, cin_pad=8), "blocks": [], "inner": [], "layer": []}
        for li in range(1, 5):
            stage = []
            for blk in getattr(b, "layer%d" % li):
                stage.append(dict(c1=_conv_entry(blk.conv1, blk.bn1), c2=_conv_entry(blk.conv2, blk.bn2), c3=_conv_entry(blk.conv3, blk.bn3),
                                  ds=_conv_entry(blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None, blk=blk))
            P["blocks"].append(stage)
        for c in self.fpn.inner_blocks:
            P["inner"].append(_conv_entry(c))
        for c in self.fpn.layer_blocks:
            P["layer"].append(_conv_entry(c))
        if self.p6p7:
            P["p6"], P["p7"] = _conv_entry(self.fpn.extra_blocks.p6), _conv_entry(self.fpn.extra_blocks.p7)
        self._pack = P
        return P

    # -------------------------------------------------------------- execution
    def _forward(self, x, save, n_active=None):
        P = self.pack()
        na = x.shape[0] if n_active is None else n_active
        rec = {"x": x[:na], "blocks": []} if save else None
        s = _fwd(P["stem"], x, act=ACT_RELU)
        p, pidx = ops.maxpool3x3s2_idx(s) if save else (ops.maxpool3x3s2(s), None)
        cur = p
        C = []
        for si, stage in enumerate(P["blocks"]):
            srec = []
            for bi, e in enumerate(stage):
                o1 = _fwd(e["c1"], cur, act=ACT_RELU)
                o2 = _fwd(e["c2"], o1, act=ACT_RELU)
                idt = cur if e["ds"] is None else _fwd(e["ds"], cur)
                out = _fwd(e["c3"], o2, act=ACT_RELU, res=idt)
                if save:
                    srec.append((cur[:na], o1[:na], o2[:na], out[:na]))
                cur = out
            C.append(cur)
            if save:
                rec["blocks"].append(srec)
        L = len(self.returned_layers)
        Cr = [C[l - 1] for l in self.returned_layers]
        # the lateral 1x1 convs of all levels as one grid, the top-down additions, then the 3x3 output convs of all levels as one
        # grid (the 19x19 / 10x10 levels are 16-150 tiles each: they ride in the tail of the largest level's launch)
        inner = _fwd_many(P["inner"], Cr)
        for i in range(L - 2, -1, -1):
            inner[i] = ops.upsample_add(inner[i], inner[i + 1])
        outs = _fwd_many(P["layer"], inner)
        if self.p6p7:
            p6 = _fwd(P["p6"], outs[L - 1])
            extra = [p6, _fwd(P["p7"], torch.relu(p6))]
        else:
            extra = [ops.subsample2(outs[L - 1])]
        if save:
            rec["inner"] = [t[:na] for t in inner] if self.train_params else None
            rec["outs_top"] = outs[L - 1][:na] if (self.train_params and self.p6p7) else None
            rec.update(stem=s[:na], pool_idx=pidx[:na], pooled=p[:na], C=[c[:na] for c in C], out_shapes=[(na,) + tuple(t.shape[1:]) for t in outs],
                       p6=extra[0][:na] if self.p6p7 else None)
        return outs + extra, rec

    def _backward(self, rec, grads, need_dx=True, dx_out=None):
        P = self.pack()
        tp = self.train_params
        inv = 1.0 / self.grad_scale
        shapes = rec["out_shapes"]
        L = len(self.returned_layers)
        dev = rec["x"].device
        dP = []
        for i in range(L):
            g = grads[i]
            dP.append(torch.zeros(shapes[i], dtype=rec["x"].dtype, device=dev) if g is None else g.contiguous())
        top_hw = (shapes[L - 1][1], shapes[L - 1][2])
        if self.p6p7:
            g6, g7, p6 = grads[L], grads[L + 1], rec["p6"]
            eb = self.fpn.extra_blocks
            if g7 is not None:
                g7 = g7.contiguous()
                if tp:
                    _wgrad_into(eb.p7.weight, P["p7"], torch.relu(p6), g7, inv)
                    _bgrad_into(eb.p7.bias, g7, inv)
                d6 = _dgrad(P["p7"], g7, (p6.shape[1], p6.shape[2]), mask=p6)     # through ReLU(P6)
                g6 = d6 if g6 is None else ops.add_f16(g6.contiguous(), d6)
            if g6 is not None:
                g6 = g6.contiguous()
                if tp:
                    _wgrad_into(eb.p6.weight, P["p6"], rec["outs_top"], g6, inv)
                    _bgrad_into(eb.p6.bias, g6, inv)
                dP[L - 1] = _dgrad(P["p6"], g6, top_hw, res=dP[L - 1])
        elif grads[L] is not None:
            if grads[L - 1] is not None:
                dP[L - 1] = dP[L - 1].clone()
            ops.subsample2_bwd(grads[L].contiguous(), dP[L - 1], accumulate=True)
        C = rec["C"]
        Cr = [C[l - 1] for l in self.returned_layers]
        d_li = _dgrad_many(P["layer"], dP, [(shapes[i][1], shapes[i][2]) for i in range(L)])
        for i in range(1, L):
            ops.upsample_add_bwd(d_li[i - 1], d_li[i], accumulate=True)
        if tp:
            for i in range(L):
                lb, ib = self.fpn.layer_blocks[i], self.fpn.inner_blocks[i]
                _wgrad_into(lb.weight, P["layer"][i], rec["inner"][i], dP[i], inv)
                _bgrad_into(lb.bias, dP[i], inv)
                _wgrad_into(ib.weight, P["inner"][i], Cr[i], d_li[i], inv)
                _bgrad_into(ib.bias, d_li[i], inv)
        # lateral 1x1 convs -> gradients w.r.t. the returned C levels (C5's carries the ReLU mask of layer4's output)
        dC = [None] * 4
        dlat = _dgrad_many(P["inner"], d_li, [(Cr[i].shape[1], Cr[i].shape[2]) for i in range(L)], masks=[Cr[i] if i == L - 1 else None for i in range(L)])
        for i, l in enumerate(self.returned_layers):
            dC[l - 1] = dlat[i]
        gm = dC[3]
        for si in (3, 2, 1, 0):
            stage, srec = P["blocks"][si], rec["blocks"][si]
            for bi in range(len(stage) - 1, -1, -1):
                e = stage[bi]
                x, o1, o2, out = srec[bi]
                extra = dC[si - 1] if (bi == 0 and si > 0) else None
                hw_o = (o2.shape[1], o2.shape[2])
                d2 = _dgrad(e["c3"], gm, hw_o, mask=o2)
                d1 = _dgrad(e["c2"], d2, (o1.shape[1], o1.shape[2]), mask=o1)
                hw_x = (x.shape[1], x.shape[2])
                if tp and si >= 1:                                     # layer2..4 are trainable (trainable_backbone_layers=3)
                    blk = e["blk"]
                    _wgrad_into(blk.conv3.weight, e["c3"], o2, gm, inv, blk.bn3.scale_shift()[0])
                    _wgrad_into(blk.conv2.weight, e["c2"], o1, d2, inv, blk.bn2.scale_shift()[0])
                    _wgrad_into(blk.conv1.weight, e["c1"], x, d1, inv, blk.bn1.scale_shift()[0])
                    if e["ds"] is not None:
                        _wgrad_into(blk.downsample[0].weight, e["ds"], x, gm, inv, blk.downsample[1].scale_shift()[0])
                if not need_dx and si == 1 and bi == 0:
                    return None                                        # nothing below layer2 is trainable and the input is data
                # the block input is a post-ReLU tensor except for layer1.0 (max-pooled stem)
                xmask = None if (si == 0 and bi == 0) else x
                if e["ds"] is not None:
                    t = _dgrad(e["ds"], gm, hw_x, res=extra)
                else:
                    t = gm if extra is None else ops.add_f16(gm, extra)
                gm = _dgrad(e["c1"], d1, hw_x, res=t, mask=xmask)
        ds_ = ops.maxpool3x3s2_bwd_idx(rec["pool_idx"], gm, (rec["stem"].shape[1], rec["stem"].shape[2]))
        x = rec["x"]
        e = P["stem"]
        if _STEM_SUBPIXEL and ds_.dtype == torch.float16 and e["k"] == 7 and e["stride"] == 2 and e["pad"] == 3 and e["cout"] == 64 and e["cin"] <= 4 and e["cin_p"] == 8:
            # the gradient handed to the hallucination network: sub-pixel form of the stem's data gradient, its ReLU backward in the staging
            if "wsub" not in e:
                e["wsub"] = ops.stem_dgrad_weights(e["wf"], e["cin_p"])
            if dx_out is not None and (tuple(dx_out.shape) != (ds_.shape[0], x.shape[1], x.shape[2], 8) or dx_out.dtype != torch.float16):
                dx_out = None
            return ops.conv7x7s2_dgrad_thin(ds_, e["wsub"], (x.shape[1], x.shape[2]), mask_z=rec["stem"], out=dx_out)
        ds_ = ops.relu_bwd(ds_, rec["stem"])
        return _dgrad(P["stem"], ds_, (x.shape[1], x.shape[2]))

    def forward(self, x, n_active=None):
        if not (isinstance(x, torch.Tensor) and x.dim() == 4 and x.dtype in (torch.float16, torch.float32) and x.shape[-1] == 8):
            raise TypeError("hallucidet_amd backbone expects ImageList.tensors from CustomGeneralizedRCNNTransform "
                            "(NHWC float16 -- float32 with precision=32 --, 8 channels)")
        if not x.is_cuda:
            raise RuntimeError("hallucidet_amd backbone runs on the GPU only; there is no CPU path")
        if torch.is_grad_enabled() and (x.requires_grad or self.train_params):
            if self._hook is None or self._hook.device != x.device:
                self._hook = torch.zeros(1, device=x.device, requires_grad=True)
            outs = _BackboneFn.apply(x, self._hook, self, x.shape[0] if n_active is None else n_active)
            n = len(self.out_names)
            if len(outs) > n:
                for full, act in zip(outs[:n], outs[n:]):
                    full._hd_active = act
                outs = outs[:n]
        else:
            outs, _ = self._forward(x, save=False)
        return OrderedDict(zip(self.out_names, outs))

    @torch.no_grad()
    def calibrate_(self, x_nhwc8):
        """Synthetic-weights helper: set every FrozenBN's running statistics to the batch statistics observed on
        `x_nhwc8` (what a trained checkpoint looks like), layer by layer, using the conv kernel's statistics epilogue.
        Without it a randomly initialised, un-normalised 50-layer residual stack overflows fp16."""
        b = self.body

        def fit(conv, bn, x, res=None, act=ACT_RELU, gamma=1.0):
            bn.running_mean.zero_(); bn.running_var.fill_(1.0); bn.weight.fill_(1.0); bn.bias.zero_()
            e = _conv_entry(conv, None, cin_pad=8 if conv.in_channels == 3 else None)
            y, st = ops.conv2d(x, e["wf"], e["k"], e["k"], stride=e["stride"], pad=e["pad"], want_stats=True, cout=e["cout"])
            sums = ops.colsum(st.view(st.shape[0], -1))
            n = y.numel() // e["cout"]
            mean = sums[: e["cout"]] / n
            var = (sums[e["cout"]:] / n - mean * mean).clamp_(min=1e-6)
            bn.running_mean.copy_(mean)
            bn.running_var.copy_(var)
            bn.weight.fill_(gamma)
            e2 = _conv_entry(conv, bn, cin_pad=8 if conv.in_channels == 3 else None)
            return _fwd(e2, x, act=act, res=res)

        cur = ops.maxpool3x3s2(fit(b.conv1, b.bn1, x_nhwc8))
        for li in range(1, 5):
            for blk in getattr(b, "layer%d" % li):
                o1 = fit(blk.conv1, blk.bn1, cur)
                o2 = fit(blk.conv2, blk.bn2, o1)
                idt = cur if blk.downsample is None else fit(blk.downsample[0], blk.downsample[1], cur, act=ACT_NONE)
                cur = fit(blk.conv3, blk.bn3, o2, res=idt, gamma=0.5)   # 0.5: keep the residual sum from growing
        self.invalidate()


# ======================================================================================================================
# box utilities (fp32 tensor arithmetic on the GPU)
# ======================================================================================================================
def box_area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def box_iou(a, b):
    return ops.box_iou(a.float().contiguous(), b.float().contiguous())


def clip_boxes_to_image(boxes, size):
    h, w = size
    bx = boxes[..., 0::2].clamp(min=0, max=w)
    by = boxes[..., 1::2].clamp(min=0, max=h)
    return torch.stack((bx, by), dim=boxes.dim()).reshape(boxes.shape)


class BoxCoder:
    def __init__(self, weights, bbox_xform_clip=math.log(1000.0 / 16)):
        self.weights, self.bbox_xform_clip = weights, bbox_xform_clip

    def encode_single(self, ref, prop):
        wx, wy, ww, wh = self.weights
        px1, py1, px2, py2 = [prop[:, i].unsqueeze(1) for i in range(4)]
        rx1, ry1, rx2, ry2 = [ref[:, i].unsqueeze(1) for i in range(4)]
        ew, eh = px2 - px1, py2 - py1
        ecx, ecy = px1 + 0.5 * ew, py1 + 0.5 * eh
        gw, gh = rx2 - rx1, ry2 - ry1
        gcx, gcy = rx1 + 0.5 * gw, ry1 + 0.5 * gh
        return torch.cat((wx * (gcx - ecx) / ew, wy * (gcy - ecy) / eh, ww * torch.log(gw / ew), wh * torch.log(gh / eh)), dim=1)

    def encode(self, reference_boxes, proposals):
        n = [len(b) for b in reference_boxes]
        return self.encode_single(torch.cat(reference_boxes, 0), torch.cat(proposals, 0)).split(n, 0)

    def decode_single(self, codes, boxes):
        boxes = boxes.to(codes.dtype)
        w, h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
        cx, cy = boxes[:, 0] + 0.5 * w, boxes[:, 1] + 0.5 * h
        wx, wy, ww, wh = self.weights
        dx, dy = codes[:, 0::4] / wx, codes[:, 1::4] / wy
        dw = torch.clamp(codes[:, 2::4] / ww, max=self.bbox_xform_clip)
        dh = torch.clamp(codes[:, 3::4] / wh, max=self.bbox_xform_clip)
        pcx, pcy = dx * w[:, None] + cx[:, None], dy * h[:, None] + cy[:, None]
        pw, ph = torch.exp(dw) * w[:, None], torch.exp(dh) * h[:, None]
        return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), dim=2).flatten(1)

    def decode(self, rel_codes, boxes):
        cat = torch.cat(list(boxes), dim=0)
        total = cat.shape[0]
        if total > 0:
            rel_codes = rel_codes.reshape(total, -1)
        pred = self.decode_single(rel_codes, cat)
        if total > 0:
            pred = pred.reshape(total, -1, 4)
        return pred


class Matcher:
    BELOW_LOW_THRESHOLD = -1
    BETWEEN_THRESHOLDS = -2

    def __init__(self, high_threshold, low_threshold, allow_low_quality_matches=False):
        self.high_threshold, self.low_threshold = high_threshold, low_threshold
        self.allow_low_quality_matches = allow_low_quality_matches

    def __call__(self, mq):
        vals, matches = mq.max(dim=0)
        all_matches = matches.clone() if self.allow_low_quality_matches else None
        below = vals < self.low_threshold
        between = (vals >= self.low_threshold) & (vals < self.high_threshold)
        matches = torch.where(below, torch.full_like(matches, self.BELOW_LOW_THRESHOLD), matches)
        matches = torch.where(between, torch.full_like(matches, self.BETWEEN_THRESHOLDS), matches)
        if self.allow_low_quality_matches:
            best_per_gt, _ = mq.max(dim=1)
            is_best = (mq == best_per_gt[:, None]).any(dim=0)
            matches = torch.where(is_best, all_matches, matches)
        return matches


class BalancedPositiveNegativeSampler:
    """torchvision semantics [EXT]: per image two `randperm` draws (positives, negatives).  `randperm_fn(n, device)`
    is injectable; the default draws `torch.randperm(n, device=device)` in the reference's call order."""

    def __init__(self, batch_size_per_image, positive_fraction, randperm_fn=None):
        self.batch_size_per_image, self.positive_fraction = batch_size_per_image, positive_fraction
        self.randperm_fn = randperm_fn

    def _perm(self, n, device):
        if self.randperm_fn is not None:
            return self.randperm_fn(n).to(device)
        return torch.randperm(n, device=device)

    def __call__(self, matched_idxs):
        # one host sync for all images' class counts instead of several per image
        pos_l = [torch.where(m >= 1)[0] for m in matched_idxs]
        neg_l = [torch.where(m == 0)[0] for m in matched_idxs]
        pos_idx, neg_idx = [], []
        for m, positive, negative in zip(matched_idxs, pos_l, neg_l):
            num_pos = min(positive.numel(), int(self.batch_size_per_image * self.positive_fraction))
            num_neg = min(negative.numel(), self.batch_size_per_image - num_pos)
            p = positive[self._perm(positive.numel(), m.device)[:num_pos]]
            n = negative[self._perm(negative.numel(), m.device)[:num_neg]]
            pm = torch.zeros_like(m, dtype=torch.uint8)
            nm = torch.zeros_like(m, dtype=torch.uint8)
            pm[p] = 1
            nm[n] = 1
            pos_idx.append(pm)
            neg_idx.append(nm)
        return pos_idx, neg_idx


def _batched_nms_padded(boxes, scores, idxs, valid, iou_thr, top_n):
    """Batched, padded form of torchvision.ops.batched_nms (coordinate-offset trick [EXT]).

    boxes [B,n,4], scores [B,n], idxs [B,n] (level / class), valid [B,n] bool.  Returns (order [B,n] = candidate index
    sorted by descending score, sel [B,n] bool in sorted order = kept and within the first `top_n` kept, counts [B]).
    """
    B, n = scores.shape
    neg = torch.full_like(scores, float("-inf"))
    key = torch.where(valid, scores, neg)
    order = torch.sort(key, dim=1, descending=True, stable=True)[1]
    counts = valid.sum(dim=1).to(torch.int32)
    bmax = torch.where(valid[..., None], boxes, torch.full_like(boxes, float("-inf"))).amax(dim=(1, 2))
    bmax = torch.where(counts > 0, bmax, torch.zeros_like(bmax))
    offsets = idxs.to(boxes) * (bmax[:, None] + 1)          # python scalar: a `torch.tensor(1).to(device)` is a blocking H2D copy
    shifted = boxes + offsets[:, :, None]
    sorted_boxes = torch.gather(shifted, 1, order[:, :, None].expand(-1, -1, 4)).contiguous()
    keep = ops.nms_sorted_batched(sorted_boxes, counts, iou_thr, max_keep=top_n)
    rank = torch.cumsum(keep.to(torch.int32), dim=1)
    sel = keep & (rank <= top_n)
    return order, sel, sel.sum(dim=1)


def _batched_nms_pick(boxes, scores, idxs, valid, iou_thr, top_n):
    """_batched_nms_padded + the compaction its callers do: -> (pick [B, min(top_n, n)] = candidate index of the k-th survivor
    in descending-score order (padding past the count), counts [B]).  On the GPU: one sort + three launches (ops.batched_nms_pick:
    shifted sorted boxes, suppression mask, serial reduce that emits the ordered survivors) instead of ~40."""
    if not boxes.is_cuda:
        order, sel, counts = _batched_nms_padded(boxes, scores, idxs, valid, iou_thr, top_n)
        return torch.gather(order, 1, _front(sel, top_n)), counts
    key = torch.where(valid, scores, float("-inf"))            # (scalar `other`: one launch, no full_like)
    n = key.shape[1]
    if n <= 4096 and key.dtype == torch.float32:
        # full descending stable order of every row by the radix-select + in-LDS sort kernel (one launch; rocprim's segmented radix sort
        # behind torch.sort is 2-3 launches of 17-25 us at these sizes); same order: equal keys by ascending index
        order = ops.topk_rows_segments(key.contiguous(), [n], n)
    else:
        order = torch.sort(key, dim=1, descending=True, stable=True)[1]
    return ops.batched_nms_pick(boxes, idxs, valid, order, iou_thr, top_n)


_NMS_SEGMENTS = os.environ.get("HD_NMS_SEGMENTS", "1") != "0"      # A/B knob: 0 = one scan per image over all levels


def _batched_nms_pick_segments(boxes, scores, seg_sizes, valid, iou_thr, top_n):
    """`_batched_nms_pick` for candidates whose NMS categories are the SEGMENTS `seg_sizes` of every row, each segment already in
    descending score order (what the RPN hands over: per-level top-k lists).  Same (pick, counts): the survivors of the
    category-shifted NMS in global descending-score order -- but one greedy scan per (image, level) instead of one per image."""
    return ops.batched_nms_pick_segments(boxes, scores, valid, seg_sizes, iou_thr, top_n)


# ======================================================================================================================
# RPN
# ======================================================================================================================
class AnchorGenerator(nn.Module):
    def __init__(self, sizes=((32,), (64,), (128,), (256,), (512,)), aspect_ratios=((0.5, 1.0, 2.0),) * 5):
        super().__init__()
        self.sizes, self.aspect_ratios = sizes, aspect_ratios
        self._cache = {}

    def num_anchors_per_location(self):
        return [len(s) * len(a) for s, a in zip(self.sizes, self.aspect_ratios)]

    @staticmethod
    def generate_anchors(scales, aspect_ratios, device):
        scales = torch.as_tensor(scales, dtype=torch.float32, device=device)
        ratios = torch.as_tensor(aspect_ratios, dtype=torch.float32, device=device)
        h_r = torch.sqrt(ratios)
        w_r = 1 / h_r
        ws = (w_r[:, None] * scales[None, :]).view(-1)
        hs = (h_r[:, None] * scales[None, :]).view(-1)
        return (torch.stack([-ws, -hs, ws, hs], dim=1) / 2).round()

    def forward(self, image_list, feature_maps):
        # the product's feature maps are NHWC (image lists of CustomGeneralizedRCNNTransform carry `layout`); torchvision-style callers pass NCHW
        nhwc = getattr(image_list, "layout", None) is not None or feature_maps[0].dtype == torch.float16
        grid_sizes = [(fm.shape[1], fm.shape[2]) if nhwc else tuple(fm.shape[-2:]) for fm in feature_maps]
        ih, iw = image_list.image_sizes[0] if getattr(image_list, "layout", None) else image_list.tensors.shape[-2:]
        device = feature_maps[0].device
        key = (tuple(grid_sizes), ih, iw, str(device))
        if key not in self._cache:
            per_level = []
            for (gh, gw), s, a in zip(grid_sizes, self.sizes, self.aspect_ratios):
                sh, sw = ih // gh, iw // gw
                sx = torch.arange(0, gw, dtype=torch.int32, device=device) * sw
                sy = torch.arange(0, gh, dtype=torch.int32, device=device) * sh
                yy, xx = torch.meshgrid(sy, sx, indexing="ij")
                xx, yy = xx.reshape(-1), yy.reshape(-1)
                shifts = torch.stack((xx, yy, xx, yy), dim=1)
                per_level.append((shifts.view(-1, 1, 4) + self.generate_anchors(s, a, device).view(1, -1, 4)).reshape(-1, 4))
            self._cache = {key: torch.cat(per_level)}
        allv = self._cache[key]
        return [allv for _ in range(len(image_list.image_sizes))]


class _RPNHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hook, head, n_active, nlev, *feats):
        ctx.has_acts = len(feats) > nlev
        ctx.set_materialize_grads(False)
        feats = feats[:nlev]
        P = head.pack()
        nl = len(feats)
        tfull = _fwd_many([P["conv"]] * nl, feats, act=ACT_RELU)
        ts = [t[:n_active] for t in tfull]
        heads = _fwd_many([P["cls"]] * nl + [P["box"]] * nl, list(tfull) + list(tfull), f32="nhwc")      # 2 x levels 1x1 convs, one grid
        outs = []
        for li in range(nl):
            outs.append(heads[li])
            outs.append(heads[nl + li])
        ctx.head, ctx.ts, ctx.na, ctx.n = head, ts, n_active, feats[0].shape[0]
        ctx.feats = [f[:n_active] for f in feats] if head.train_params else None
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        P = ctx.head.pack()
        dfeats = []
        na = ctx.na
        nl = len(ctx.ts)
        if (not ctx.head.train_params and all(g is not None for g in grads[:2 * nl]) and (na == ctx.n or ctx.has_acts)):
            # frozen head, every level has both gradients (the training step): the three data-gradient convs of ALL levels as three
            # grids -- cls, then box (+ cls result as residual, ReLU mask), then the shared 3x3 conv -- instead of 3 x levels launches
            hws = [(t.shape[1], t.shape[2]) for t in ctx.ts]
            both = _head_grads_nhwc16_many([grads[2 * i][:na] for i in range(nl)] + [grads[2 * i + 1][:na] for i in range(nl)], hws + hws,
                                           [P["cls"]["cout_p"]] * nl + [P["box"]["cout_p"]] * nl, P["cls"]["wf"].dtype)
            gls, grs = both[:nl], both[nl:]
            dts = _dgrad_many([P["cls"]] * nl, gls, hws)
            dts = _dgrad_many([P["box"]] * nl, grs, hws, ress=dts, masks=list(ctx.ts))
            dfeats = _dgrad_many([P["conv"]] * nl, dts, hws)
            ctx.ts = None
            if ctx.has_acts:
                return (None, None, None, None) + (None,) * len(dfeats) + tuple(dfeats)
            return (None, None, None, None) + tuple(dfeats)
        for i, t in enumerate(ctx.ts):
            N, H, W, _ = t.shape
            dl, dr = grads[2 * i], grads[2 * i + 1]
            dl = None if dl is None else dl[:na]
            dr = None if dr is None else dr[:na]
            hw = (H, W)
            dt = None
            tp, inv = ctx.head.train_params, 1.0 / ctx.head.grad_scale
            if dl is not None:
                gl = _head_grad_nhwc16(dl, H, W, P["cls"]["cout_p"], P["cls"]["wf"].dtype)
                dt = _dgrad(P["cls"], gl, hw)
                if tp:
                    _wgrad_into(ctx.head.cls_logits.weight, P["cls"], t, gl, inv)
                    _bgrad_into(ctx.head.cls_logits.bias, gl, inv)
            if dr is not None:
                gr = _head_grad_nhwc16(dr, H, W, P["box"]["cout_p"], P["box"]["wf"].dtype)
                dt = _dgrad(P["box"], gr, hw, res=dt, mask=t)
                if tp:
                    _wgrad_into(ctx.head.bbox_pred.weight, P["box"], t, gr, inv)
                    _bgrad_into(ctx.head.bbox_pred.bias, gr, inv)
            elif dt is not None:
                dt = ops.relu_bwd(dt, t)
            if tp and dt is not None:
                _wgrad_into(ctx.head.conv.weight, P["conv"], ctx.feats[i], dt, inv)
                _bgrad_into(ctx.head.conv.bias, dt, inv)
            df = None if dt is None else _dgrad(P["conv"], dt, hw)
            if df is not None and na < ctx.n and not ctx.has_acts:
                full = torch.zeros((ctx.n,) + tuple(df.shape[1:]), dtype=df.dtype, device=df.device)
                full[:na] = df
                df = full
            dfeats.append(df)
        ctx.ts = None
        if ctx.has_acts:
            return (None, None, None, None) + (None,) * len(dfeats) + tuple(dfeats)
        return (None, None, None, None) + tuple(dfeats)


class RPNHead(nn.Module):
    def __init__(self, in_channels=256, num_anchors=3):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, 3, padding=1)
        self.cls_logits = nn.Conv2d(in_channels, num_anchors, 1)
        self.bbox_pred = nn.Conv2d(in_channels, num_anchors * 4, 1)
        for layer in self.children():
            nn.init.normal_(layer.weight, std=0.01)
            nn.init.constant_(layer.bias, 0)
        self._pack, self._hook = None, None
        self.train_params, self.grad_scale = False, 1.0

    def invalidate(self):
        self._pack = None

    def pack(self):
        if self._pack is None:
            self._pack = dict(conv=_conv_entry(self.conv), cls=_conv_entry(self.cls_logits), box=_conv_entry(self.bbox_pred))
        return self._pack

    def forward(self, x, n_active=None):
        """x: list of NHWC fp16 feature maps.  Returns (logits, bbox_reg): lists of NCHW fp32 tensors per level."""
        feats = list(x)
        if self._hook is None or self._hook.device != feats[0].device:
            self._hook = torch.zeros(1, device=feats[0].device, requires_grad=True)
        na = feats[0].shape[0] if n_active is None else n_active
        acts = _active_views(feats, na) if na < feats[0].shape[0] else None
        outs = _RPNHeadFn.apply(self._hook, self, na, len(feats), *feats, *(acts or ()))
        return list(outs[0::2]), list(outs[1::2])


def permute_and_flatten(layer, N, A, C, H, W):
    return layer.view(N, -1, C, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, C)


def concat_box_prediction_layers(box_cls, box_regression):
    cls_f, reg_f = [], []
    for c, r in zip(box_cls, box_regression):
        N, AxC, H, W = c.shape
        A = r.shape[1] // 4
        C = AxC // A
        cls_f.append(permute_and_flatten(c, N, A, C, H, W))
        reg_f.append(permute_and_flatten(r, N, A, 4, H, W))
    return torch.cat(cls_f, dim=1).flatten(0, -2), torch.cat(reg_f, dim=1).reshape(-1, 4)


class RegionProposalNetwork(nn.Module):
    def __init__(self, anchor_generator=None, head=None, fg_iou_thresh=0.7, bg_iou_thresh=0.3, batch_size_per_image=256,
                 positive_fraction=0.5, pre_nms_top_n=None, post_nms_top_n=None, nms_thresh=0.7, score_thresh=0.0):
        super().__init__()
        self.anchor_generator = anchor_generator or AnchorGenerator()
        self.head = head or RPNHead()
        self.box_coder = BoxCoder(weights=(1.0, 1.0, 1.0, 1.0))
        self.proposal_matcher = Matcher(fg_iou_thresh, bg_iou_thresh, allow_low_quality_matches=True)
        self.fg_bg_sampler = BalancedPositiveNegativeSampler(batch_size_per_image, positive_fraction)
        self._pre_nms_top_n = pre_nms_top_n or dict(training=2000, testing=1000)
        self._post_nms_top_n = post_nms_top_n or dict(training=2000, testing=1000)
        self.nms_thresh, self.score_thresh, self.min_size = nms_thresh, score_thresh, 1e-3

    def pre_nms_top_n(self):
        return self._pre_nms_top_n["training" if self.training else "testing"]

    def post_nms_top_n(self):
        return self._post_nms_top_n["training" if self.training else "testing"]

    def assign_targets_to_anchors(self, anchors, targets):
        labels, matched = [], []
        for a, t in zip(anchors, targets):
            gt = t["boxes"]
            if gt.numel() == 0:
                matched.append(torch.zeros(a.shape, dtype=torch.float32, device=a.device))
                labels.append(torch.zeros((a.shape[0],), dtype=torch.float32, device=a.device))
                continue
            mi = self.proposal_matcher(box_iou(gt, a))
            matched.append(gt[mi.clamp(min=0)])
            lab = (mi >= 0).to(torch.float32)
            lab = torch.where(mi == Matcher.BELOW_LOW_THRESHOLD, torch.zeros_like(lab), lab)
            lab = torch.where(mi == Matcher.BETWEEN_THRESHOLDS, torch.full_like(lab, -1.0), lab)
            labels.append(lab)
        return labels, matched

    def _get_top_n_idx(self, objectness, num_anchors_per_level):
        if (objectness.is_cuda and objectness.dtype == torch.float32 and len(num_anchors_per_level) <= 8 and
                min(self.pre_nms_top_n(), max(num_anchors_per_level)) <= 4096):      # limits of hd_topk_select_rows
            # radix select + in-LDS sort per (image, level): the same lists as the per-level stable sorts below (descending
            # score, ties by lowest index) without their ~45 merge-sort launches
            return ops.topk_rows_segments(objectness.contiguous(), list(num_anchors_per_level), self.pre_nms_top_n())
        r, offset = [], 0
        for ob in objectness.split(num_anchors_per_level, 1):
            n = ob.shape[1]
            idx = torch.sort(ob, dim=1, descending=True, stable=True)[1][:, : min(self.pre_nms_top_n(), n)]
            r.append(idx + offset)
            offset += n
        return torch.cat(r, dim=1)

    def filter_proposals(self, proposals, objectness, image_shapes, num_anchors_per_level):
        n_img = proposals.shape[0]
        device = proposals.device
        objectness = objectness.detach().reshape(n_img, -1)
        levels = torch.cat([torch.full((n,), i, dtype=torch.int64, device=device) for i, n in enumerate(num_anchors_per_level)], 0)
        levels = levels.reshape(1, -1).expand_as(objectness)
        top = self._get_top_n_idx(objectness, num_anchors_per_level)
        bidx = torch.arange(n_img, device=device)[:, None]
        objectness, levels, proposals = objectness[bidx, top], levels[bidx, top], proposals[bidx, top]
        prob = torch.sigmoid(objectness)
        if len(set(tuple(s) for s in image_shapes)) != 1:
            raise NotImplementedError("hallucidet_amd: batched filter_proposals needs one image size per batch (fixed_size transform)")
        boxes = clip_boxes_to_image(proposals, image_shapes[0])
        ws, hs = boxes[..., 2] - boxes[..., 0], boxes[..., 3] - boxes[..., 1]
        valid = (ws >= self.min_size) & (hs >= self.min_size) & (prob >= self.score_thresh)
        order, sel, counts = _batched_nms_padded(boxes, prob, levels, valid, self.nms_thresh, self.post_nms_top_n())
        sboxes = torch.gather(boxes, 1, order[:, :, None].expand(-1, -1, 4))
        sscores = torch.gather(prob, 1, order)
        fb, fs = [], []
        for i in range(n_img):       # boolean selection = one sync per image; sizes are data dependent by contract
            fb.append(sboxes[i][sel[i]])
            fs.append(sscores[i][sel[i]])
        return fb, fs

    def compute_loss(self, objectness, pred_bbox_deltas, labels, regression_targets):
        pos, neg = self.fg_bg_sampler(labels)
        pos = torch.where(torch.cat(pos, dim=0))[0]
        neg = torch.where(torch.cat(neg, dim=0))[0]
        sampled = torch.cat([pos, neg], dim=0)
        objectness = objectness.flatten()
        labels = torch.cat(labels, dim=0)
        regression_targets = torch.cat(regression_targets, dim=0)
        box_loss = F.smooth_l1_loss(pred_bbox_deltas[pos], regression_targets[pos], beta=1 / 9, reduction="sum") / sampled.numel()
        obj_loss = F.binary_cross_entropy_with_logits(objectness[sampled], labels[sampled])
        return obj_loss, box_loss


# ======================================================================================================================
# RoI heads
# ======================================================================================================================
class _RoIAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rois, levels, cfg, nlev, *feats):
        scales, P, sr = cfg[:3]
        ctx.has_acts = len(feats) > nlev
        acts, feats = feats[nlev:], feats[:nlev]
        ctx.save_for_backward(rois, levels)
        ctx.cfg = cfg
        ctx.shapes = [tuple(a.shape) for a in acts] if ctx.has_acts else [tuple(f.shape) for f in feats]
        return ops.roi_align_ml(list(feats), scales, rois, levels, P, P, sr)

    @staticmethod
    def backward(ctx, dout):
        rois, levels = ctx.saved_tensors
        scales, P, sr = ctx.cfg[:3]
        if P == 7 and sr == 2 and dout.shape[-1] % 32 == 0:
            # gather form: no atomics, deterministic, fp16 written once; cfg[3] = number of leading images that own RoIs
            n_images = ctx.cfg[3] if len(ctx.cfg) > 3 else None
            dfs = ops.roi_align_ml_bwd_gather(dout, rois, levels, ctx.shapes, scales, sr, n_images)
        else:
            dfs = [ops.f32_to_f16(d, dtype=dout.dtype) for d in ops.roi_align_ml_bwd(dout, rois, levels, ctx.shapes, scales, sr)]
        if ctx.has_acts:               # ctx.shapes are the [n_active] views' shapes: the maps cover exactly the images that own RoIs
            return (None, None, None, None) + (None,) * len(dfs) + tuple(dfs)
        return (None, None, None, None) + tuple(dfs)


class MultiScaleRoIAlign(nn.Module):
    def __init__(self, featmap_names=("0", "1", "2", "3"), output_size=7, sampling_ratio=2, canonical_scale=224, canonical_level=4):
        super().__init__()
        self.featmap_names = list(featmap_names)
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.sampling_ratio = sampling_ratio
        self.canonical_scale, self.canonical_level, self.eps = canonical_scale, canonical_level, 1e-6

    _SCALE_CACHE = {}

    @staticmethod
    def infer_scale(feat_hw, original_size):
        key = (tuple(feat_hw), tuple(original_size))
        c = MultiScaleRoIAlign._SCALE_CACHE
        if key not in c:           # fp32 log2 / round as torchvision computes it; cached: 16 CPU tensor ops per step otherwise
            c[key] = [2 ** float(torch.tensor(float(a) / float(b)).log2().round()) for a, b in zip(feat_hw, original_size)][0]
        return c[key]

    def forward(self, x, boxes, image_shapes):
        feats = [v for k, v in x.items() if k in self.featmap_names]
        device = feats[0].device
        rois = torch.cat([torch.cat([torch.full_like(b[:, :1], i), b], dim=1) for i, b in enumerate(boxes)], dim=0).float()
        mh = max(s[0] for s in image_shapes)
        mw = max(s[1] for s in image_shapes)
        scales = [self.infer_scale((f.shape[1], f.shape[2]), (mh, mw)) for f in feats]
        k_min = int(-math.log2(scales[0]))
        k_max = int(-math.log2(scales[-1]))
        s = torch.sqrt(torch.cat([box_area(b) for b in boxes]).float())
        t = torch.floor(self.canonical_level + torch.log2(s / self.canonical_scale) + torch.tensor(self.eps, dtype=s.dtype, device=device))
        levels = (torch.clamp(t, min=k_min, max=k_max).to(torch.int64) - k_min).to(torch.int32)
        return _RoIAlignFn.apply(rois, levels, (scales, self.output_size[0], self.sampling_ratio), len(feats), *feats)


class _MLPFn(torch.autograd.Function):
    """fc6 (as a 7x7 'valid' convolution over the pooled RoI) -> ReLU -> fc7 -> ReLU, fp16 in / fp16 out."""

    @staticmethod
    def forward(ctx, x, head):
        P = head.pack()
        h6 = _fwd(P["fc6"], x, act=ACT_RELU)
        h7 = _fwd(P["fc7"], h6, act=ACT_RELU)
        ctx.head, ctx.h6, ctx.h7, ctx.xshape = head, h6, h7, tuple(x.shape)
        ctx.x = x if head.train_params else None
        ctx.need_dx = x.requires_grad
        return h7

    @staticmethod
    def backward(ctx, d7):
        P = ctx.head.pack()
        head = ctx.head
        d7 = ops.relu_bwd(d7.contiguous(), ctx.h7)
        d6 = _dgrad(P["fc7"], d7, (1, 1), mask=ctx.h6)
        if head.train_params:
            inv = 1.0 / head.grad_scale
            # nn.Linear weights seen as OIHW tensors: fc7 [rep,rep,1,1]; fc6 [rep,C,7,7] (torch flattens RoI features C,H,W)
            _wgrad_into(head.fc7.weight, P["fc7"], ctx.h6, d7, inv)
            _bgrad_into(head.fc7.bias, d7, inv)
            _wgrad_into(head.fc6.weight, P["fc6"], ctx.x, d6, inv)
            _bgrad_into(head.fc6.bias, d6, inv)
        dx = None
        if ctx.need_dx:
            dx = ops.conv2d(d6, P["fc6_t"], 1, 1, cout=P["fc6_t"].shape[0]).view(ctx.xshape)   # plain GEMM with the transposed fc6 matrix
        ctx.h6 = ctx.h7 = ctx.x = None
        return dx, None


class TwoMLPHead(nn.Module):
    def __init__(self, in_channels=256 * 7 * 7, representation_size=1024):
        super().__init__()
        self.fc6 = nn.Linear(in_channels, representation_size)
        self.fc7 = nn.Linear(representation_size, representation_size)
        self._pack = None
        self.train_params, self.grad_scale = False, 1.0

    def invalidate(self):
        self._pack = None

    def pack(self):
        if self._pack is None:
            rep, K = self.fc6.weight.shape
            c = K // 49
            w6 = self.fc6.weight.detach().float().view(rep, c, 7, 7)
            e6 = dict(k=7, stride=1, pad=0, cin=c, cout=rep, cin_p=c, cout_p=rep, bias=self.fc6.bias.detach().float().contiguous())
            e6["wf"], _ = ops.weight_prep(w6, cin_pad=c, cout_pad=rep, want_fwd=True, want_dgrad=False)
            e6["wd"] = None
            fc6_t = w6.permute(2, 3, 1, 0).reshape(49 * c, rep).to(e6["wf"].dtype).contiguous()
            w7 = self.fc7.weight.detach().float().view(rep, rep, 1, 1)
            e7 = dict(k=1, stride=1, pad=0, cin=rep, cout=rep, cin_p=rep, cout_p=rep, bias=self.fc7.bias.detach().float().contiguous())
            e7["wf"], e7["wd"] = ops.weight_prep(w7, want_fwd=True, want_dgrad=True)
            self._pack = dict(fc6=e6, fc7=e7, fc6_t=fc6_t)
        return self._pack

    def forward(self, x):
        """x: [R,7,7,C] fp16 (RoIAlign output).  Returns [R,1,1,rep] fp16."""
        if x.shape[0] == 0:
            return x.new_zeros((0, 1, 1, self.fc7.out_features))
        if torch.is_grad_enabled() and (x.requires_grad or self.train_params):
            return _MLPFn.apply(x, self)
        P = self.pack()
        return _fwd(P["fc7"], _fwd(P["fc6"], x, act=ACT_RELU), act=ACT_RELU)


class _PredictorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pred):
        P = pred.pack()
        ctx.pred = pred
        ctx.x = x if pred.train_params else None
        R = x.shape[0]
        cls, box = _fwd_many([P["cls"], P["box"]], [x, x], f32=True)          # the two tiny-N GEMMs (1024 -> K, 1024 -> 4K) as one grid
        return cls.view(R, -1), box.view(R, -1)

    @staticmethod
    def backward(ctx, dc, db):
        pred = ctx.pred
        P = pred.pack()
        R = dc.shape[0] if dc is not None else db.shape[0]
        tp, inv = pred.train_params, 1.0 / pred.grad_scale
        dx = None
        if dc is not None:
            g = ops.nchw_to_nhwc_resize(dc.contiguous().float().view(R, -1, 1, 1), 1, 1, P["cls"]["cout_p"], dtype=P["cls"]["wf"].dtype)
            dx = _dgrad(P["cls"], g, (1, 1))
            if tp:
                _wgrad_into(pred.cls_score.weight, P["cls"], ctx.x, g, inv)
                _bgrad_into(pred.cls_score.bias, g, inv)
        if db is not None:
            g = ops.nchw_to_nhwc_resize(db.contiguous().float().view(R, -1, 1, 1), 1, 1, P["box"]["cout_p"], dtype=P["box"]["wf"].dtype)
            dx = _dgrad(P["box"], g, (1, 1), res=dx)
            if tp:
                _wgrad_into(pred.bbox_pred.weight, P["box"], ctx.x, g, inv)
                _bgrad_into(pred.bbox_pred.bias, g, inv)
        ctx.x = None
        return dx, None


class FastRCNNPredictor(nn.Module):
    def __init__(self, in_channels, num_classes):
        super().__init__()
        self.cls_score = nn.Linear(in_channels, num_classes)
        self.bbox_pred = nn.Linear(in_channels, num_classes * 4)
        self._pack = None
        self.train_params, self.grad_scale = False, 1.0

    def invalidate(self):
        self._pack = None

    def pack(self):
        if self._pack is None:
            def ent(lin):
                o, i = lin.weight.shape
                e = dict(k=1, stride=1, pad=0, cin=i, cout=o, cin_p=i, cout_p=(o + 7) // 8 * 8, bias=lin.bias.detach().float().contiguous())
                e["wf"], e["wd"] = ops.weight_prep(lin.weight.detach().float().view(o, i, 1, 1), cout_pad=e["cout_p"], want_fwd=True, want_dgrad=True)
                return e
            self._pack = dict(cls=ent(self.cls_score), box=ent(self.bbox_pred))
        return self._pack

    def forward(self, x):
        """x: [R,1,1,1024] fp16.  Returns fp32 (scores [R,K], bbox_deltas [R,4K])."""
        if x.shape[0] == 0:
            z = torch.zeros((0, self.cls_score.out_features), device=x.device)
            return z, torch.zeros((0, self.bbox_pred.out_features), device=x.device)
        if torch.is_grad_enabled() and (x.requires_grad or self.train_params):
            return _PredictorFn.apply(x, self)
        P = self.pack()
        R = x.shape[0]
        cls, box = _fwd_many([P["cls"], P["box"]], [x, x], f32=True)          # the two tiny-N GEMMs (1024 -> K, 1024 -> 4K) as one grid
        return cls.view(R, -1), box.view(R, -1)


def fastrcnn_loss(class_logits, box_regression, labels, regression_targets):
    labels = torch.cat(labels, dim=0)
    regression_targets = torch.cat(regression_targets, dim=0)
    cls_loss = F.cross_entropy(class_logits, labels)
    pos = torch.where(labels > 0)[0]
    N = class_logits.shape[0]
    box_regression = box_regression.reshape(N, box_regression.size(-1) // 4, 4)
    box_loss = F.smooth_l1_loss(box_regression[pos, labels[pos]], regression_targets[pos], beta=1 / 9, reduction="sum")
    return cls_loss, box_loss / labels.numel()


class RoIHeads(nn.Module):
    def __init__(self, box_roi_pool=None, box_head=None, box_predictor=None, fg_iou_thresh=0.5, bg_iou_thresh=0.5,
                 batch_size_per_image=512, positive_fraction=0.25, bbox_reg_weights=None, score_thresh=0.05, nms_thresh=0.5,
                 detections_per_img=100, num_classes=91):
        super().__init__()
        self.box_roi_pool = box_roi_pool or MultiScaleRoIAlign()
        self.box_head = box_head or TwoMLPHead()
        self.box_predictor = box_predictor or FastRCNNPredictor(1024, num_classes)
        self.box_coder = BoxCoder(bbox_reg_weights or (10.0, 10.0, 5.0, 5.0))
        self.proposal_matcher = Matcher(fg_iou_thresh, bg_iou_thresh, allow_low_quality_matches=False)
        self.fg_bg_sampler = BalancedPositiveNegativeSampler(batch_size_per_image, positive_fraction)
        self.score_thresh, self.nms_thresh, self.detections_per_img = score_thresh, nms_thresh, detections_per_img
        self.keypoint_roi_pool = self.keypoint_head = self.keypoint_predictor = None
        self.mask_roi_pool = self.mask_head = self.mask_predictor = None

    def has_keypoint(self):
        return False

    def has_mask(self):
        return False

    def select_training_samples(self, proposals, targets):
        if targets is None:
            raise ValueError("targets should not be None")
        dtype = proposals[0].dtype
        gt_boxes = [t["boxes"].to(dtype) for t in targets]
        gt_labels = [t["labels"] for t in targets]
        proposals = [torch.cat((p, g)) for p, g in zip(proposals, gt_boxes)]
        matched_idxs, labels = [], []
        for p, g, gl in zip(proposals, gt_boxes, gt_labels):
            if g.numel() == 0:
                matched_idxs.append(torch.zeros((p.shape[0],), dtype=torch.int64, device=p.device))
                labels.append(torch.zeros((p.shape[0],), dtype=torch.int64, device=p.device))
                continue
            mi = self.proposal_matcher(box_iou(g, p))
            lab = gl[mi.clamp(min=0)].to(torch.int64)
            lab = torch.where(mi == Matcher.BELOW_LOW_THRESHOLD, torch.zeros_like(lab), lab)
            lab = torch.where(mi == Matcher.BETWEEN_THRESHOLDS, torch.full_like(lab, -1), lab)
            matched_idxs.append(mi.clamp(min=0))
            labels.append(lab)
        pos, neg = self.fg_bg_sampler(labels)
        matched_gt = []
        for i in range(len(proposals)):
            s = torch.where(pos[i] | neg[i])[0]
            proposals[i], labels[i], matched_idxs[i] = proposals[i][s], labels[i][s], matched_idxs[i][s]
            g = gt_boxes[i] if gt_boxes[i].numel() else torch.zeros((1, 4), dtype=dtype, device=proposals[i].device)
            matched_gt.append(g[matched_idxs[i]])
        return proposals, matched_idxs, labels, self.box_coder.encode(matched_gt, proposals)

    def postprocess_detections(self, class_logits, box_regression, proposals, image_shapes):
        num_classes = class_logits.shape[-1]
        device = class_logits.device
        per = [b.shape[0] for b in proposals]
        pred_boxes = self.box_coder.decode(box_regression.detach(), proposals)
        pred_scores = F.softmax(class_logits.detach(), -1)
        n_img, nmax = len(per), max(per) if per else 0
        ncand = nmax * (num_classes - 1)
        B = torch.zeros((n_img, ncand, 4), device=device)
        S = torch.zeros((n_img, ncand), device=device)
        Lb = torch.zeros((n_img, ncand), dtype=torch.int64, device=device)
        V = torch.zeros((n_img, ncand), dtype=torch.bool, device=device)
        off = 0
        for i, (n, shp) in enumerate(zip(per, image_shapes)):
            b = clip_boxes_to_image(pred_boxes[off:off + n], shp)[:, 1:].reshape(-1, 4)
            s = pred_scores[off:off + n][:, 1:].reshape(-1)
            l = torch.arange(1, num_classes, device=device).view(1, -1).expand(n, -1).reshape(-1)
            k = b.shape[0]
            B[i, :k], S[i, :k], Lb[i, :k] = b, s, l
            ws, hs = b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]
            V[i, :k] = (s > self.score_thresh) & (ws >= 1e-2) & (hs >= 1e-2)
            off += n
        order, sel, counts = _batched_nms_padded(B, S, Lb, V, self.nms_thresh, self.detections_per_img)
        sb = torch.gather(B, 1, order[:, :, None].expand(-1, -1, 4))
        ss = torch.gather(S, 1, order)
        sl = torch.gather(Lb, 1, order)
        ab, as_, al = [], [], []
        for i in range(n_img):
            ab.append(sb[i][sel[i]])
            as_.append(ss[i][sel[i]])
            al.append(sl[i][sel[i]])
        return ab, as_, al


# ======================================================================================================================
# model
# ======================================================================================================================
class FasterRCNN(nn.Module):
    def __init__(self, num_classes=91, min_size=800, max_size=1333):
        super().__init__()
        # torchvision's GeneralizedRCNNTransform is always replaced by the reference (detector.py:43-48); build that one.
        self.transform = CustomGeneralizedRCNNTransform(min_size=300, max_size=300, image_mean=[0.0], image_std=[1.0],
                                                        size_divisible=1, fixed_size=(300, 300))
        self.backbone = BackboneWithFPN()
        self.rpn = RegionProposalNetwork()
        self.roi_heads = RoIHeads(num_classes=num_classes)
        # True: eval_forward_fasterrcnn uses the batched/padded forms of the per-image loops (identical results)
        self.batched_heads = True

    def invalidate_packs(self):
        self.backbone.invalidate()
        self.rpn.head.invalidate()
        self.roi_heads.box_head.invalidate()
        self.roi_heads.box_predictor.invalidate()

    def set_trainable(self, flag=True, grad_scale=1.0):
        """Detector fine-tuning switch (train_detector.py / `train_det=True`): the hand-written backward chains also emit
        parameter gradients (into `param.grad`), for what torchvision's fasterrcnn_resnet50_fpn leaves trainable
        (trainable_backbone_layers=3 [EXT]: body.layer2-4 convs, FPN, RPN head, box head, predictor; FrozenBN, conv1 and
        layer1 stay fixed)."""
        for m in (self.backbone, self.rpn.head, self.roi_heads.box_head, self.roi_heads.box_predictor):
            m.train_params, m.grad_scale = bool(flag), float(grad_scale)
        for name, p in self.backbone.body.named_parameters():
            p.requires_grad_(bool(flag) and name.split(".")[0] in ("layer2", "layer3", "layer4"))
        for mod in (self.backbone.fpn, self.rpn, self.roi_heads):
            for p in mod.parameters():
                p.requires_grad_(bool(flag))

    def trainable_parameters(self):
        return [p for p in self.parameters() if p.requires_grad]

    @staticmethod
    def remap_state_dict_keys(state_dict):
        """torchvision >= 0.13 key names -> the 0.12 names this module (and the reference's checkpoints) use."""
        sd = OrderedDict()
        for k, v in state_dict.items():
            for i in range(4):
                k = k.replace("fpn.inner_blocks.%d.0." % i, "fpn.inner_blocks.%d." % i).replace("fpn.layer_blocks.%d.0." % i, "fpn.layer_blocks.%d." % i)
            k = k.replace("rpn.head.conv.0.0.", "rpn.head.conv.")
            if k.endswith("num_batches_tracked"):
                continue
            sd[k] = v
        return sd

    def load_state_dict(self, state_dict, strict=True):
        sd = self.remap_state_dict_keys(state_dict)
        out = super().load_state_dict(sd, strict=strict)
        self.invalidate_packs()
        return out

    def _apply(self, fn, *a, **k):
        self.invalidate_packs()
        return super()._apply(fn, *a, **k)


def fasterrcnn_resnet50_fpn(pretrained=False, progress=True, num_classes=91, pretrained_backbone=False, weights_path=None, **kwargs):
    """torchvision.models.detection.fasterrcnn_resnet50_fpn [EXT].  COCO weights cannot be downloaded offline
    (SURVEY App. D.8): `pretrained=True` is accepted for signature compatibility and ignored unless `weights_path`
    points at a local torchvision state_dict."""
    model = FasterRCNN(num_classes=num_classes)
    if weights_path is not None:
        model.load_state_dict(torch.load(weights_path, map_location="cpu"))
    return model


# ======================================================================================================================
# Batched (padded) forms of the per-image torchvision loops.  Same arithmetic and the same `randperm` call order as the
# list-based methods above (tests assert identical outputs); an order of magnitude fewer device launches and two host
# synchronisations per detector pass (sampler population sizes, detection counts).
# ======================================================================================================================
def stack_rows(ts):
    """torch.stack(ts) -- without the copy when `ts` are consecutive, equally shaped, contiguous views of ONE buffer (the staged
    targets of det_graph.py, and what `resize_boxes_many` hands back: views of one scaled tensor)."""
    t0 = ts[0]
    base, n = t0._base, t0.numel()
    if base is not None and n > 0 and t0.is_contiguous() and all(
            t._base is base and t.shape == t0.shape and t.is_contiguous() and t.storage_offset() == t0.storage_offset() + i * n
            for i, t in enumerate(ts)):
        return base.as_strided((len(ts),) + tuple(t0.shape), (n,) + tuple(t0.stride()), t0.storage_offset())
    return torch.stack(ts)


_PAD_IDX_CACHE = {}


def pad_targets(targets, device):
    """-> gt [N,G,4] fp32, labels [N,G] int64, valid [N,G] bool (G >= 1).
    One concatenation + one gather through a [N,G] row-index table (pad_sequence issues two device copies per target: 48
    launches for the 24 targets of a step).  The table depends on the box counts only: built on the host, cached per count
    tuple, uploaded from pinned memory without a stream synchronisation."""
    if targets and "_rows" in targets[0]:
        # staged targets (det_graph.py): G rows for every image already, zero rows as padding
        gt = stack_rows([t["boxes"].to(torch.float32) for t in targets])
        return gt, stack_rows([t["labels"] for t in targets]), gt[:, :, 2] > gt[:, :, 0]
    lens = tuple(int(t["boxes"].shape[0]) for t in targets)
    N, S = len(lens), sum(lens)
    G = max(1, max(lens))
    key = (lens, str(device))
    idx = _PAD_IDX_CACHE.get(key)
    if idx is None:
        rows, lo = [], 0
        for n in lens:
            rows.append(list(range(lo, lo + n)) + [S] * (G - n))      # row S of the sources = the all-zero padding row
            lo += n
        host = torch.tensor(rows, dtype=torch.int64).reshape(N, G)
        if torch.device(device).type == "cuda":
            idx = host.pin_memory().to(device, non_blocking=True)
        else:
            idx = host.to(device)
        if len(_PAD_IDX_CACHE) > 256:
            _PAD_IDX_CACHE.clear()
        _PAD_IDX_CACHE[key] = idx
    boxes = torch.cat([t["boxes"].to(torch.float32).reshape(-1, 4) for t in targets] + [torch.zeros((1, 4), dtype=torch.float32, device=device)], dim=0)
    labels = torch.cat([t["labels"].reshape(-1) for t in targets] + [torch.zeros((1,), dtype=targets[0]["labels"].dtype, device=device)], dim=0)
    gt, lb = boxes[idx], labels[idx]
    # real boxes are non-degenerate (checked upstream: x2 > x1), padding rows are all-zero: no host->device traffic
    valid = gt[:, :, 2] > gt[:, :, 0]
    return gt, lb, valid


def _match_batched(iou, gvalid, high, low, allow_low_quality):
    """Matcher over [N,G,A] IoUs (invalid GT rows excluded).  Returns matched idx [N,A] with -1 / -2 codes."""
    iou = torch.where(gvalid[:, :, None], iou, torch.full_like(iou, -1.0))
    vals, idx = iou.max(dim=1)
    m = torch.where(vals < low, torch.full_like(idx, Matcher.BELOW_LOW_THRESHOLD), idx)
    m = torch.where((vals >= low) & (vals < high), torch.full_like(idx, Matcher.BETWEEN_THRESHOLDS), m)
    if allow_low_quality:
        best = iou.max(dim=2).values
        is_best = ((iou == best[:, :, None]) & gvalid[:, :, None]).any(dim=1)
        m = torch.where(is_best, idx, m)
    return m


def _sample_batched(sampler, labels, host_counts=True):
    """BalancedPositiveNegativeSampler over labels [N,A] (>=1 positive, 0 negative, <0 ignored).
    Returns (pos_sel, neg_sel) bool [N,A] and the per-image (num_pos, num_neg) python ints."""
    N, A = labels.shape
    dev = labels.device
    if sampler.randperm_fn is None and labels.is_cuda:
        # one launch (ops.sample_pos_neg: class counts, radix select of the smallest random keys per class, membership
        # masks) after the key draw -- the same draw and the same subsets as _sample_batched_keys below, which took ~35
        B = sampler.batch_size_per_image
        keys = torch.randint(0, 1 << 30, (N, A), dtype=torch.int32, device=dev)
        pos_sel, neg_sel, counts = ops.sample_pos_neg(labels.to(torch.int64), keys, B, int(B * sampler.positive_fraction))
        if not host_counts:
            return pos_sel, neg_sel, counts
        return pos_sel, neg_sel, [tuple(t) for t in counts.tolist()]     # the one host sync of the sampler
    pos, neg = labels >= 1, labels == 0
    if sampler.randperm_fn is None:
        return _sample_batched_keys(sampler, pos, neg, host_counts)
    cnt = torch.stack([pos.sum(1), neg.sum(1)], dim=1).tolist()          # the one host sync of the sampler
    rows_p, cols_p, rows_n, cols_n, picked = [], [], [], [], []
    for i, (P_i, N_i) in enumerate(cnt):
        num_pos = min(P_i, int(sampler.batch_size_per_image * sampler.positive_fraction))
        num_neg = min(N_i, sampler.batch_size_per_image - num_pos)
        p1 = sampler._perm(P_i, dev)[:num_pos]
        p2 = sampler._perm(N_i, dev)[:num_neg]
        cols_p.append(p1 + i * A)
        cols_n.append(p2 + i * A)
        picked.append((num_pos, num_neg))
    rank_sel_p = torch.zeros(N * A, dtype=torch.bool, device=dev)
    rank_sel_n = torch.zeros(N * A, dtype=torch.bool, device=dev)
    rank_sel_p[torch.cat(cols_p)] = True
    rank_sel_n[torch.cat(cols_n)] = True
    rp = (torch.cumsum(pos, dim=1) - 1).clamp(min=0) + torch.arange(N, device=dev)[:, None] * A
    rn = (torch.cumsum(neg, dim=1) - 1).clamp(min=0) + torch.arange(N, device=dev)[:, None] * A
    pos_sel = pos & rank_sel_p[rp]
    neg_sel = neg & rank_sel_n[rn]
    return pos_sel, neg_sel, picked


def _sample_batched_keys(sampler, pos, neg, host_counts=True):
    """The sampler's draws for a whole batch in ONE selection: a uniformly random subset of size k of a population is the k
    smallest of iid random keys -- which is also how `torch.randperm` is implemented -- so every candidate gets a random
    key and each image takes the smallest keys of its positives and of its negatives.  Same distribution as the
    reference's 2 draws per image (uniform subsets of sizes min(P, 0.x*B) and min(N, B - num_pos)), 1 top-k instead of 2*N
    randperm launches.  Used when no `randperm_fn` is injected (parity tests inject one and take the per-image path above)."""
    N, A = pos.shape
    dev = pos.device
    B = sampler.batch_size_per_image
    cap_p = int(B * sampler.positive_fraction)
    # 30 random bits per candidate (collision probability per row ~A^2 / 2^31: a handful of ties in a million draws, broken
    # arbitrarily).  Only the B smallest keys of each population are ever used, so a row-wise top-k (one radix-select block
    # per row) over [positives' keys ; negatives' keys] replaces the full sort of every row (rocprim merge sort: ~20 launches
    # for the RPN's 21 765 anchors per image).
    r = torch.randint(0, 1 << 30, (N, A), dtype=torch.int32, device=dev)
    big = torch.full_like(r, 0x7FFFFFFF)
    k = min(B, A)
    keys = torch.stack([torch.where(pos, r, big), torch.where(neg, r, big)], dim=0).reshape(2 * N, A)
    order = torch.topk(keys, k, dim=1, largest=False, sorted=True)[1].reshape(2, N, k)
    P, Nn = pos.sum(1), neg.sum(1)
    num_pos = P.clamp(max=cap_p)
    num_neg = torch.minimum(Nn, B - num_pos)
    ar = torch.arange(k, device=dev)[None, :]
    pos_sel = torch.zeros_like(pos).scatter_(1, order[0], ar < num_pos[:, None])
    neg_sel = torch.zeros_like(neg).scatter_(1, order[1], ar < num_neg[:, None])
    if not host_counts:          # caller only needs the totals as device scalars: no host synchronisation at all
        return pos_sel, neg_sel, torch.stack([num_pos, num_neg], dim=1)
    picked = [tuple(t) for t in torch.stack([num_pos, num_neg], dim=1).tolist()]     # the one host sync of the sampler
    return pos_sel, neg_sel, picked


def _compact(mask_flat, total):
    """Indices of the True entries of a flat bool mask, in order, when their number `total` is already known on the host:
    rank by prefix sum, scatter the positions (False entries go to a spare slot) -- no sort."""
    n = mask_flat.numel()
    rank = torch.cumsum(mask_flat, 0) - 1
    tgt = torch.where(mask_flat, rank, torch.full_like(rank, total))
    out = torch.zeros(total + 1, dtype=torch.int64, device=mask_flat.device)
    out.scatter_(0, tgt, torch.arange(n, device=mask_flat.device))
    return out[:total]


def _front(sel, top):
    """[B,top] column indices of the True entries of sel [B,n], in order (rows with fewer than `top` are padded with 0):
    the stable-sort-by-flag this replaces cost a multi-pass merge sort per call."""
    B, n = sel.shape
    top = min(top, n)
    rank = torch.cumsum(sel, dim=1) - 1
    tgt = torch.where(sel & (rank < top), rank, torch.full_like(rank, top))
    out = torch.zeros((B, top + 1), dtype=torch.int64, device=sel.device)
    out.scatter_(1, tgt, torch.arange(n, device=sel.device)[None, :].expand(B, n))
    return out[:, :top]


def rpn_targets_sample_batched(rpn, anchors0, gt, gvalid, n_loss=None):
    """assign_targets_to_anchors + sampler for N images sharing one anchor set.  Depends on the targets only (not on
    the network), so a training step can run it -- and take the sampler's host sync -- before the detector trunk is
    even launched.  Returns what `rpn_loss_from_samples` needs."""
    N = gvalid.shape[0]
    # IoU + Matcher(0.7, 0.3, low-quality) + labels (1 / 0 / -1; 0 for GT-less images) + box_coder.encode of the matched GT
    # in two launches (ops.match_targets) instead of ~70 elementwise launches over [N,G,A] / [N,A] tensors
    _, lab, reg_t = ops.match_targets(gt, gvalid, None, anchors0, rpn.proposal_matcher.high_threshold, rpn.proposal_matcher.low_threshold,
                                      True, coder_weights=rpn.box_coder.weights)
    labels = lab.to(torch.float32)                   # BCE target dtype; the sampler takes the int64 labels as they are
    # the RPN losses only need the NUMBER of sampled anchors: keep it on the device (no host sync) unless a permutation
    # function is injected (parity tests), whose per-image loop needs the counts on the host anyway
    pos_sel, neg_sel, picked = _sample_batched(rpn.fg_bg_sampler, lab, host_counts=False)
    if n_loss is not None and n_loss < N:
        keep_img = _const_mask(N, n_loss, labels.device)
        pos_sel, neg_sel = pos_sel & keep_img, neg_sel & keep_img
        picked = picked[:n_loss]
    n_sampled = picked.sum() if torch.is_tensor(picked) else sum(a + b for a, b in picked)
    pos_f = pos_sel.reshape(-1)
    samp_f = pos_f | neg_sel.reshape(-1)
    # regression targets are only ever read at sampled positives (elsewhere they may be inf for GT-less images)
    return dict(labels=labels.reshape(-1).clamp(min=0), reg_t=reg_t.reshape(-1, 4), pos_f=pos_f, samp_f=samp_f, n_sampled=n_sampled)


_CONST_MASKS = {}


def _const_mask(n, k, device):
    """[n, 1] bool, True for the first k rows: constant per (n, k, device) -- built once instead of arange + compare per step."""
    key = (int(n), int(k), str(device))
    m = _CONST_MASKS.get(key)
    if m is None:
        m = _CONST_MASKS[key] = (torch.arange(n, device=device) < k)[:, None].contiguous()
    return m


_ARANGES = {}


def _arange(n, device):
    key = (int(n), str(device))
    a = _ARANGES.get(key)
    if a is None:
        a = _ARANGES[key] = torch.arange(n, device=device)
    return a


def _u8(t):
    return t.contiguous().view(torch.uint8) if t.dtype == torch.bool else t.to(torch.uint8).contiguous()


class _RPNLossFn(torch.autograd.Function):
    """RegionProposalNetwork.compute_loss as one forward (+ finish) and one backward launch (hd_rpn_loss / hd_rpn_loss_bwd);
    the torch-op form below is ~15 launches each way over the N*A anchors."""

    @staticmethod
    def forward(ctx, objectness, deltas, labels, reg_t, pos_f, samp_f, n_sampled):
        from .. import _abi
        lib = _abi.load()
        obj, dl = objectness.detach().reshape(-1).contiguous().float(), deltas.detach().reshape(-1, 4).contiguous().float()
        lab, rt = labels.reshape(-1).contiguous().float(), reg_t.reshape(-1, 4).contiguous().float()
        pos, samp = _u8(pos_f.reshape(-1)), _u8(samp_f.reshape(-1))
        T = obj.numel()
        dev_n = n_sampled if torch.is_tensor(n_sampled) else None
        if dev_n is not None:
            dev_n = dev_n.reshape(1).to(torch.int64)
        host_n = 0.0 if dev_n is not None else float(n_sampled)
        ws = torch.empty(512, dtype=torch.float32, device=obj.device)
        out = torch.empty(2, dtype=torch.float32, device=obj.device)
        s = torch.cuda.current_stream().cuda_stream
        _abi.check(lib.hd_rpn_loss(_abi.ptr(obj), _abi.ptr(dl), _abi.ptr(lab), _abi.ptr(rt), _abi.ptr(pos), _abi.ptr(samp), T, 1.0 / 9,
                                   _abi.ptr(dev_n), host_n, _abi.ptr(ws), _abi.ptr(out), s), "hd_rpn_loss")
        ctx.save_for_backward(obj, dl, lab, rt, pos, samp)
        ctx.dev_n, ctx.host_n, ctx.shapes = dev_n, host_n, (objectness.shape, deltas.shape)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_obj, g_box):
        from .. import _abi
        lib = _abi.load()
        obj, dl, lab, rt, pos, samp = ctx.saved_tensors
        d_obj, d_dl = torch.empty_like(obj), torch.empty_like(dl)
        go = None if g_obj is None else g_obj.contiguous().float()
        gb = None if g_box is None else g_box.contiguous().float()
        _abi.check(lib.hd_rpn_loss_bwd(_abi.ptr(obj), _abi.ptr(dl), _abi.ptr(lab), _abi.ptr(rt), _abi.ptr(pos), _abi.ptr(samp), obj.numel(), 1.0 / 9,
                                       _abi.ptr(go), _abi.ptr(gb), _abi.ptr(ctx.dev_n), ctx.host_n, _abi.ptr(d_obj), _abi.ptr(d_dl),
                                       torch.cuda.current_stream().cuda_stream), "hd_rpn_loss_bwd")
        return d_obj.reshape(ctx.shapes[0]), d_dl.reshape(ctx.shapes[1]), None, None, None, None, None


class _FastRCNNLossFn(torch.autograd.Function):
    """roi_heads.fastrcnn_loss as one forward (+ finish) and one backward launch (hd_fastrcnn_loss / _bwd)."""

    @staticmethod
    def forward(ctx, class_logits, box_regression, labels, reg_t, n_valid=None):
        """n_valid (device int64 scalar): fixed-size RoI list, rows with label < 0 are padding and the divisor is n_valid."""
        from .. import _abi
        lib = _abi.load()
        lg, br = class_logits.detach().contiguous().float(), box_regression.detach().contiguous().float()
        lab, rt = labels.contiguous().to(torch.int64), reg_t.contiguous().float()
        R, K = lg.shape
        ws = torch.empty(512, dtype=torch.float32, device=lg.device)
        out = torch.empty(2, dtype=torch.float32, device=lg.device)
        st = torch.cuda.current_stream().cuda_stream
        if n_valid is None:
            _abi.check(lib.hd_fastrcnn_loss(_abi.ptr(lg), _abi.ptr(br), _abi.ptr(lab), _abi.ptr(rt), R, K, 1.0 / 9, _abi.ptr(ws), _abi.ptr(out), st),
                       "hd_fastrcnn_loss")
            ctx.save_for_backward(lg, br, lab, rt)
        else:
            nv = n_valid.reshape(1).to(torch.int64).contiguous()
            _abi.check(lib.hd_fastrcnn_loss_masked(_abi.ptr(lg), _abi.ptr(br), _abi.ptr(lab), _abi.ptr(rt), R, K, 1.0 / 9, _abi.ptr(nv), _abi.ptr(ws),
                                                   _abi.ptr(out), st), "hd_fastrcnn_loss_masked")
            ctx.save_for_backward(lg, br, lab, rt, nv)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_cls, g_box):
        from .. import _abi
        lib = _abi.load()
        saved = ctx.saved_tensors
        lg, br, lab, rt = saved[:4]
        R, K = lg.shape
        d_lg, d_br = torch.empty_like(lg), torch.empty_like(br)
        gc = None if g_cls is None else g_cls.contiguous().float()
        gb = None if g_box is None else g_box.contiguous().float()
        st = torch.cuda.current_stream().cuda_stream
        if len(saved) == 4:
            _abi.check(lib.hd_fastrcnn_loss_bwd(_abi.ptr(lg), _abi.ptr(br), _abi.ptr(lab), _abi.ptr(rt), R, K, 1.0 / 9, _abi.ptr(gc), _abi.ptr(gb),
                                                _abi.ptr(d_lg), _abi.ptr(d_br), st), "hd_fastrcnn_loss_bwd")
        else:
            _abi.check(lib.hd_fastrcnn_loss_masked_bwd(_abi.ptr(lg), _abi.ptr(br), _abi.ptr(lab), _abi.ptr(rt), R, K, 1.0 / 9, _abi.ptr(saved[4]),
                                                       _abi.ptr(gc), _abi.ptr(gb), _abi.ptr(d_lg), _abi.ptr(d_br), st), "hd_fastrcnn_loss_masked_bwd")
        return d_lg, d_br, None, None, None


def rpn_loss_from_samples(st, objectness, deltas):
    pos_f, samp_f, n_sampled = st["pos_f"], st["samp_f"], st["n_sampled"]
    if not (objectness.is_cuda and objectness.dtype == torch.float32 and deltas.dtype == torch.float32):
        raise RuntimeError("hallucidet_amd: the RPN loss runs in hd_rpn_loss on fp32 CUDA head outputs (got %s %s on %s); there is no CPU path"
                           % (objectness.dtype, deltas.dtype, objectness.device))
    return _RPNLossFn.apply(objectness, deltas, st["labels"], st["reg_t"], pos_f, samp_f, n_sampled)


def rpn_targets_loss_batched(rpn, anchors0, gt, gvalid, objectness, deltas, n_loss=None):
    """assign_targets_to_anchors + box_coder.encode + compute_loss for N images sharing one anchor set.
    `n_loss`: the losses are taken over the first n_loss images only (the sampler still draws for all N, in order)."""
    return rpn_loss_from_samples(rpn_targets_sample_batched(rpn, anchors0, gt, gvalid, n_loss), objectness, deltas)


def filter_proposals_padded(rpn, proposals, objectness, image_shape, num_anchors_per_level, deltas=None, anchors0=None):
    """filter_proposals without the per-image boolean selections: returns boxes [N,post,4] (first counts[i] rows valid,
    in decreasing-score order), scores [N,post], counts [N] (device int64).
    With `deltas` [N*A,4] and `anchors0` [A,4] (one anchor set shared by the images) instead of decoded `proposals`, only the
    pre-NMS top-k anchors are decoded: decode + clip + sigmoid + min-size / score tests in one launch
    (ops.rpn_decode_filter) instead of ~25 elementwise launches over all N*A anchors plus ~20 over the selection."""
    n_img = objectness.numel() // sum(num_anchors_per_level)
    device = objectness.device
    objectness = objectness.detach().reshape(n_img, -1)
    top = rpn._get_top_n_idx(objectness, num_anchors_per_level)
    k = rpn.pre_nms_top_n()
    lkey = (tuple(num_anchors_per_level), k, str(device))
    cache = rpn.__dict__.setdefault("_levels_of_top", {})
    if lkey not in cache:          # level id of every position of `top` (its layout depends on the level sizes only)
        cache[lkey] = torch.cat([torch.full((min(k, n),), i, dtype=torch.int64, device=device) for i, n in enumerate(num_anchors_per_level)], 0)
    levels = cache[lkey].reshape(1, -1).expand(n_img, -1)
    if deltas is not None and objectness.is_cuda and tuple(rpn.box_coder.weights) == (1.0, 1.0, 1.0, 1.0):
        boxes, prob, valid = ops.rpn_decode_filter(deltas.detach().reshape(n_img, -1, 4), objectness, anchors0, top, rpn.box_coder.bbox_xform_clip,
                                                   image_shape, rpn.min_size, rpn.score_thresh)
    else:
        if proposals is None:
            proposals = rpn.box_coder.decode_single(deltas.detach(), anchors0.repeat(n_img, 1)).reshape(n_img, -1, 4)
        bidx = torch.arange(n_img, device=device)[:, None]
        objectness, proposals = objectness[bidx, top], proposals[bidx, top]
        prob = torch.sigmoid(objectness)
        boxes = clip_boxes_to_image(proposals, image_shape)
        ws, hs = boxes[..., 2] - boxes[..., 0], boxes[..., 3] - boxes[..., 1]
        valid = (ws >= rpn.min_size) & (hs >= rpn.min_size) & (prob >= rpn.score_thresh)
    post = rpn.post_nms_top_n()
    # survivors in score order at the front; rows past counts[i] are padding
    seg = [min(k, n) for n in num_anchors_per_level]
    if _NMS_SEGMENTS and boxes.is_cuda and prob.dtype == torch.float32 and len(seg) <= 8 and sum(seg) == prob.shape[1]:
        # the levels are the NMS categories and `top` lists every level in descending score order: one scan per (image, level)
        pick, counts = _batched_nms_pick_segments(boxes, prob, seg, valid, rpn.nms_thresh, post)
    else:
        pick, counts = _batched_nms_pick(boxes, prob, levels, valid, rpn.nms_thresh, post)
    out_b = torch.gather(boxes, 1, pick[:, :, None].expand(-1, -1, 4))
    out_s = torch.gather(prob, 1, pick)
    return out_b, out_s, counts


def select_training_samples_batched(rh, props, pcounts, gt, glabels, gvalid):
    """RoIHeads.select_training_samples on padded proposals [N,Pm,4] (+counts): returns rois [R,5], labels [R],
    regression targets [R,4], per-image RoI counts (python ints)."""
    N, Pm, _ = props.shape
    G = gt.shape[1]
    dev = props.device
    pvalid = torch.arange(Pm, device=dev)[None, :] < pcounts[:, None]
    comb = torch.cat([props, gt], dim=1)                     # torchvision order: proposals, then GT boxes
    cvalid = torch.cat([pvalid, gvalid], dim=1)
    T = Pm + G
    # IoU + Matcher(0.5, 0.5) + class lookup (0 below the threshold and for GT-less images) in one launch
    m, lab, _ = ops.match_targets(gt, gvalid, glabels, comb, rh.proposal_matcher.high_threshold, rh.proposal_matcher.low_threshold, False)
    has_gt = gvalid.any(dim=1)
    lab = torch.where(cvalid, lab, torch.full_like(lab, -1))   # padding slots are neither positive nor negative
    pos_sel, neg_sel, picked = _sample_batched(rh.fg_bg_sampler, lab)
    per = [a + b for a, b in picked]
    R = sum(per)
    sel = _compact((pos_sel | neg_sel).reshape(-1), R)
    if comb.is_cuda:               # gathers + box_coder.encode + roi assembly in one launch (was ~40)
        rois, labels, reg_t = ops.roi_samples_finish(sel, comb, lab, m, gt, gvalid, rh.box_coder.weights)
        return rois, labels, reg_t, per
    img = torch.div(sel, T, rounding_mode="floor")
    boxes = comb.reshape(-1, 4)[sel]
    labels = lab.reshape(-1)[sel]
    midx = m.clamp(min=0).reshape(-1)[sel]
    matched_gt = gt[img, midx]
    matched_gt = torch.where(has_gt[img][:, None], matched_gt, torch.zeros_like(matched_gt))
    reg_t = rh.box_coder.encode_single(matched_gt, boxes)
    rois = torch.cat([img.to(boxes.dtype)[:, None], boxes], dim=1)
    return rois, labels, reg_t, per


def select_training_samples_padded(rh, props, pcounts, gt, glabels, gvalid):
    """select_training_samples_batched with a FIXED number of rows per image (S = batch_size_per_image): image i owns rows
    [i*S, (i+1)*S), its sampled RoIs first (same candidates, same order as the variable-size form), then padding rows (empty box,
    label -1, zero target).  Returns rois [N*S,5], labels [N*S], regression targets [N*S,4] and the per-image counts as a DEVICE tensor:
    no host synchronisation sizes anything downstream (RoI pooling, box head, losses, post-processing)."""
    N, Pm, _ = props.shape
    dev = props.device
    pvalid = _arange(Pm, dev)[None, :] < pcounts[:, None]
    comb = torch.cat([props, gt], dim=1)                     # torchvision order: proposals, then GT boxes
    cvalid = torch.cat([pvalid, gvalid], dim=1)
    T = comb.shape[1]
    m, lab, _ = ops.match_targets(gt, gvalid, glabels, comb, rh.proposal_matcher.high_threshold, rh.proposal_matcher.low_threshold, False)
    lab = torch.where(cvalid, lab, -1)
    pos_sel, neg_sel, _ = _sample_batched(rh.fg_bg_sampler, lab, host_counts=False)
    # compaction (r-th selected candidate of image n -> row n*S + r), padding rows, box targets and the counts in one launch
    return ops.roi_samples_padded(pos_sel, neg_sel, comb, lab, m, gt, gvalid, rh.fg_bg_sampler.batch_size_per_image, rh.box_coder.weights)


_LABEL_GRID = {}


def postprocess_detections_padded_rois(rh, class_logits, box_regression, rois, per_dev, S, image_shape):
    """postprocess_detections_flat for the fixed-size RoI list (rows [i*S, i*S + per_dev[i]) of image i are real): softmax, decode,
    clip and the candidate tests in one launch (hd_roi_postprocess), then the batched NMS."""
    device = class_logits.device
    num_classes = class_logits.shape[-1]
    n_img, K = per_dev.shape[0], num_classes - 1
    b, s, v = ops.roi_postprocess(class_logits.detach(), box_regression.detach(), rois, per_dev, S, rh.box_coder.weights,
                                  rh.box_coder.bbox_xform_clip, image_shape, rh.score_thresh)
    B, Sx, V = b.view(n_img, S * K, 4), s.view(n_img, S * K), v.view(n_img, S * K)
    key = (n_img, S, K, str(device))
    Lb = _LABEL_GRID.get(key)
    if Lb is None:                       # class id of every (row, class) slot: constant per shape
        Lb = _LABEL_GRID[key] = torch.arange(1, num_classes, device=device).view(1, 1, K).expand(n_img, S, K).reshape(n_img, S * K).contiguous()
    pick, counts = _batched_nms_pick(B, Sx, Lb, V, rh.nms_thresh, rh.detections_per_img)
    return torch.gather(B, 1, pick[:, :, None].expand(-1, -1, 4)), torch.gather(Sx, 1, pick), torch.gather(Lb, 1, pick), counts


def _need_cuda_rois(rois):
    if not rois.is_cuda:
        raise RuntimeError("hallucidet_amd: RoI pooling runs on the GPU only (RoIs on %s); there is no CPU path" % rois.device)


def roi_pool_rois(pool, feats_dict, rois, image_shape, n_images=None):
    feats = [v for k, v in feats_dict.items() if k in pool.featmap_names]
    device = rois.device
    scales = [pool.infer_scale((f.shape[1], f.shape[2]), image_shape) for f in feats]
    k_min, k_max = int(-math.log2(scales[0])), int(-math.log2(scales[-1]))
    rois = rois.float().contiguous()
    _need_cuda_rois(rois)
    levels = ops.roi_levels(rois, pool.canonical_scale, pool.canonical_level, pool.eps, k_min, k_max)      # LevelMapper in one launch
    acts = _active_views(feats, n_images) if (n_images is not None and n_images < feats[0].shape[0]) else None
    return _RoIAlignFn.apply(rois, levels, (scales, pool.output_size[0], pool.sampling_ratio, n_images), len(feats), *feats, *(acts or ()))


def fastrcnn_loss_flat(class_logits, box_regression, labels, regression_targets, n_valid=None):
    if not (class_logits.is_cuda and class_logits.dtype == torch.float32 and box_regression.dtype == torch.float32 and class_logits.shape[0] > 0
            and box_regression.shape[1] == 4 * class_logits.shape[1]):
        raise RuntimeError("hallucidet_amd: the Fast R-CNN loss runs in hd_fastrcnn_loss on fp32 CUDA predictor outputs with >= 1 RoI "
                           "(got logits %s %s on %s); there is no CPU path" % (tuple(class_logits.shape), class_logits.dtype, class_logits.device))
    return _FastRCNNLossFn.apply(class_logits, box_regression, labels, regression_targets, n_valid)


def postprocess_detections_flat(rh, class_logits, box_regression, rois, per, image_shape):
    """postprocess_detections from flat RoIs; returns per-image lists (one host sync for the detection counts)."""
    device = class_logits.device
    num_classes = class_logits.shape[-1]
    n_img, cap = len(per), max(max(per), 1)
    pred_scores = F.softmax(class_logits.detach(), -1)
    img = rois[:, 0].to(torch.int64)
    # position of every RoI inside its image (RoIs are grouped by image, in order): index minus first index of the image
    # (torch.bincount synchronises with the host to size its output: count by comparison instead)
    cnt = (img[None, :] == torch.arange(n_img, device=device)[:, None]).sum(dim=1)
    first = torch.cumsum(cnt, 0) - cnt
    slot = img * cap + (torch.arange(rois.shape[0], device=device) - first[img])
    K = num_classes - 1
    if box_regression.is_cuda:     # decode + clip in one launch (was ~35 elementwise launches)
        b = ops.roi_decode_clip(box_regression.detach(), rois, rh.box_coder.weights, rh.box_coder.bbox_xform_clip, image_shape)[:, 1:]
    else:
        pred_boxes = rh.box_coder.decode_single(box_regression.detach(), rois[:, 1:]).reshape(rois.shape[0], -1, 4)
        b = clip_boxes_to_image(pred_boxes, image_shape)[:, 1:]          # [R,K,4]
    s = pred_scores[:, 1:]
    B = torch.zeros((n_img * cap, K, 4), device=device)
    S = torch.zeros((n_img * cap, K), device=device)
    V = torch.zeros((n_img * cap, K), dtype=torch.bool, device=device)
    ws, hs = b[..., 2] - b[..., 0], b[..., 3] - b[..., 1]
    B[slot], S[slot], V[slot] = b, s, (s > rh.score_thresh) & (ws >= 1e-2) & (hs >= 1e-2)
    Lb = torch.arange(1, num_classes, device=device).view(1, 1, K).expand(n_img, cap, K).reshape(n_img, cap * K)
    B, S, V = B.view(n_img, cap * K, 4), S.view(n_img, cap * K), V.view(n_img, cap * K)
    pick, counts = _batched_nms_pick(B, S, Lb, V, rh.nms_thresh, rh.detections_per_img)
    sb = torch.gather(B, 1, pick[:, :, None].expand(-1, -1, 4))
    ss = torch.gather(S, 1, pick)
    sl = torch.gather(Lb, 1, pick)
    return sb, ss, sl, counts


class LazyDetections(list):
    """list[dict(boxes, labels, scores)] whose per-image slicing (one host sync for the detection counts) happens on
    first access: a training step never reads its detections (train_hallucidet.py:211-215 uses them in validation only),
    so the step stays free of that synchronisation.

    `deferred(thunk, n)`: the batched post-processing itself (score / top-k / decode / NMS: ~25 small launches, issue-bound on
    the host) is not launched where the reference's glue calls it but when the result is first touched or when the training step
    calls `flush()` -- right after it has issued the backward pass, so that those launches queue up behind several milliseconds of
    GPU work instead of starving the GPU between the RoI heads and the loss.  Still once per step, still inside the step."""

    def __init__(self, boxes, scores, labels, counts, box_fn=None):
        super().__init__()
        self._pad = (boxes, scores, labels, counts, box_fn)
        self._thunk = self._parent = None
        self._n = int(counts.shape[0])

    @classmethod
    def deferred(cls, thunk, n):
        """thunk() -> (boxes [n,D,4], scores, labels, counts [n], box_fn|None), evaluated under no_grad at flush()."""
        self = cls.__new__(cls)
        list.__init__(self)
        self._pad, self._thunk, self._parent, self._n = None, thunk, None, int(n)
        if not _DEFER_DETS and thunk is not None:
            self.flush()
        return self

    def flush(self):
        """Launch the deferred post-processing now (no host synchronisation)."""
        if self._parent is not None:
            par, lo, n = self._parent
            par.flush()
            if par._pad is None:                        # the parent was already sliced per image: nothing left to defer
                raise RuntimeError("LazyDetections: parent materialised before its split views were flushed")
            b, s, l, counts, fn = par._pad
            self._pad, self._parent = (b[lo:lo + n], s[lo:lo + n], l[lo:lo + n], counts[lo:lo + n], fn), None
        elif self._thunk is not None:
            thunk, self._thunk = self._thunk, None
            with torch.no_grad():
                self._pad = tuple(thunk())
        return self

    def _materialize(self):
        if self._thunk is not None or self._parent is not None:
            self.flush()
        if self._pad is not None:
            b, s, l, counts, fn = self._pad
            self._pad = None
            if fn is not None:
                b = fn(b)
            super().extend({"boxes": b[i, :c], "labels": l[i, :c], "scores": s[i, :c]} for i, c in enumerate(counts.tolist()))

    def __getitem__(self, i):
        self._materialize()
        return super().__getitem__(i)

    def __iter__(self):
        self._materialize()
        return super().__iter__()

    def __len__(self):
        return self._n if (self._pad is not None or self._thunk is not None or self._parent is not None) else super().__len__()

    def __add__(self, other):
        self._materialize()
        return list(self) + list(other)

    def split(self, sizes):
        """Per-pass views of a fused multi-pass result."""
        out, lo = [], 0
        if self._thunk is not None:
            for n in sizes:
                v = LazyDetections.deferred(None, n)
                v._parent = (self, lo, n)
                out.append(v)
                lo += n
            return out
        b, s, l, counts, fn = self._pad
        for n in sizes:
            out.append(LazyDetections(b[lo:lo + n], s[lo:lo + n], l[lo:lo + n], counts[lo:lo + n], fn))
            lo += n
        return out
