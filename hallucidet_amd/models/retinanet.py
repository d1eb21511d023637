"""RetinaNet ResNet-50-FPN (BASELINE config 4) on the HIP kernels, with torchvision 0.12's attribute tree.

The reference builds `torchvision.models.detection.retinanet_resnet50_fpn` (src/models/detector.py:137-139), re-heads
`head.classification_head.cls_logits` to `num_anchors * n_classes` outputs (N(0, 0.01) weights, bias -log(99), :57-66)
and drives it from src/utils/eval_forward_retinanet.py:83-160 through `.transform`, `.backbone(x) -> OrderedDict('0','1',
'2','p6','p7')`, `.head(features) -> {'cls_logits' [N, sum(HWA), K], 'bbox_regression' [N, sum(HWA), 4]}`,
`.anchor_generator`, `.proposal_matcher`, `.box_coder`, `.head.classification_head.BETWEEN_THRESHOLDS`,
`.postprocess_detections(split_head_outputs, split_anchors, image_sizes)`.  This module provides that surface with the
same state_dict keys (`backbone.fpn.extra_blocks.p6.weight`, `head.classification_head.conv.0.weight`,
`head.regression_head.bbox_reg.bias`, ...).

Execution: frozen detector -- FrozenBN folded into fp16 GEMM-layout weights, implicit-GEMM convs with bias/ReLU
epilogues forward, data gradients only backward (ReLU masks fused into the dgrad epilogue); the 4-conv towers of the
two heads share weights across the 5 pyramid levels.  Matching, focal loss, smooth-L1 and post-processing are batched
fp32 tensor ops on the GPU (IoU and NMS are HIP kernels); there is no CPU path.
"""
import math
import re
from collections import OrderedDict

import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..ops import ACT_RELU
from . import detection as D
from .detection import _conv_entry, _dgrad, _fwd


class _HeadFn(torch.autograd.Function):
    """Both towers + output convs over every level.  Outputs: (cls, reg) NCHW fp32 per level."""

    @staticmethod
    def forward(ctx, hook, head, n_active, nlev, *feats):
        ctx.has_acts = len(feats) > nlev        # the [n_active] views the gradients are returned for (detection._active_views)
        ctx.set_materialize_grads(False)          # unused outputs arrive as None in backward, not as zero-filled maps
        feats = feats[:nlev]
        P = head.pack()
        # layer k of BOTH towers on ALL levels is one grid (2 x levels independent 3x3 convs; D._fwd_many / hd_conv2d_multi): 5 launches
        # for the head instead of 50, the 10x10 / 5x5 levels riding in the tail of the 38x38 level's tiles
        nl = len(feats)
        cur = list(feats) + list(feats)
        saved = [[[], []] for _ in range(nl)]
        for k in range(4):
            cur = D._fwd_many([P["cls_tower"][k]] * nl + [P["reg_tower"][k]] * nl, cur, act=ACT_RELU)
            for li in range(nl):
                saved[li][0].append(cur[li][:n_active])
                saved[li][1].append(cur[nl + li][:n_active])
        heads = D._fwd_many([P["cls_out"]] * nl + [P["reg_out"]] * nl, cur, f32="nhwc")
        outs = []
        for li in range(nl):
            outs.append(heads[li])
            outs.append(heads[nl + li])
        ctx.head, ctx.saved, ctx.na, ctx.n = head, saved, n_active, feats[0].shape[0]
        ctx.feats = [f[:n_active] for f in feats] if head.train_params else None
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        head = ctx.head
        P = head.pack()
        na = ctx.na
        tp, inv = head.train_params, 1.0 / head.grad_scale
        mods = ((head.classification_head, head.classification_head.cls_logits), (head.regression_head, head.regression_head.bbox_reg))
        dfeats = []
        nl = len(ctx.saved)
        if not tp and all(g is not None for g in grads[:2 * nl]) and (na == ctx.n or ctx.has_acts):
            # frozen head, every output has a gradient (the training step): the data-gradient conv of layer k of both towers on all
            # levels as one grid, walking the towers backwards; the two towers' first layers meet in the residual of the second
            hws = [(lv[0][0].shape[1], lv[0][0].shape[2]) for lv in ctx.saved]
            d = [D._head_grad_nhwc16(grads[2 * i][:na], hws[i][0], hws[i][1], P["cls_out"]["cout_p"], P["cls_out"]["wf"].dtype) for i in range(nl)] + \
                [D._head_grad_nhwc16(grads[2 * i + 1][:na], hws[i][0], hws[i][1], P["reg_out"]["cout_p"], P["reg_out"]["wf"].dtype) for i in range(nl)]
            acts = lambda k: [ctx.saved[i][0][k] for i in range(nl)] + [ctx.saved[i][1][k] for i in range(nl)]
            d = D._dgrad_many([P["cls_out"]] * nl + [P["reg_out"]] * nl, d, hws + hws, masks=acts(3))
            for k in (3, 2, 1):
                d = D._dgrad_many([P["cls_tower"][k]] * nl + [P["reg_tower"][k]] * nl, d, hws + hws, masks=acts(k - 1))
            df = D._dgrad_many([P["cls_tower"][0]] * nl, d[:nl], hws)
            dfeats = D._dgrad_many([P["reg_tower"][0]] * nl, d[nl:], hws, ress=df)
            ctx.saved = None
            if ctx.has_acts:
                return (None, None, None, None) + (None,) * len(dfeats) + tuple(dfeats)
            return (None, None, None, None) + tuple(dfeats)
        for i, lv in enumerate(ctx.saved):
            df = None
            for j, (tower, last) in enumerate(((P["cls_tower"], P["cls_out"]), (P["reg_tower"], P["reg_out"]))):
                g = grads[2 * i + j]
                if g is None:
                    continue
                acts = lv[j]
                H, W = acts[0].shape[1], acts[0].shape[2]
                convs = [l for l in mods[j][0].conv if isinstance(l, nn.Conv2d)]
                d = D._head_grad_nhwc16(g[:na], H, W, last["cout_p"], last["wf"].dtype)
                if tp:                                   # the towers share their weights over the 5 levels: accumulate
                    D._wgrad_into(mods[j][1].weight, last, acts[3], d, inv)
                    D._bgrad_into(mods[j][1].bias, d, inv)
                d = _dgrad(last, d, (H, W), mask=acts[3])
                for k in (3, 2, 1):
                    if tp:
                        D._wgrad_into(convs[k].weight, tower[k], acts[k - 1], d, inv)
                        D._bgrad_into(convs[k].bias, d, inv)
                    d = _dgrad(tower[k], d, (H, W), mask=acts[k - 1])
                if tp:
                    D._wgrad_into(convs[0].weight, tower[0], ctx.feats[i], d, inv)
                    D._bgrad_into(convs[0].bias, d, inv)
                df = _dgrad(tower[0], d, (H, W), res=df)
            if df is not None and na < ctx.n and not ctx.has_acts:
                full = torch.zeros((ctx.n,) + tuple(df.shape[1:]), dtype=df.dtype, device=df.device)
                full[:na] = df
                df = full
            dfeats.append(df)
        ctx.saved = None
        if ctx.has_acts:
            return (None, None, None, None) + (None,) * len(dfeats) + tuple(dfeats)
        return (None, None, None, None) + tuple(dfeats)


class _Tower(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        layers = []
        for _ in range(4):
            layers += [nn.Conv2d(in_channels, in_channels, 3, padding=1), nn.ReLU()]
        self.conv = nn.Sequential(*layers)
        for l in self.conv.children():
            if isinstance(l, nn.Conv2d):
                nn.init.normal_(l.weight, std=0.01)
                nn.init.constant_(l.bias, 0)

    def tower_entries(self):
        return [_conv_entry(l) for l in self.conv if isinstance(l, nn.Conv2d)]


class RetinaNetClassificationHead(_Tower):
    BETWEEN_THRESHOLDS = D.Matcher.BETWEEN_THRESHOLDS

    def __init__(self, in_channels, num_anchors, num_classes, prior_probability=0.01):
        super().__init__(in_channels)
        self.cls_logits = nn.Conv2d(in_channels, num_anchors * num_classes, 3, padding=1)
        nn.init.normal_(self.cls_logits.weight, std=0.01)
        nn.init.constant_(self.cls_logits.bias, -math.log((1 - prior_probability) / prior_probability))
        self.num_classes, self.num_anchors = num_classes, num_anchors


class RetinaNetRegressionHead(_Tower):
    def __init__(self, in_channels, num_anchors):
        super().__init__(in_channels)
        self.bbox_reg = nn.Conv2d(in_channels, num_anchors * 4, 3, padding=1)
        nn.init.normal_(self.bbox_reg.weight, std=0.01)
        nn.init.zeros_(self.bbox_reg.bias)


class RetinaNetHead(nn.Module):
    def __init__(self, in_channels, num_anchors, num_classes):
        super().__init__()
        self.classification_head = RetinaNetClassificationHead(in_channels, num_anchors, num_classes)
        self.regression_head = RetinaNetRegressionHead(in_channels, num_anchors)
        self._pack, self._hook = None, None
        self.train_params, self.grad_scale = False, 1.0

    def invalidate(self):
        self._pack = None

    def pack(self):
        if self._pack is None:
            c, r = self.classification_head, self.regression_head
            self._pack = dict(cls_tower=c.tower_entries(), cls_out=_conv_entry(c.cls_logits),
                              reg_tower=r.tower_entries(), reg_out=_conv_entry(r.bbox_reg))
        return self._pack

    def forward(self, x, n_active=None):
        """x: list of NHWC fp16 feature maps -> {'cls_logits': [N, sum(H*W*A), K], 'bbox_regression': [N, sum(H*W*A), 4]} fp32."""
        feats = list(x)
        if self._hook is None or self._hook.device != feats[0].device:
            self._hook = torch.zeros(1, device=feats[0].device, requires_grad=True)
        na = feats[0].shape[0] if n_active is None else n_active
        acts = D._active_views(feats, na) if na < feats[0].shape[0] else None
        outs = _HeadFn.apply(self._hook, self, na, len(feats), *feats, *(acts or ()))
        K = self.classification_head.cls_logits.out_channels // self.classification_head.num_anchors
        cls, reg = [], []
        for c, r in zip(outs[0::2], outs[1::2]):
            N, _, H, W = c.shape
            cls.append(c.view(N, -1, K, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, K))
            reg.append(r.view(N, -1, 4, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, 4))
        return {"cls_logits": torch.cat(cls, dim=1), "bbox_regression": torch.cat(reg, dim=1)}


def _default_anchorgen():
    sizes = tuple((x, int(x * 2 ** (1.0 / 3)), int(x * 2 ** (2.0 / 3))) for x in [32, 64, 128, 256, 512])
    return D.AnchorGenerator(sizes, ((0.5, 1.0, 2.0),) * len(sizes))


class RetinaNet(nn.Module):
    def __init__(self, num_classes=91, min_size=800, max_size=1333, score_thresh=0.05, nms_thresh=0.5, detections_per_img=300,
                 fg_iou_thresh=0.5, bg_iou_thresh=0.4, topk_candidates=1000):
        super().__init__()
        self.backbone = D.BackboneWithFPN(returned_layers=(2, 3, 4), extra_blocks=D.LastLevelP6P7(256, 256))
        self.anchor_generator = _default_anchorgen()
        self.head = RetinaNetHead(256, self.anchor_generator.num_anchors_per_location()[0], num_classes)
        self.proposal_matcher = D.Matcher(fg_iou_thresh, bg_iou_thresh, allow_low_quality_matches=True)
        self.box_coder = D.BoxCoder((1.0, 1.0, 1.0, 1.0))
        # torchvision's GeneralizedRCNNTransform is always replaced by the reference (detector.py:43-48); build that one.
        self.transform = D.CustomGeneralizedRCNNTransform(min_size=300, max_size=300, image_mean=[0.0], image_std=[1.0],
                                                          size_divisible=1, fixed_size=(300, 300))
        self.score_thresh, self.nms_thresh = score_thresh, nms_thresh
        self.detections_per_img, self.topk_candidates = detections_per_img, topk_candidates
        self.batched_heads = True

    def invalidate_packs(self):
        self.backbone.invalidate()
        self.head.invalidate()

    def set_trainable(self, flag=True, grad_scale=1.0):
        """Detector fine-tuning switch (train_detector.py with detector_name='retinanet'): parameter gradients for what
        torchvision's retinanet_resnet50_fpn leaves trainable (trainable_backbone_layers=3 [EXT]: body.layer2-4, FPN incl.
        P6/P7, both head towers and their output convs)."""
        for m in (self.backbone, self.head):
            m.train_params, m.grad_scale = bool(flag), float(grad_scale)
        for name, p in self.backbone.body.named_parameters():
            p.requires_grad_(bool(flag) and name.split(".")[0] in ("layer2", "layer3", "layer4"))
        for mod in (self.backbone.fpn, self.head):
            for p in mod.parameters():
                p.requires_grad_(bool(flag))

    def trainable_parameters(self):
        return [p for p in self.parameters() if p.requires_grad]

    @staticmethod
    def remap_state_dict_keys(state_dict):
        """torchvision >= 0.13 key names -> the 0.12 names this module (and the reference's checkpoints) use."""
        sd = OrderedDict()
        for k, v in state_dict.items():
            for i in range(3):                       # torchvision >= 0.13 wraps the FPN convs in Conv2dNormActivation
                k = k.replace("fpn.inner_blocks.%d.0." % i, "fpn.inner_blocks.%d." % i).replace("fpn.layer_blocks.%d.0." % i, "fpn.layer_blocks.%d." % i)
            # ... and the tower convs: head.<h>.conv.<i>.0.weight -> head.<h>.conv.<2i>.weight
            k = re.sub(r"(head\.\w+_head\.conv\.)(\d)\.0\.", lambda m: "%s%d." % (m.group(1), 2 * int(m.group(2))), k)
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
        out = super()._apply(fn, *a, **k)
        self.invalidate_packs()
        return out

    # ------------------------------------------------------------------ reference-shaped (list based) post-processing
    def postprocess_detections(self, head_outputs, anchors, image_shapes):
        """torchvision 0.12 RetinaNet.postprocess_detections [EXT]: per image, per level: sigmoid, > score_thresh, top-k
        (1000) candidates, decode + clip; then class-aware NMS (0.5) and the first detections_per_img."""
        class_logits, box_regression = head_outputs["cls_logits"], head_outputs["bbox_regression"]
        detections = []
        for index in range(len(image_shapes)):
            ib, is_, il = [], [], []
            for breg, logits, anc in zip((b[index] for b in box_regression), (c[index] for c in class_logits), anchors[index]):
                num_classes = logits.shape[-1]
                scores = torch.sigmoid(logits.detach()).flatten()
                keep = scores > self.score_thresh
                scores, topk_idxs = scores[keep], torch.where(keep)[0]
                num_topk = min(self.topk_candidates, topk_idxs.size(0))
                order = torch.sort(scores, descending=True, stable=True)[1][:num_topk]
                scores, topk_idxs = scores[order], topk_idxs[order]
                anchor_idxs = torch.div(topk_idxs, num_classes, rounding_mode="floor")
                boxes = self.box_coder.decode_single(breg.detach()[anchor_idxs], anc[anchor_idxs])
                ib.append(D.clip_boxes_to_image(boxes, image_shapes[index]))
                is_.append(scores)
                il.append(topk_idxs % num_classes)
            ib, is_, il = torch.cat(ib, 0), torch.cat(is_, 0), torch.cat(il, 0)
            order, sel, _ = D._batched_nms_padded(ib[None], is_[None], il[None], torch.ones_like(is_[None], dtype=torch.bool),
                                                  self.nms_thresh, self.detections_per_img)
            keep = order[0][sel[0]]
            detections.append({"boxes": ib[keep], "scores": is_[keep], "labels": il[keep]})
        return detections

    # ------------------------------------------------------------------ batched form (no per-image / per-level host syncs)
    def postprocess_detections_padded(self, cls_logits, bbox_regression, anchors0, napl, image_shape):
        """Same selection on padded tensors.  cls_logits [B, A, K], bbox_regression [B, A, 4], anchors0 [A, 4], napl =
        anchors per level.  Returns boxes [B, D, 4], scores [B, D], labels [B, D], counts [B] (D = detections_per_img)."""
        B, A, K = cls_logits.shape
        # every level's score > thresh / top-k (1000) in ONE launch: hd_topk_select_rows over the level segments of the [B, A*K] score
        # rows returns, per segment, the row indices of its largest entries in descending order, ties by ascending index -- what the
        # per-level `sort(descending, stable)[:k]` of the list form gives; candidates at or below the threshold carry -inf.
        scores = torch.sigmoid(cls_logits.detach().reshape(B, A * K))
        key = torch.where(scores > self.score_thresh, scores, float("-inf"))
        idx = ops.topk_rows_segments(key, [n * K for n in napl], self.topk_candidates)          # [B, sum(min(k, n*K))]
        sc = torch.gather(key, 1, idx)
        valid = sc > float("-inf")
        aidx = torch.div(idx, K, rounding_mode="floor")                                           # anchor index within the image
        breg = torch.gather(bbox_regression.detach(), 1, aidx[:, :, None].expand(-1, -1, 4))
        anc = anchors0[aidx]
        # BoxCoder.decode_single + clip_boxes_to_image of the selected candidates in one launch (was ~25 elementwise launches per level)
        cb = ops.roi_decode_clip(breg.reshape(-1, 4), anc.reshape(-1, 4), self.box_coder.weights, self.box_coder.bbox_xform_clip,
                                 image_shape).reshape(B, -1, 4)
        cs = torch.where(valid, sc, 0.0)
        cl = idx % K
        pick, counts = D._batched_nms_pick(cb, cs, cl, valid, self.nms_thresh, self.detections_per_img)
        return (torch.gather(cb, 1, pick[:, :, None].expand(-1, -1, 4)), torch.gather(cs, 1, pick), torch.gather(cl, 1, pick), counts)


def retinanet_resnet50_fpn(pretrained=False, progress=True, num_classes=91, pretrained_backbone=False, weights_path=None, **kwargs):
    """torchvision.models.detection.retinanet_resnet50_fpn [EXT].  COCO weights cannot be downloaded offline:
    `pretrained=True` is accepted for signature compatibility and ignored unless `weights_path` points at a local
    torchvision state_dict (same policy as fasterrcnn_resnet50_fpn)."""
    model = RetinaNet(num_classes=num_classes, **kwargs)
    if weights_path is not None:
        model.load_state_dict(torch.load(weights_path, map_location="cpu"))
    return model


# ======================================================================================================================
# batched losses (eval_forward_retinanet.py:163-244 without the per-image loops)
# ======================================================================================================================
def retinanet_match_batched(model, anchors0, gt, gvalid):
    """:165-175: box_iou + proposal_matcher per image; GT-less images are all -1.  -> matched idx [B, A]."""
    pm = model.proposal_matcher
    return ops.match_targets(gt, gvalid, None, anchors0, pm.high_threshold, pm.low_threshold, pm.allow_low_quality_matches, want_labels=False)[0]


class _RetinaNetLossFn(torch.autograd.Function):
    """Both RetinaNet losses of a batch as one forward (+ finish) and one backward launch (hd_retinanet_loss / _bwd): focal
    loss over the counted anchors, smooth-L1 against box targets that are encoded in-kernel from the matched ground truth."""

    @staticmethod
    def forward(ctx, cls_logits, bbox_regression, matched, gt, glab, anchors0, coder_weights, alpha, gamma, beta):
        from .. import _abi
        lib = _abi.load()
        lg, br = cls_logits.detach().contiguous().float(), bbox_regression.detach().contiguous().float()
        m, g, gl, an = matched.contiguous().to(torch.int64), gt.contiguous().float(), glab.contiguous().to(torch.int64), anchors0.contiguous().float()
        B, A, K = lg.shape
        G = g.shape[1]
        dev = lg.device
        ws = torch.empty(B * 32 * 3, dtype=torch.float32, device=dev)
        nfg = torch.empty(B, dtype=torch.float32, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)
        cw = (ctypes.c_float * 4)(*[float(w) for w in coder_weights])
        _abi.check(lib.hd_retinanet_loss(_abi.ptr(lg), _abi.ptr(br), _abi.ptr(m), _abi.ptr(g), _abi.ptr(gl), _abi.ptr(an), B, A, K, G, alpha, gamma, beta,
                                         cw, _abi.ptr(ws), _abi.ptr(nfg), _abi.ptr(out), torch.cuda.current_stream().cuda_stream), "hd_retinanet_loss")
        ctx.save_for_backward(lg, br, m, g, gl, an, nfg)
        ctx.consts = (cw, alpha, gamma, beta)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_cls, g_reg):
        from .. import _abi
        lib = _abi.load()
        lg, br, m, g, gl, an, nfg = ctx.saved_tensors
        cw, alpha, gamma, beta = ctx.consts
        B, A, K = lg.shape
        d_lg, d_br = torch.empty_like(lg), torch.empty_like(br)
        gc = None if g_cls is None else g_cls.contiguous().float()
        gr = None if g_reg is None else g_reg.contiguous().float()
        _abi.check(lib.hd_retinanet_loss_bwd(_abi.ptr(lg), _abi.ptr(br), _abi.ptr(m), _abi.ptr(g), _abi.ptr(gl), _abi.ptr(an), B, A, K, g.shape[1], alpha,
                                             gamma, beta, cw, _abi.ptr(nfg), _abi.ptr(gc), _abi.ptr(gr), _abi.ptr(d_lg), _abi.ptr(d_br),
                                             torch.cuda.current_stream().cuda_stream), "hd_retinanet_loss_bwd")
        return d_lg, d_br, None, None, None, None, None, None, None, None


def retinanet_loss_batched(model, anchors0, gt, glab, gvalid, cls_logits, bbox_regression, matched=None, alpha=0.25, gamma=2.0, beta=1.0):
    """compute_retinanet_loss (eval_forward_retinanet.py:163-244) for B images sharing one anchor set: focal loss (alpha .25,
    gamma 2) over the non-ignored anchors and smooth-L1 (beta 1) over the foreground anchors, each / max(1, num_foreground) per
    image, averaged over images -- two HIP launches forward, one backward."""
    m = retinanet_match_batched(model, anchors0, gt, gvalid) if matched is None else matched
    if gt.shape[1] == 0:                                  # no ground truth anywhere: one dummy (never matched) target row
        gt = torch.zeros((gt.shape[0], 1, 4), dtype=torch.float32, device=cls_logits.device)
        glab = torch.zeros((gt.shape[0], 1), dtype=torch.int64, device=cls_logits.device)
    cls_loss, reg_loss = _RetinaNetLossFn.apply(cls_logits, bbox_regression, m, gt, glab, anchors0, tuple(model.box_coder.weights), alpha, gamma, beta)
    return {"classification": cls_loss, "bbox_regression": reg_loss}
