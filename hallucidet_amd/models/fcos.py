"""FCOS ResNet-50-FPN on the HIP kernels, with torchvision's attribute tree.

The reference builds `torchvision.models.detection.fcos_resnet50_fpn` (src/models/detector.py:135-136), re-heads
`head.classification_head.cls_logits` to `num_anchors * n_classes` outputs (N(0, 0.01) weights, bias -log(99), :57-66) and
drives it from src/utils/eval_forward_fcos.py:54-83 through `.transform`, `.backbone(x) -> OrderedDict('0','1','2','p6','p7')`,
`.head(features) -> {'cls_logits' [N, sum(HW), K], 'bbox_regression' [N, sum(HW), 4], 'bbox_ctrness' [N, sum(HW), 1]}`,
`.anchor_generator`, `.compute_loss(targets, head_outputs, anchors, num_anchors_per_level)`,
`.postprocess_detections(split_head_outputs, split_anchors, image_sizes)`.  This module provides that surface with the same
state_dict keys (`head.classification_head.conv.{0,1,3,4,6,7,9,10}.*`, `head.regression_head.bbox_ctrness.weight`, ...).

Execution: frozen detector -- the trunk is RetinaNet's (layer2-4 + FPN + P6/P7: same HIP convs); the two head towers are
4 x [implicit-GEMM conv3x3 (bias epilogue) -> hd_groupnorm8_relu]; backward = data gradients only (hd_groupnorm8_relu_bwd +
dgrad convs; with `set_trainable(True)` also parameter gradients: hd_wgrad for the convs, hd_groupnorm8_param_grad for the GroupNorm
affines).  Target assignment (hd_fcos_match), the three losses (hd_fcos_loss / _bwd) and NMS are HIP kernels; score / top-k / decode
of the post-processing are batched fp32 tensor ops on the GPU.  There is no CPU path.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import ops
from . import detection as D
from .detection import _conv_entry, _dgrad, _fwd


class BoxLinearCoder:
    """torchvision.models.detection._utils.BoxLinearCoder [EXT]: ltrb distances from the anchor centre, in anchor sizes."""

    def __init__(self, normalize_by_size=True):
        self.normalize_by_size = normalize_by_size

    def encode_single(self, reference_boxes, proposals):
        cx = 0.5 * (reference_boxes[:, 0] + reference_boxes[:, 2])
        cy = 0.5 * (reference_boxes[:, 1] + reference_boxes[:, 3])
        t = torch.stack((cx - proposals[:, 0], cy - proposals[:, 1], proposals[:, 2] - cx, proposals[:, 3] - cy), dim=1)
        if self.normalize_by_size:
            w = reference_boxes[:, 2] - reference_boxes[:, 0]
            h = reference_boxes[:, 3] - reference_boxes[:, 1]
            t = t / torch.stack((w, h, w, h), dim=1)
        return t

    def decode_single(self, rel_codes, boxes):
        boxes = boxes.to(rel_codes.dtype)
        cx = 0.5 * (boxes[:, 0] + boxes[:, 2])
        cy = 0.5 * (boxes[:, 1] + boxes[:, 3])
        if self.normalize_by_size:
            w = boxes[:, 2] - boxes[:, 0]
            h = boxes[:, 3] - boxes[:, 1]
            rel_codes = rel_codes * torch.stack((w, h, w, h), dim=1)
        return torch.stack((cx - rel_codes[:, 0], cy - rel_codes[:, 1], cx + rel_codes[:, 2], cy + rel_codes[:, 3]), dim=1)


class _HeadFn(torch.autograd.Function):
    """Both towers + the three output convs over every level.  Outputs per level: (cls, reg, ctr) NCHW fp32 (reg before its ReLU)."""

    @staticmethod
    def forward(ctx, hook, head, n_active, nlev, *feats):
        ctx.has_acts = len(feats) > nlev        # the [n_active] views the gradients are returned for (detection._active_views)
        ctx.set_materialize_grads(False)          # unused outputs arrive as None in backward, not as zero-filled maps
        feats = feats[:nlev]
        P = head.pack()
        # the 3x3 conv of layer k of BOTH towers on ALL levels is one grid (D._fwd_many / hd_conv2d_multi); GroupNorm + ReLU per tensor
        nl = len(feats)
        cur = list(feats) + list(feats)
        saved = [[[], []] for _ in range(nl)]
        for k in range(4):
            cs = D._fwd_many([P["cls_tower"][k][0]] * nl + [P["reg_tower"][k][0]] * nl, cur)          # conv + bias, NHWC f16
            nxt = []
            for idx, c in enumerate(cs):
                j, li = (0, idx) if idx < nl else (1, idx - nl)
                ga, be, eps = (P["cls_tower"] if j == 0 else P["reg_tower"])[k][1]
                z, stat = ops.groupnorm8_relu(c, ga, be, eps)
                saved[li][j].append((cur[idx][:n_active], c[:n_active], z[:n_active], stat[:n_active]))
                nxt.append(z)
            cur = nxt
        h1 = D._fwd_many([P["cls_out"]] * nl + [P["reg_out"]] * nl, cur, f32="nhwc")
        h2 = D._fwd_many([P["ctr_out"]] * nl, cur[nl:], f32="nhwc")
        outs = []
        for li in range(nl):
            outs += [h1[li], h1[nl + li], h2[li]]
        ctx.head, ctx.saved, ctx.na, ctx.n = head, saved, n_active, feats[0].shape[0]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        head = ctx.head
        P = head.pack()
        na = ctx.na
        tp, inv = head.train_params, 1.0 / head.grad_scale
        ch, rh = head.classification_head, head.regression_head
        dfeats = []
        for i, lv in enumerate(ctx.saved):
            df = None
            plan = ((P["cls_tower"], ((P["cls_out"], grads[3 * i], ch.cls_logits),), lv[0], ch),
                    (P["reg_tower"], ((P["reg_out"], grads[3 * i + 1], rh.bbox_reg), (P["ctr_out"], grads[3 * i + 2], rh.bbox_ctrness)), lv[1], rh))
            for tower, lasts, acts, mod in plan:
                H, W = acts[0][0].shape[1], acts[0][0].shape[2]
                d = None
                for last, g, conv in lasts:
                    if g is None:
                        continue
                    gl = D._head_grad_nhwc16(g[:na], H, W, last["cout_p"], last["wf"].dtype)
                    if tp:                               # the towers and their output convs are shared by the 5 levels: accumulate
                        D._wgrad_into(conv.weight, last, acts[3][2], gl, inv)
                        D._bgrad_into(conv.bias, gl, inv)
                    d = _dgrad(last, gl, (H, W), res=d)               # gradient w.r.t. the tower output (post-ReLU)
                if d is None:
                    continue
                convs = [l for l in mod.conv if isinstance(l, nn.Conv2d)]
                norms = [l for l in mod.conv if isinstance(l, nn.GroupNorm)]
                for k in (3, 2, 1, 0):
                    x_in, c, z, stat = acts[k]
                    dc = ops.groupnorm8_relu_bwd(d, c, z, tower[k][1][0], stat)
                    if tp:
                        ops.groupnorm8_param_grad(d, c, z, stat, norms[k].weight.grad, norms[k].bias.grad, inv)
                        D._wgrad_into(convs[k].weight, tower[k][0], x_in, dc, inv)
                        D._bgrad_into(convs[k].bias, dc, inv)
                    d = _dgrad(tower[k][0], dc, (H, W), res=df if k == 0 else None)
                df = d
            if df is not None and na < ctx.n and not ctx.has_acts:
                full = torch.zeros((ctx.n,) + tuple(df.shape[1:]), dtype=df.dtype, device=df.device)
                full[:na] = df
                df = full
            dfeats.append(df)
        ctx.saved = None
        if ctx.has_acts:
            return (None, None, None, None) + (None,) * len(dfeats) + tuple(dfeats)
        return (None, None, None, None) + tuple(dfeats)


class _GNTower(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        layers = []
        for _ in range(4):
            layers += [nn.Conv2d(in_channels, in_channels, 3, padding=1), nn.GroupNorm(32, in_channels), nn.ReLU()]
        self.conv = nn.Sequential(*layers)
        for l in self.conv.children():
            if isinstance(l, nn.Conv2d):
                nn.init.normal_(l.weight, std=0.01)
                nn.init.constant_(l.bias, 0)

    def tower_entries(self):
        convs = [l for l in self.conv if isinstance(l, nn.Conv2d)]
        norms = [l for l in self.conv if isinstance(l, nn.GroupNorm)]
        for n in norms:
            if n.num_channels != 8 * n.num_groups:
                raise NotImplementedError("hallucidet_amd: the FCOS towers use GroupNorm(32, 256) (8 channels per group); got %r" % (n,))
        return [(_conv_entry(c), (n.weight.detach().float().contiguous(), n.bias.detach().float().contiguous(), float(n.eps)))
                for c, n in zip(convs, norms)]


class FCOSClassificationHead(_GNTower):
    def __init__(self, in_channels, num_anchors, num_classes, prior_probability=0.01):
        super().__init__(in_channels)
        self.num_classes, self.num_anchors = num_classes, num_anchors
        self.cls_logits = nn.Conv2d(in_channels, num_anchors * num_classes, 3, padding=1)
        nn.init.normal_(self.cls_logits.weight, std=0.01)
        nn.init.constant_(self.cls_logits.bias, -math.log((1 - prior_probability) / prior_probability))


class FCOSRegressionHead(_GNTower):
    def __init__(self, in_channels, num_anchors):
        super().__init__(in_channels)
        self.bbox_reg = nn.Conv2d(in_channels, num_anchors * 4, 3, padding=1)
        self.bbox_ctrness = nn.Conv2d(in_channels, num_anchors * 1, 3, padding=1)
        for l in (self.bbox_reg, self.bbox_ctrness):
            nn.init.normal_(l.weight, std=0.01)
            nn.init.zeros_(l.bias)


class FCOSHead(nn.Module):
    def __init__(self, in_channels, num_anchors, num_classes):
        super().__init__()
        self.box_coder = BoxLinearCoder(normalize_by_size=True)
        self.classification_head = FCOSClassificationHead(in_channels, num_anchors, num_classes)
        self.regression_head = FCOSRegressionHead(in_channels, num_anchors)
        self._pack, self._hook = None, None
        self.train_params, self.grad_scale = False, 1.0

    def invalidate(self):
        self._pack = None

    def pack(self):
        if self._pack is None:
            c, r = self.classification_head, self.regression_head
            self._pack = dict(cls_tower=c.tower_entries(), cls_out=_conv_entry(c.cls_logits), reg_tower=r.tower_entries(),
                              reg_out=_conv_entry(r.bbox_reg), ctr_out=_conv_entry(r.bbox_ctrness))
        return self._pack

    def forward(self, x, n_active=None):
        """x: list of NHWC fp16 feature maps -> {'cls_logits' [N, sum(HW*A), K], 'bbox_regression' [N, sum(HW*A), 4] (>= 0),
        'bbox_ctrness' [N, sum(HW*A), 1]} fp32."""
        feats = list(x)
        if self._hook is None or self._hook.device != feats[0].device:
            self._hook = torch.zeros(1, device=feats[0].device, requires_grad=True)
        na = feats[0].shape[0] if n_active is None else n_active
        acts = D._active_views(feats, na) if na < feats[0].shape[0] else None
        outs = _HeadFn.apply(self._hook, self, na, len(feats), *feats, *(acts or ()))
        K = self.classification_head.cls_logits.out_channels // self.classification_head.num_anchors

        def flat(t, k):
            N, _, H, W = t.shape
            return t.view(N, -1, k, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, k)
        return {"cls_logits": torch.cat([flat(c, K) for c in outs[0::3]], dim=1),
                "bbox_regression": torch.relu(torch.cat([flat(r, 4) for r in outs[1::3]], dim=1)),
                "bbox_ctrness": torch.cat([flat(c, 1) for c in outs[2::3]], dim=1)}

    def compute_loss(self, targets, head_outputs, anchors, matched_idxs):
        """FCOSHead.compute_loss [EXT]: `matched_idxs` is the per-image list FCOS.compute_loss builds."""
        dev = head_outputs["cls_logits"].device
        gt, glab, _ = D.pad_targets(targets, dev)
        m = torch.stack([v.to(dev) for v in matched_idxs])
        return fcos_loss_batched(anchors[0], gt, glab, head_outputs, m)


def _default_anchorgen():
    return D.AnchorGenerator(((8,), (16,), (32,), (64,), (128,)), ((1.0,),) * 5)


class FCOS(nn.Module):
    def __init__(self, num_classes=91, min_size=800, max_size=1333, center_sampling_radius=1.5, score_thresh=0.2, nms_thresh=0.6,
                 detections_per_img=100, topk_candidates=1000):
        super().__init__()
        self.backbone = D.BackboneWithFPN(returned_layers=(2, 3, 4), extra_blocks=D.LastLevelP6P7(256, 256))
        self.anchor_generator = _default_anchorgen()
        self.head = FCOSHead(256, self.anchor_generator.num_anchors_per_location()[0], num_classes)
        self.box_coder = BoxLinearCoder(normalize_by_size=True)
        # torchvision's GeneralizedRCNNTransform is always replaced by the reference (detector.py:43-48); build that one.
        self.transform = D.CustomGeneralizedRCNNTransform(min_size=300, max_size=300, image_mean=[0.0], image_std=[1.0],
                                                          size_divisible=1, fixed_size=(300, 300))
        self.center_sampling_radius = center_sampling_radius
        self.score_thresh, self.nms_thresh = score_thresh, nms_thresh
        self.detections_per_img, self.topk_candidates = detections_per_img, topk_candidates
        self.batched_heads = True

    def invalidate_packs(self):
        self.backbone.invalidate()
        self.head.invalidate()

    def set_trainable(self, flag=True, grad_scale=1.0):
        """Detector fine-tuning switch (train_detector.py with detector_name='fcos'): parameter gradients for what torchvision's
        fcos_resnet50_fpn leaves trainable (trainable_backbone_layers=3 [EXT]: body.layer2-4, FPN incl. P6/P7, both head towers
        with their GroupNorm affines and the three output convs)."""
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
        """torchvision >= 0.13 key names -> the 0.12 names this module uses (FPN convs wrapped in Conv2dNormActivation)."""
        sd = OrderedDict()
        for k, v in state_dict.items():
            for i in range(3):
                k = k.replace("fpn.inner_blocks.%d.0." % i, "fpn.inner_blocks.%d." % i).replace("fpn.layer_blocks.%d.0." % i, "fpn.layer_blocks.%d." % i)
            if k.endswith("num_batches_tracked"):
                continue
            sd[k] = v
        return sd

    def load_state_dict(self, state_dict, strict=True):
        out = super().load_state_dict(self.remap_state_dict_keys(state_dict), strict=strict)
        self.invalidate_packs()
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.invalidate_packs()
        return out

    # ------------------------------------------------------------------ target assignment + losses
    def match_batched(self, anchors0, gt, gvalid, num_anchors_per_level):
        """FCOS.compute_loss's assignment [EXT] for B images sharing one location set -> matched [B, A] int64 (-1 = background)."""
        from .. import _abi
        B, G = gt.shape[0], gt.shape[1]
        A = anchors0.shape[0]
        an, g = anchors0.contiguous().float(), gt.contiguous().float()
        gv = gvalid.contiguous().to(torch.uint8)
        m = torch.empty((B, A), dtype=torch.int64, device=gt.device)
        _abi.check(_abi.load().hd_fcos_match(_abi.ptr(an), _abi.ptr(g), _abi.ptr(gv), B, A, G, int(num_anchors_per_level[0]),
                                             A - int(num_anchors_per_level[-1]), float(self.center_sampling_radius), _abi.ptr(m),
                                             torch.cuda.current_stream().cuda_stream), "hd_fcos_match")
        return m

    def compute_loss(self, targets, head_outputs, anchors, num_anchors_per_level):
        """torchvision FCOS.compute_loss signature (called by eval_forward_fcos.py:70)."""
        dev = head_outputs["cls_logits"].device
        gt, glab, gvalid = D.pad_targets(targets, dev)
        m = self.match_batched(anchors[0], gt, gvalid, num_anchors_per_level)
        return fcos_loss_batched(anchors[0], gt, glab, head_outputs, m)

    # ------------------------------------------------------------------ reference-shaped (list based) post-processing
    def postprocess_detections(self, head_outputs, anchors, image_shapes):
        """torchvision FCOS.postprocess_detections [EXT]: per image, per level: score = sqrt(sigmoid(cls) * sigmoid(ctr)) >
        score_thresh, top-k (1000) candidates, decode + clip; then class-aware NMS (0.6) and the first detections_per_img."""
        class_logits, box_regression, box_ctrness = head_outputs["cls_logits"], head_outputs["bbox_regression"], head_outputs["bbox_ctrness"]
        detections = []
        for index in range(len(image_shapes)):
            ib, is_, il = [], [], []
            for breg, logits, ctr, anc in zip((b[index] for b in box_regression), (c[index] for c in class_logits),
                                              (c[index] for c in box_ctrness), anchors[index]):
                num_classes = logits.shape[-1]
                scores = torch.sqrt(torch.sigmoid(logits.detach()) * torch.sigmoid(ctr.detach())).flatten()
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
    def postprocess_detections_padded(self, head_outputs, anchors0, napl, image_shape):
        """Same selection on padded tensors.  Returns boxes [B, D, 4], scores [B, D], labels [B, D], counts [B]."""
        cls_logits, bbox_regression, ctr = head_outputs["cls_logits"], head_outputs["bbox_regression"], head_outputs["bbox_ctrness"]
        B, A, K = cls_logits.shape
        # all levels at once (as RetinaNet.postprocess_detections_padded): threshold, per-level top-k in one hd_topk_select_rows launch,
        # one gather / decode / clip over the selected candidates
        scores = torch.sqrt(torch.sigmoid(cls_logits.detach()) * torch.sigmoid(ctr.detach())).reshape(B, A * K)
        key = torch.where(scores > self.score_thresh, scores, float("-inf"))
        idx = ops.topk_rows_segments(key, [n * K for n in napl], self.topk_candidates)
        sc = torch.gather(key, 1, idx)
        valid = sc > float("-inf")
        aidx = torch.div(idx, K, rounding_mode="floor")
        breg = torch.gather(bbox_regression.detach(), 1, aidx[:, :, None].expand(-1, -1, 4))
        boxes = self.box_coder.decode_single(breg.reshape(-1, 4), anchors0[aidx].reshape(-1, 4)).reshape(B, -1, 4)
        cb = D.clip_boxes_to_image(boxes, image_shape)
        cs = torch.where(valid, sc, 0.0)
        cl = idx % K
        pick, counts = D._batched_nms_pick(cb, cs, cl, valid, self.nms_thresh, self.detections_per_img)
        return (torch.gather(cb, 1, pick[:, :, None].expand(-1, -1, 4)), torch.gather(cs, 1, pick), torch.gather(cl, 1, pick), counts)


def fcos_resnet50_fpn(pretrained=False, progress=True, num_classes=91, pretrained_backbone=False, weights_path=None, **kwargs):
    """torchvision.models.detection.fcos_resnet50_fpn [EXT].  COCO weights cannot be downloaded offline: `pretrained=True` is
    accepted for signature compatibility and ignored unless `weights_path` points at a local torchvision state_dict."""
    model = FCOS(num_classes=num_classes, **kwargs)
    if weights_path is not None:
        model.load_state_dict(torch.load(weights_path, map_location="cpu"))
    return model


# ======================================================================================================================
# batched losses (FCOSHead.compute_loss [EXT] without the per-image loops)
# ======================================================================================================================
class _FcosLossFn(torch.autograd.Function):
    """The three FCOS losses of a batch as one forward (+ finish) and one backward launch (hd_fcos_loss / _bwd)."""

    @staticmethod
    def forward(ctx, cls_logits, bbox_regression, bbox_ctrness, matched, gt, glab, anchors0, alpha, gamma):
        from .. import _abi
        lib = _abi.load()
        lg, br = cls_logits.detach().contiguous().float(), bbox_regression.detach().contiguous().float()
        ct = bbox_ctrness.detach().contiguous().float()
        m, g, gl, an = matched.contiguous().to(torch.int64), gt.contiguous().float(), glab.contiguous().to(torch.int64), anchors0.contiguous().float()
        B, A, K = lg.shape
        dev = lg.device
        ws = torch.empty(64 * 4, dtype=torch.float32, device=dev)
        nfg = torch.empty(1, dtype=torch.float32, device=dev)
        out = torch.empty(3, dtype=torch.float32, device=dev)
        _abi.check(lib.hd_fcos_loss(_abi.ptr(lg), _abi.ptr(br), _abi.ptr(ct), _abi.ptr(m), _abi.ptr(g), _abi.ptr(gl), _abi.ptr(an), B, A, K, g.shape[1],
                                    alpha, gamma, _abi.ptr(ws), _abi.ptr(nfg), _abi.ptr(out), torch.cuda.current_stream().cuda_stream), "hd_fcos_loss")
        ctx.save_for_backward(lg, br, ct, m, g, gl, an, nfg)
        ctx.consts = (alpha, gamma)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g_cls, g_reg, g_ctr):
        from .. import _abi
        lib = _abi.load()
        lg, br, ct, m, g, gl, an, nfg = ctx.saved_tensors
        alpha, gamma = ctx.consts
        B, A, K = lg.shape
        z = torch.zeros((), dtype=torch.float32, device=lg.device)
        g3 = torch.stack([z if t is None else t.reshape(()).float() for t in (g_cls, g_reg, g_ctr)]).contiguous()
        d_lg, d_br, d_ct = torch.empty_like(lg), torch.empty_like(br), torch.empty_like(ct)
        _abi.check(lib.hd_fcos_loss_bwd(_abi.ptr(lg), _abi.ptr(br), _abi.ptr(ct), _abi.ptr(m), _abi.ptr(g), _abi.ptr(gl), _abi.ptr(an), B, A, K,
                                        g.shape[1], alpha, gamma, _abi.ptr(nfg), _abi.ptr(g3), _abi.ptr(d_lg), _abi.ptr(d_br), _abi.ptr(d_ct),
                                        torch.cuda.current_stream().cuda_stream), "hd_fcos_loss_bwd")
        return d_lg, d_br, d_ct, None, None, None, None, None, None


def fcos_loss_batched(anchors0, gt, glab, head_outputs, matched, alpha=0.25, gamma=2.0):
    """{'classification', 'bbox_regression', 'bbox_ctrness'} for B images sharing one location set."""
    cls_logits = head_outputs["cls_logits"]
    ctr = head_outputs["bbox_ctrness"]
    ctr = ctr.reshape(ctr.shape[0], ctr.shape[1])
    c, r, t = _FcosLossFn.apply(cls_logits, head_outputs["bbox_regression"], ctr, matched, gt, glab, anchors0, alpha, gamma)
    return {"classification": c, "bbox_regression": r, "bbox_ctrness": t}
