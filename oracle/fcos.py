"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain torch fp32) of the FCOS side of the hot path.

Follows
  * reference tree: src/models/detector.py:57-66 (re-heading of cls_logits to n_classes: N(0, 0.01) weights, bias -log(99)),
    :113-114,135-136 (selection), src/utils/eval_forward_fcos.py:11-83 (eval_forward_fcos: transform -> backbone -> head ->
    anchor_generator -> model.compute_loss -> split per level -> model.postprocess_detections -> transform.postprocess);
  * un-vendored torchvision `fcos_resnet50_fpn` (torchvision.models.detection.fcos, 0.12+) [EXT], restated from its published
    algorithm: ResNet-50 body returning layer2-4, FPN over (512, 1024, 2048) with LastLevelP6P7(256, 256); FCOSHead = two
    towers of 4 x [conv3x3, GroupNorm(32), ReLU] -> cls_logits conv3x3 | bbox_reg conv3x3 + ReLU and bbox_ctrness conv3x3 (the
    centre-ness branch hangs off the REGRESSION tower); AnchorGenerator sizes ((8,), (16,), (32,), (64,), (128,)) x ratio 1
    (one stride-sized square per location; only its centre and size are used); BoxLinearCoder(normalize_by_size=True);
    FCOS.compute_loss (centre sampling radius 1.5, location inside the box, scale range (4, 8) x size with 0 / inf at the
    ends, smallest box wins), FCOSHead.compute_loss (sigmoid focal sum, generalized-IoU sum over decoded foreground boxes,
    centre-ness BCE sum, each / max(1, #foreground of the batch)), postprocess (score = sqrt(sigmoid(cls) * sigmoid(ctr)) >
    0.2, top-1000 per level, decode + clip, class-aware NMS 0.6, 100 detections).
    PARITY UNPINNED against torchvision itself (absent from this image); the ORCHESTRATION is pinned by driving the
    reference's own eval_forward_fcos.py over this object (tests/golden/make_golden.py: glue_fcos.npz); generalized_box_iou_loss also
    against the DETR utility of the installed `transformers` wheel.
State-dict keys follow torchvision (`head.classification_head.conv.{0,1,3,4,...}`, `head.regression_head.bbox_ctrness.weight`).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import detection as od
from . import retinanet as orn


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


def generalized_box_iou_loss(boxes1, boxes2, reduction="none", eps=1e-7):
    """torchvision.ops.generalized_box_iou_loss [EXT]."""
    x1, y1, x2, y2 = boxes1.unbind(dim=-1)
    x1g, y1g, x2g, y2g = boxes2.unbind(dim=-1)
    xkis1, ykis1 = torch.max(x1, x1g), torch.max(y1, y1g)
    xkis2, ykis2 = torch.min(x2, x2g), torch.min(y2, y2g)
    intsctk = torch.zeros_like(x1)
    mask = (ykis2 > ykis1) & (xkis2 > xkis1)
    intsctk[mask] = (xkis2[mask] - xkis1[mask]) * (ykis2[mask] - ykis1[mask])
    unionk = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - intsctk
    iouk = intsctk / (unionk + eps)
    xc1, yc1 = torch.min(x1, x1g), torch.min(y1, y1g)
    xc2, yc2 = torch.max(x2, x2g), torch.max(y2, y2g)
    area_c = (xc2 - xc1) * (yc2 - yc1)
    miouk = iouk - ((area_c - unionk) / (area_c + eps))
    loss = 1 - miouk
    if reduction == "mean":
        loss = loss.mean() if loss.numel() > 0 else 0.0 * loss.sum()
    elif reduction == "sum":
        loss = loss.sum()
    return loss


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
        self.q = lambda t: t
        self.pins, self.pin_name = od.NO_PINS, "reg"

    def tower(self, x, level=0):
        k = 0
        for l in self.conv:
            if isinstance(l, nn.ReLU):
                x = self.pins.relu((self.pin_name, level, k), x)
                k += 1
            else:
                x = l(x)
            if not isinstance(l, nn.GroupNorm):      # the product stores the conv output and the ReLU output in fp16
                x = self.q(x)
        return x


def _flat(t, k):
    N, _, H, W = t.shape
    return t.view(N, -1, k, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, k)


class FCOSClassificationHead(_GNTower):
    def __init__(self, in_channels, num_anchors, num_classes, prior_probability=0.01):
        super().__init__(in_channels)
        self.pin_name = "cls"
        self.num_classes, self.num_anchors = num_classes, num_anchors
        self.cls_logits = nn.Conv2d(in_channels, num_anchors * num_classes, 3, padding=1)
        nn.init.normal_(self.cls_logits.weight, std=0.01)
        nn.init.constant_(self.cls_logits.bias, -math.log((1 - prior_probability) / prior_probability))

    def forward(self, x):
        return torch.cat([_flat(self.cls_logits(self.tower(f, li)), self.num_classes) for li, f in enumerate(x)], dim=1)


class FCOSRegressionHead(_GNTower):
    def __init__(self, in_channels, num_anchors):
        super().__init__(in_channels)
        self.bbox_reg = nn.Conv2d(in_channels, num_anchors * 4, 3, padding=1)
        self.bbox_ctrness = nn.Conv2d(in_channels, num_anchors * 1, 3, padding=1)
        for l in (self.bbox_reg, self.bbox_ctrness):
            nn.init.normal_(l.weight, std=0.01)
            nn.init.zeros_(l.bias)

    def forward(self, x):
        reg, ctr = [], []
        for li, f in enumerate(x):
            t = self.tower(f, li)
            reg.append(_flat(self.pins.relu(("reg_out", li), self.bbox_reg(t)), 4))
            ctr.append(_flat(self.bbox_ctrness(t), 1))
        return torch.cat(reg, dim=1), torch.cat(ctr, dim=1)


class FCOSHead(nn.Module):
    def __init__(self, in_channels=256, num_anchors=1, num_classes=91):
        super().__init__()
        self.box_coder = BoxLinearCoder(normalize_by_size=True)
        self.classification_head = FCOSClassificationHead(in_channels, num_anchors, num_classes)
        self.regression_head = FCOSRegressionHead(in_channels, num_anchors)

    def forward(self, x):
        reg, ctr = self.regression_head(x)
        return {"cls_logits": self.classification_head(x), "bbox_regression": reg, "bbox_ctrness": ctr}

    def compute_loss(self, targets, head_outputs, anchors, matched_idxs):
        cls_logits, bbox_regression, bbox_ctrness = head_outputs["cls_logits"], head_outputs["bbox_regression"], head_outputs["bbox_ctrness"]
        all_cls, all_boxes = [], []
        for t, mi in zip(targets, matched_idxs):
            if len(t["labels"]) == 0:
                gc = t["labels"].new_zeros((len(mi),))
                gb = t["boxes"].new_zeros((len(mi), 4))
            else:
                gc = t["labels"][mi.clip(min=0)]
                gb = t["boxes"][mi.clip(min=0)]
            gc[mi < 0] = -1
            all_cls.append(gc)
            all_boxes.append(gb)
        all_cls = torch.stack(all_cls)
        fg = all_cls >= 0
        num_fg = fg.sum().item()
        onehot = torch.zeros_like(cls_logits)
        onehot[fg, all_cls[fg]] = 1.0
        loss_cls = orn.sigmoid_focal_loss(cls_logits, onehot, reduction="sum")
        pred = [self.box_coder.decode_single(br, a) for a, br in zip(anchors, bbox_regression)]
        loss_reg = generalized_box_iou_loss(torch.stack(pred)[fg].float(), torch.stack(all_boxes)[fg], reduction="sum")
        reg_t = torch.stack([self.box_coder.encode_single(a, b) for a, b in zip(anchors, all_boxes)], dim=0)
        if len(reg_t) == 0:
            ctr_t = reg_t.new_zeros(reg_t.size()[:-1])
        else:
            lr, tb = reg_t[:, :, [0, 2]], reg_t[:, :, [1, 3]]
            ctr_t = torch.sqrt((lr.min(dim=-1)[0] / lr.max(dim=-1)[0]) * (tb.min(dim=-1)[0] / tb.max(dim=-1)[0]))
        loss_ctr = F.binary_cross_entropy_with_logits(bbox_ctrness.squeeze(dim=2)[fg], ctr_t[fg], reduction="sum")
        d = max(1, num_fg)
        return {"classification": loss_cls / d, "bbox_regression": loss_reg / d, "bbox_ctrness": loss_ctr / d}


def fcos_anchor_generator():
    return od.AnchorGenerator(((8,), (16,), (32,), (64,), (128,)), ((1.0,),) * 5)


class FCOS(nn.Module):
    """fcos_resnet50_fpn re-headed to `num_classes` (detector.py:57-66) with the reference's fixed-size transform."""

    def __init__(self, num_classes=2, size=300):
        super().__init__()
        self.transform = od.FixedSizeTransform(size)
        self.backbone = orn.RetinaBackbone()           # same trunk: layer2-4 + FPN + LastLevelP6P7(256, 256)
        self.anchor_generator = fcos_anchor_generator()
        self.head = FCOSHead(256, self.anchor_generator.num_anchors_per_location()[0], 91)
        cls = nn.Conv2d(256, 1 * num_classes, 3, 1, 1)          # reference re-heading
        nn.init.normal_(cls.weight, std=0.01)
        nn.init.constant_(cls.bias, -math.log((1 - 0.01) / 0.01))
        self.head.classification_head.cls_logits = cls
        self.head.classification_head.num_classes = num_classes
        self.box_coder = BoxLinearCoder(normalize_by_size=True)
        self.center_sampling_radius = 1.5
        self.score_thresh, self.nms_thresh, self.detections_per_img, self.topk_candidates = 0.2, 0.6, 100, 1000

    def set_quant(self, q):
        self.backbone.q = q
        self.head.classification_head.q = q
        self.head.regression_head.q = q

    def set_pins(self, pins):
        """oracle.detection.Pins: ReLU / max-pool decisions taken from recorded tensors (tests)."""
        pins = pins or od.NO_PINS
        self.backbone.pins = self.head.classification_head.pins = self.head.regression_head.pins = pins

    def match(self, anchors_per_image, targets_per_image, num_anchors_per_level):
        if targets_per_image["boxes"].numel() == 0:
            return torch.full((anchors_per_image.size(0),), -1, dtype=torch.int64)
        gt = targets_per_image["boxes"]
        gt_centers = (gt[:, :2] + gt[:, 2:]) / 2
        ac = (anchors_per_image[:, :2] + anchors_per_image[:, 2:]) / 2
        size = anchors_per_image[:, 2] - anchors_per_image[:, 0]
        pm = (ac[:, None, :] - gt_centers[None, :, :]).abs_().max(dim=2).values < self.center_sampling_radius * size[:, None]
        x, y = ac.unsqueeze(dim=2).unbind(dim=1)
        x0, y0, x1, y1 = gt.unsqueeze(dim=0).unbind(dim=2)
        dist = torch.stack([x - x0, y - y0, x1 - x, y1 - y], dim=2)
        pm &= dist.min(dim=2).values > 0
        lower = size * 4
        lower[: num_anchors_per_level[0]] = 0
        upper = size * 8
        upper[-num_anchors_per_level[-1]:] = float("inf")
        dmax = dist.max(dim=2).values
        pm &= (dmax > lower[:, None]) & (dmax < upper[:, None])
        areas = (gt[:, 2] - gt[:, 0]) * (gt[:, 3] - gt[:, 1])
        val = pm.to(torch.float32) * (1e8 - areas[None, :])
        best, idx = val.max(dim=1)
        idx[best < 1e-5] = -1
        return idx

    def compute_loss(self, targets, head_outputs, anchors, num_anchors_per_level):
        matched = [self.match(a, t, num_anchors_per_level) for a, t in zip(anchors, targets)]
        return self.head.compute_loss(targets, head_outputs, anchors, matched)

    def postprocess_detections(self, head_outputs, anchors, image_shapes):
        class_logits, box_regression, box_ctrness = head_outputs["cls_logits"], head_outputs["bbox_regression"], head_outputs["bbox_ctrness"]
        detections = []
        for index in range(len(image_shapes)):
            ib, is_, il = [], [], []
            for breg, logits, ctr, anc in zip((b[index] for b in box_regression), (c[index] for c in class_logits),
                                              (c[index] for c in box_ctrness), anchors[index]):
                num_classes = logits.shape[-1]
                scores = torch.sqrt(torch.sigmoid(logits) * torch.sigmoid(ctr)).flatten()
                keep = scores > self.score_thresh
                scores = scores[keep]
                topk_idxs = torch.where(keep)[0]
                num_topk = min(self.topk_candidates, topk_idxs.size(0))
                order = torch.sort(scores, descending=True, stable=True)[1][:num_topk]   # == topk with deterministic ties
                scores, topk_idxs = scores[order], topk_idxs[order]
                anchor_idxs = torch.div(topk_idxs, num_classes, rounding_mode="floor")
                boxes = self.box_coder.decode_single(breg[anchor_idxs], anc[anchor_idxs])
                ib.append(od.clip_boxes_to_image(boxes, image_shapes[index]))
                is_.append(scores)
                il.append(topk_idxs % num_classes)
            ib, is_, il = torch.cat(ib, 0), torch.cat(is_, 0), torch.cat(il, 0)
            keep = od.batched_nms(ib, is_, il, self.nms_thresh)[: self.detections_per_img]
            detections.append({"boxes": ib[keep], "scores": is_[keep], "labels": il[keep]})
        return detections


def eval_forward_fcos(model, images, targets, train_det=False):
    """src/utils/eval_forward_fcos.py:11-83."""
    if not train_det:
        model.eval()
    original_sizes = [(img.shape[-2], img.shape[-1]) for img in images]
    il, targets = model.transform(images, targets)
    features = list(model.backbone(il.tensors).values())
    head_outputs = model.head(features)
    anchors = model.anchor_generator(il, features)
    napl = [f.size(2) * f.size(3) for f in features]
    losses = model.compute_loss(targets, head_outputs, anchors, napl)
    split = {k: list(v.split(napl, dim=1)) for k, v in head_outputs.items()}
    split_anchors = [list(a.split(napl)) for a in anchors]
    dets = model.postprocess_detections(split, split_anchors, il.image_sizes)
    dets = model.transform.postprocess(dets, il.image_sizes, original_sizes)
    return losses, dets
