"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain torch fp32) of the RetinaNet side of the hot path (BASELINE config 4).

Follows
  * reference tree: src/models/detector.py:57-66 (re-heading of cls_logits to n_classes, N(0,0.01) weights, bias -log(99)),
    src/utils/eval_forward_retinanet.py:22-80 (sigmoid_focal_loss, box_loss with smooth-L1 beta=1 default),
    :83-160 (eval_forward_retinanet), :163-244 (compute_retinanet_loss / classification / regression heads);
  * un-vendored torchvision 0.12 `retinanet_resnet50_fpn` [EXT]: ResNet-50 body returning layer2-4, FPN over
    (512,1024,2048) with LastLevelP6P7(256,256) (P6 = conv3x3 s2 on P5 because in==out channels, P7 = conv3x3 s2 on
    ReLU(P6)), RetinaNetHead (4x[conv3x3+ReLU] towers, cls_logits / bbox_reg conv3x3), AnchorGenerator with sizes
    (x, int(x*2^(1/3)), int(x*2^(2/3))) for x in 32..512 and ratios (0.5,1,2) => 9 anchors per location, Matcher(0.5, 0.4,
    allow_low_quality=True), BoxCoder (1,1,1,1), postprocess (score>0.05, top-1000 per level, NMS 0.5, 300 detections).
    PARITY UNPINNED against torchvision itself (absent); the loss functions and the orchestration ARE pinned by driving the
    reference's own eval_forward_retinanet.py over this object (tests/golden/make_golden.py: glue_retinanet.npz); sigmoid_focal_loss
    also against the DETR utility of the installed `transformers` wheel.
State-dict keys follow torchvision 0.12 (`backbone.fpn.extra_blocks.p6.weight`, `head.classification_head.conv.0.weight`, ...).
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import detection as od
from . import kernels as ok


class LastLevelP6P7(nn.Module):
    def __init__(self, in_channels=256, out_channels=256):
        super().__init__()
        self.p6 = nn.Conv2d(in_channels, out_channels, 3, 2, 1)
        self.p7 = nn.Conv2d(out_channels, out_channels, 3, 2, 1)
        for m in (self.p6, self.p7):
            nn.init.kaiming_uniform_(m.weight, a=1)
            nn.init.constant_(m.bias, 0)
        self.use_P5 = in_channels == out_channels


class FPN3(nn.Module):
    def __init__(self, in_channels=(512, 1024, 2048), out_channels=256):
        super().__init__()
        self.inner_blocks = nn.ModuleList(nn.Conv2d(c, out_channels, 1) for c in in_channels)
        self.layer_blocks = nn.ModuleList(nn.Conv2d(out_channels, out_channels, 3, padding=1) for _ in in_channels)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1)
                nn.init.constant_(m.bias, 0)
        self.extra_blocks = LastLevelP6P7(out_channels, out_channels)

    def forward(self, xs, q, pins=od.NO_PINS):
        last = q(self.inner_blocks[-1](xs[-1]))
        results = [q(self.layer_blocks[-1](last))]
        for i in range(len(xs) - 2, -1, -1):
            lat = q(self.inner_blocks[i](xs[i]))
            last = q(lat + F.interpolate(last, size=lat.shape[-2:], mode="nearest"))
            results.insert(0, q(self.layer_blocks[i](last)))
        p6 = q(self.extra_blocks.p6(results[-1]))
        p7 = q(self.extra_blocks.p7(q(pins.relu(("p7in",), p6))))
        return OrderedDict(zip(("0", "1", "2", "p6", "p7"), results + [p6, p7]))


class RetinaBackbone(nn.Module):
    out_channels = 256

    def __init__(self):
        super().__init__()
        self.body = od.ResNet50Body()
        self.fpn = FPN3()
        self.q = lambda t: t
        self.pins = od.NO_PINS

    def forward(self, x):
        feats = self.body(self.q(x), self.q, self.pins)          # '0'..'3' = layer1..4
        return self.fpn([feats["1"], feats["2"], feats["3"]], self.q, self.pins)


class _Tower(nn.Module):
    def __init__(self, in_channels, out_name, out_channels):
        super().__init__()
        layers = []
        for _ in range(4):
            layers += [nn.Conv2d(in_channels, in_channels, 3, padding=1), nn.ReLU()]
        self.conv = nn.Sequential(*layers)
        for l in self.conv.children():
            if isinstance(l, nn.Conv2d):
                nn.init.normal_(l.weight, std=0.01)
                nn.init.constant_(l.bias, 0)
        setattr(self, out_name, nn.Conv2d(in_channels, out_channels, 3, padding=1))
        self._out = out_name
        self.q = lambda t: t
        self.pins, self.pin_name = od.NO_PINS, "cls" if out_name == "cls_logits" else "reg"

    def tower(self, x, level=0):
        k = 0
        for l in self.conv:
            if isinstance(l, nn.ReLU):
                x = self.q(self.pins.relu((self.pin_name, level, k), x))
                k += 1
            else:
                x = l(x)
        return x


class RetinaNetClassificationHead(_Tower):
    BETWEEN_THRESHOLDS = od.Matcher.BETWEEN_THRESHOLDS

    def __init__(self, in_channels, num_anchors, num_classes, prior_probability=0.01):
        super().__init__(in_channels, "cls_logits", num_anchors * num_classes)
        nn.init.normal_(self.cls_logits.weight, std=0.01)
        nn.init.constant_(self.cls_logits.bias, -math.log((1 - prior_probability) / prior_probability))
        self.num_classes, self.num_anchors = num_classes, num_anchors

    def forward(self, x):
        out = []
        for li, f in enumerate(x):
            t = self.cls_logits(self.tower(f, li))
            N, _, H, W = t.shape
            out.append(t.view(N, -1, self.num_classes, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, self.num_classes))
        return torch.cat(out, dim=1)


class RetinaNetRegressionHead(_Tower):
    def __init__(self, in_channels, num_anchors):
        super().__init__(in_channels, "bbox_reg", num_anchors * 4)
        nn.init.normal_(self.bbox_reg.weight, std=0.01)
        nn.init.zeros_(self.bbox_reg.bias)

    def forward(self, x):
        out = []
        for li, f in enumerate(x):
            t = self.bbox_reg(self.tower(f, li))
            N, _, H, W = t.shape
            out.append(t.view(N, -1, 4, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, 4))
        return torch.cat(out, dim=1)


class RetinaNetHead(nn.Module):
    def __init__(self, in_channels=256, num_anchors=9, num_classes=91):
        super().__init__()
        self.classification_head = RetinaNetClassificationHead(in_channels, num_anchors, num_classes)
        self.regression_head = RetinaNetRegressionHead(in_channels, num_anchors)

    def forward(self, x):
        return {"cls_logits": self.classification_head(x), "bbox_regression": self.regression_head(x)}


def retinanet_anchor_generator():
    sizes = tuple((x, int(x * 2 ** (1.0 / 3)), int(x * 2 ** (2.0 / 3))) for x in [32, 64, 128, 256, 512])
    return od.AnchorGenerator(sizes, ((0.5, 1.0, 2.0),) * len(sizes))


class RetinaNet(nn.Module):
    """retinanet_resnet50_fpn re-headed to `num_classes` (detector.py:57-66) with the reference's fixed-size transform."""

    def __init__(self, num_classes=2, size=300):
        super().__init__()
        self.transform = od.FixedSizeTransform(size)
        self.backbone = RetinaBackbone()
        self.anchor_generator = retinanet_anchor_generator()
        self.head = RetinaNetHead(256, self.anchor_generator.num_anchors_per_location()[0], 91)
        # reference re-heading
        cls = nn.Conv2d(256, 9 * num_classes, 3, 1, 1)
        nn.init.normal_(cls.weight, std=0.01)
        nn.init.constant_(cls.bias, -math.log((1 - 0.01) / 0.01))
        self.head.classification_head.cls_logits = cls
        self.head.classification_head.num_classes = num_classes
        self.proposal_matcher = od.Matcher(0.5, 0.4, allow_low_quality_matches=True)
        self.box_coder = od.BoxCoder((1.0, 1.0, 1.0, 1.0))
        self.score_thresh, self.nms_thresh, self.detections_per_img, self.topk_candidates = 0.05, 0.5, 300, 1000

    def set_quant(self, q):
        self.backbone.q = q
        self.head.classification_head.q = q
        self.head.regression_head.q = q

    def set_pins(self, pins):
        """oracle.detection.Pins: ReLU / max-pool decisions taken from recorded tensors (tests)."""
        pins = pins or od.NO_PINS
        self.backbone.pins = self.head.classification_head.pins = self.head.regression_head.pins = pins

    def postprocess_detections(self, head_outputs, anchors, image_shapes):
        class_logits, box_regression = head_outputs["cls_logits"], head_outputs["bbox_regression"]
        num_images = len(image_shapes)
        detections = []
        for index in range(num_images):
            box_regression_per_image = [br[index] for br in box_regression]
            logits_per_image = [cl[index] for cl in class_logits]
            anchors_per_image, image_shape = anchors[index], image_shapes[index]
            ib, is_, il = [], [], []
            for breg, logits, anc in zip(box_regression_per_image, logits_per_image, anchors_per_image):
                num_classes = logits.shape[-1]
                scores = torch.sigmoid(logits).flatten()
                keep = scores > self.score_thresh
                scores = scores[keep]
                topk_idxs = torch.where(keep)[0]
                num_topk = min(self.topk_candidates, topk_idxs.size(0))
                order = torch.sort(scores, descending=True, stable=True)[1][:num_topk]   # == topk with deterministic ties
                scores, topk_idxs = scores[order], topk_idxs[order]
                anchor_idxs = torch.div(topk_idxs, num_classes, rounding_mode="floor")
                labels = topk_idxs % num_classes
                boxes = self.box_coder.decode_single(breg[anchor_idxs], anc[anchor_idxs])
                boxes = od.clip_boxes_to_image(boxes, image_shape)
                ib.append(boxes)
                is_.append(scores)
                il.append(labels)
            ib, is_, il = torch.cat(ib, 0), torch.cat(is_, 0), torch.cat(il, 0)
            keep = od.batched_nms(ib, is_, il, self.nms_thresh)[: self.detections_per_img]
            detections.append({"boxes": ib[keep], "scores": is_[keep], "labels": il[keep]})
        return detections


# ----------------------------------------------------------------------------------------------------------------------
# losses (eval_forward_retinanet.py:22-80,163-244)
# ----------------------------------------------------------------------------------------------------------------------
def sigmoid_focal_loss(inputs, targets, alpha=0.25, gamma=2, reduction="none"):
    p = torch.sigmoid(inputs)
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    if reduction == "mean":
        loss = loss.mean()
    elif reduction == "sum":
        loss = loss.sum()
    return loss


def compute_retinanet_loss(targets, head_outputs, anchors, model):
    matched_idxs = []
    for a, t in zip(anchors, targets):
        if t["boxes"].numel() == 0:
            matched_idxs.append(torch.full((a.size(0),), -1, dtype=torch.int64))
            continue
        matched_idxs.append(model.proposal_matcher(ok.box_iou(t["boxes"], a)))
    cls_losses, reg_losses = [], []
    for t, cl, br, a, mi in zip(targets, head_outputs["cls_logits"], head_outputs["bbox_regression"], anchors, matched_idxs):
        fg = mi >= 0
        num_fg = fg.sum()
        tgt = torch.zeros_like(cl)
        tgt[fg, t["labels"][mi[fg]]] = 1.0
        valid = mi != od.Matcher.BETWEEN_THRESHOLDS
        cls_losses.append(sigmoid_focal_loss(cl[valid], tgt[valid], reduction="sum") / max(1, num_fg))
        fi = torch.where(fg)[0]
        target_reg = model.box_coder.encode_single(t["boxes"][mi[fi]], a[fi])
        reg_losses.append(F.smooth_l1_loss(br[fi], target_reg, reduction="sum", beta=1.0) / max(1, fi.numel()))
    return {"classification": sum(cls_losses) / len(targets), "bbox_regression": sum(reg_losses) / max(1, len(targets))}


def eval_forward_retinanet(model, images, targets, train_det=False):
    if not train_det:
        model.eval()
    original_sizes = [(img.shape[-2], img.shape[-1]) for img in images]
    il, targets = model.transform(images, targets)
    features = list(model.backbone(il.tensors).values())
    head_outputs = model.head(features)
    anchors = model.anchor_generator(il, features)
    losses = compute_retinanet_loss(targets, head_outputs, anchors, model)
    napl = [f.size(2) * f.size(3) for f in features]
    A = head_outputs["cls_logits"].size(1) // sum(napl)
    napl = [n * A for n in napl]
    split = {k: list(v.split(napl, dim=1)) for k, v in head_outputs.items()}
    split_anchors = [list(a.split(napl)) for a in anchors]
    dets = model.postprocess_detections(split, split_anchors, il.image_sizes)
    dets = model.transform.postprocess(dets, il.image_sizes, original_sizes)
    return losses, dets
