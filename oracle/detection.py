"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain torch fp32) of the detector side of the hot path.

What it follows
  * reference tree:
      src/models/detector.py:39-66,104-141                 model choice, transform swap, re-heading to 2 classes
      src/models/custom_generalized_transform.py:52-100,136-296,325-338   normalize / nearest resize / batch / postprocess
      src/utils/eval_forward_fasterrcnn.py:13-68,72-102,105-136           loss + detections in one pass
  * un-vendored dependency: torchvision 0.12.0 (requirements.txt:92) -- `torchvision.models.detection`
    (fasterrcnn_resnet50_fpn, RPN, RoIHeads, AnchorGenerator, BoxCoder, Matcher, samplers, MultiScaleRoIAlign) and
    `torchvision.ops` (nms, batched_nms, roi_align, box_iou).  torchvision is absent from this image, so its published
    algorithm is restated here.  PARITY UNPINNED against torchvision itself: the reference holds no
    tests or golden vectors for it.  What *is* pinned: the orchestration (tests drive the reference's own
    eval_forward_fasterrcnn.py over this object), the transform (golden from the reference file) and hand-worked
    known answers for NMS / RoIAlign / matcher / box coder / anchors; and, against code held outside this repository: the frozen
    ResNet-50 trunk against the ResNet of the installed `transformers` wheel (same weights: four stage outputs), box_iou against its
    DETR utilities, RoIAlign against ATen's grid_sample (tests/test_oracle_golden.py).

The attribute tree (`transform, backbone.body/fpn, rpn.{head,anchor_generator,box_coder,...}, roi_heads.{...}`) and the
state_dict key names mirror torchvision 0.12 so that (a) the reference glue can call into it duck-typed and (b) weights
are interchangeable with the product modules.

Device note: the reference runs on a GPU, where torchvision's batched_nms switches to the per-class loop only above
20000 box coordinates (4000 on CPU).  `NMS_VANILLA_NUMEL` defaults to the GPU value.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import kernels as ok

NMS_VANILLA_NUMEL = 20000


class Pins:
    """Test aid: take the DISCRETE decisions of the forward pass (ReLU on/off per element, the winning position of every
    max-pool window, optionally the post-NMS proposals) from recorded tensors instead of from this restatement's own
    activations.  The product stores activations in fp16, so its values differ from an fp32 evaluation by ~1e-3 relative and a
    ReLU whose input is within that noise of zero flips; each flip re-routes the gradient of everything behind it.  With the
    decisions pinned both sides evaluate the SAME piecewise-linear function and differ by rounding / summation order only, which
    is what a tight gradient comparison needs.  `masks`: {tag: float tensor (1 = pass) shaped like the oracle's activation};
    `pool`: int64 [N, C, Ho, Wo] in 0..8 (row-major position inside the 3x3 window); missing tags fall back to the plain op.

    A borrowed decision is only legitimate where this restatement's OWN value sits inside the fp16 noise band of the decision
    boundary, so every use is AUDITED against the oracle's own pre-activation: `audit[tag] = (elements, decisions that differ from the
    oracle's own, largest |own pre-activation| (ReLU) or value gap between the two winners (max-pool) among those, RMS of the layer's
    pre-activation, sigma / dmax = RMS / largest difference between the recorded activation and the oracle's own over the elements both
    sides pass -- the measured noise of that layer, None without `values`)`; for pinned proposals `audit[("proposals", image)] = (pinned
    boxes, boxes of the oracle's own post-NMS set, pinned boxes that coincide at IoU >= 0.9 / 0.7 / 0.5 with one of the oracle's own
    CANDIDATES -- every anchor decoded with the oracle's own deltas and clipped, before any score-driven selection)`.
    tests/_pins.py::assert_borrowed_decisions_are_noise turns the audit into assertions -- a systematically wrong mask in the
    product would otherwise be copied into the oracle and pass."""

    def __init__(self, masks=None, pool=None, proposals=None, values=None):
        self.masks, self.pool, self.proposals = masks or {}, pool, proposals
        self.values = values or {}       # {tag: the recorded post-ReLU activation}: gives the audit the layer's measured noise level
        self.used = set()
        self.audit = {}

    def relu(self, tag, x):
        m = self.masks.get(tag)
        if m is None:
            return F.relu(x)
        self.used.add(tag)
        m = m.reshape(x.shape)
        with torch.no_grad():
            xd = x.detach()
            flip = (m > 0) != (xd > 0)
            nf = int(flip.sum())
            sigma = dmax = None
            v = self.values.get(tag)
            if v is not None:                # difference between the two evaluations where both say "on": the layer's measured noise
                both = (m > 0) & (xd > 0)
                d = (v.reshape(x.shape) - xd)[both].abs()
                sigma = float(d.float().pow(2).mean().sqrt()) if d.numel() else 0.0
                dmax = float(d.max()) if d.numel() else 0.0
            self.audit[tag] = (xd.numel(), nf, float(xd[flip].abs().max()) if nf else 0.0, float(xd.float().pow(2).mean().sqrt()), sigma, dmax)
        return x * m

    def maxpool3x3s2(self, x):
        if self.pool is None:
            return F.max_pool2d(x, 3, 2, 1)
        xp = F.pad(x, (1, 1, 1, 1), value=float("-inf"))
        win = xp.unfold(2, 3, 2).unfold(3, 3, 2)                       # [N, C, Ho, Wo, 3, 3]
        win = win.reshape(win.shape[:4] + (9,))
        out = win.gather(4, self.pool.reshape(win.shape[:4] + (1,))).squeeze(4)
        with torch.no_grad():
            own = win.detach().max(dim=4).values
            gap = own - out.detach()                                    # >= 0; > 0 where the borrowed winner is not a maximum of the oracle's window
            diff = gap > 0
            nf = int(diff.sum())
            self.audit[("pool",)] = (out.numel(), nf, float(gap.max()) if nf else 0.0, float(x.detach().float().pow(2).mean().sqrt()), None, None)
        return out


NO_PINS = Pins()


# ----------------------------------------------------------------------------- box utilities
def box_area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def clip_boxes_to_image(boxes, size):
    h, w = size
    bx = boxes[..., 0::2].clamp(min=0, max=w)
    by = boxes[..., 1::2].clamp(min=0, max=h)
    return torch.stack((bx, by), dim=boxes.dim()).reshape(boxes.shape)


def remove_small_boxes(boxes, min_size):
    ws, hs = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    return torch.where((ws >= min_size) & (hs >= min_size))[0]


def nms(boxes, scores, thr):
    """Returns kept indices sorted by decreasing score (stable order among equal scores)."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    order = torch.sort(scores, descending=True, stable=True)[1]
    keep = ok.nms_sorted(boxes[order], thr)
    return order[keep]


def batched_nms(boxes, scores, idxs, thr):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    if boxes.numel() > NMS_VANILLA_NUMEL:
        keep_mask = torch.zeros_like(scores, dtype=torch.bool)
        for cid in torch.unique(idxs):
            cur = torch.where(idxs == cid)[0]
            keep_mask[cur[nms(boxes[cur], scores[cur], thr)]] = True
        ki = torch.where(keep_mask)[0]
        return ki[torch.sort(scores[ki], descending=True, stable=True)[1]]
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    return nms(boxes + offsets[:, None], scores, thr)


class BoxCoder:
    def __init__(self, weights, clip=math.log(1000.0 / 16)):
        self.weights = weights
        self.bbox_xform_clip = clip

    def encode_single(self, ref, prop):
        wx, wy, ww, wh = [torch.as_tensor(w, dtype=ref.dtype) for w in self.weights]
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
        hw_, hh_ = torch.tensor(0.5, dtype=pcx.dtype) * pw, torch.tensor(0.5, dtype=pcy.dtype) * ph
        return torch.stack((pcx - hw_, pcy - hh_, pcx + hw_, pcy + hh_), dim=2).flatten(1)

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

    def __init__(self, high, low, allow_low_quality_matches=False):
        self.high_threshold, self.low_threshold, self.allow = high, low, allow_low_quality_matches

    def __call__(self, mq):
        vals, matches = mq.max(dim=0)
        all_matches = matches.clone() if self.allow else None
        below = vals < self.low_threshold
        between = (vals >= self.low_threshold) & (vals < self.high_threshold)
        matches[below] = self.BELOW_LOW_THRESHOLD
        matches[between] = self.BETWEEN_THRESHOLDS
        if self.allow:
            best_per_gt, _ = mq.max(dim=1)
            pred_inds = torch.where(mq == best_per_gt[:, None])[1]
            matches[pred_inds] = all_matches[pred_inds]
        return matches


class BalancedPositiveNegativeSampler:
    """`randperm_fn(n)` is injectable so tests can feed identical permutations to oracle and product."""

    def __init__(self, batch_size_per_image, positive_fraction, randperm_fn=None):
        self.batch_size_per_image, self.positive_fraction = batch_size_per_image, positive_fraction
        self.randperm_fn = randperm_fn or (lambda n: torch.randperm(n))

    def __call__(self, matched_idxs):
        pos_idx, neg_idx = [], []
        for m in matched_idxs:
            positive = torch.where(m >= 1)[0]
            negative = torch.where(m == 0)[0]
            num_pos = min(positive.numel(), int(self.batch_size_per_image * self.positive_fraction))
            num_neg = min(negative.numel(), self.batch_size_per_image - num_pos)
            p = positive[self.randperm_fn(positive.numel())[:num_pos]]
            n = negative[self.randperm_fn(negative.numel())[:num_neg]]
            pm = torch.zeros_like(m, dtype=torch.uint8)
            nm = torch.zeros_like(m, dtype=torch.uint8)
            pm[p] = 1
            nm[n] = 1
            pos_idx.append(pm)
            neg_idx.append(nm)
        return pos_idx, neg_idx


# ----------------------------------------------------------------------------- transform
class ImageList:
    def __init__(self, tensors, image_sizes):
        self.tensors, self.image_sizes = tensors, image_sizes


def resize_boxes(boxes, original_size, new_size):
    rh, rw = [torch.tensor(s, dtype=torch.float32) / torch.tensor(o, dtype=torch.float32) for s, o in zip(new_size, original_size)]
    x1, y1, x2, y2 = boxes.unbind(1)
    return torch.stack((x1 * rw, y1 * rh, x2 * rw, y2 * rh), dim=1)


class FixedSizeTransform(nn.Module):
    """custom_generalized_transform.py:103-296 with fixed_size=(S,S), mean [0.0], std [1.0], size_divisible=1."""

    def __init__(self, size=300, image_mean=(0.0,), image_std=(1.0,)):
        super().__init__()
        self.size, self.image_mean, self.image_std = size, list(image_mean), list(image_std)
        self.fixed_size = (size, size)

    def forward(self, images, targets=None):
        images = [img for img in images]
        if targets is not None:
            targets = [dict(t) for t in targets]
        for i, img in enumerate(images):
            if img.dim() != 3:
                raise ValueError(f"images is expected to be a list of 3d tensors of shape [C, H, W], got {img.shape}")
            if not img.is_floating_point():
                raise TypeError(f"Expected input images to be of floating type (in range [0, 1]), but found type {img.dtype} instead")
            mean = torch.as_tensor(self.image_mean, dtype=img.dtype)
            std = torch.as_tensor(self.image_std, dtype=img.dtype)
            img = (img - mean[:, None, None]) / std[:, None, None]
            h, w = img.shape[-2:]
            img = F.interpolate(img[None], size=[self.fixed_size[1], self.fixed_size[0]])[0]
            images[i] = img
            if targets is not None:
                targets[i]["boxes"] = resize_boxes(targets[i]["boxes"], (h, w), img.shape[-2:])
        sizes = [(int(im.shape[-2]), int(im.shape[-1])) for im in images]
        return ImageList(torch.stack(images), sizes), targets

    def postprocess(self, result, image_shapes, original_image_sizes):
        if self.training:
            return result
        for i, (pred, im_s, o_im_s) in enumerate(zip(result, image_shapes, original_image_sizes)):
            result[i]["boxes"] = resize_boxes(pred["boxes"], im_s, o_im_s)
        return result


# ----------------------------------------------------------------------------- backbone
class FrozenBatchNorm2d(nn.Module):
    def __init__(self, c, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(c))
        self.register_buffer("bias", torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))

    def scale_shift(self):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        return scale, self.bias - self.running_mean * scale

    def forward(self, x):
        s, b = self.scale_shift()
        return x * s.reshape(1, -1, 1, 1) + b.reshape(1, -1, 1, 1)


class Bottleneck(nn.Module):
    def __init__(self, cin, width, stride):
        super().__init__()
        cout = width * 4
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = FrozenBatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, 1, bias=False)   # v1.5: stride on the 3x3
        self.bn2 = FrozenBatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, cout, 1, bias=False)
        self.bn3 = FrozenBatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), FrozenBatchNorm2d(cout))

    def forward(self, x, q, pins=NO_PINS, tag=()):
        idt = x if self.downsample is None else q(self.downsample[1](self.downsample[0](x)))
        o = q(pins.relu(tag + (1,), self.bn1(self.conv1(x))))
        o = q(pins.relu(tag + (2,), self.bn2(self.conv2(o))))
        return q(pins.relu(tag + (3,), self.bn3(self.conv3(o)) + idt))


class ResNet50Body(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = FrozenBatchNorm2d(64)
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

    def forward(self, x, q, pins=NO_PINS):
        x = q(pins.relu(("stem",), self.bn1(self.conv1(x))))
        x = q(pins.maxpool3x3s2(x))
        out = OrderedDict()
        for i, name in enumerate(("layer1", "layer2", "layer3", "layer4")):
            for bi, blk in enumerate(getattr(self, name)):
                x = blk(x, q, pins, ("b", i, bi))
            out[str(i)] = x
        return out


class FeaturePyramidNetwork(nn.Module):
    def __init__(self, in_channels=(256, 512, 1024, 2048), out_channels=256):
        super().__init__()
        self.inner_blocks = nn.ModuleList(nn.Conv2d(c, out_channels, 1) for c in in_channels)
        self.layer_blocks = nn.ModuleList(nn.Conv2d(out_channels, out_channels, 3, padding=1) for _ in in_channels)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x, q):
        names, xs = list(x.keys()), list(x.values())
        last = q(self.inner_blocks[-1](xs[-1]))
        results = [q(self.layer_blocks[-1](last))]
        for i in range(len(xs) - 2, -1, -1):
            lat = q(self.inner_blocks[i](xs[i]))
            last = q(lat + F.interpolate(last, size=lat.shape[-2:], mode="nearest"))
            results.insert(0, q(self.layer_blocks[i](last)))
        names.append("pool")
        results.append(F.max_pool2d(results[-1], 1, 2, 0))
        return OrderedDict(zip(names, results))


class BackboneWithFPN(nn.Module):
    out_channels = 256

    def __init__(self):
        super().__init__()
        self.body = ResNet50Body()
        self.fpn = FeaturePyramidNetwork()
        self.q = lambda t: t
        self.pins = NO_PINS

    def forward(self, x):
        return self.fpn(self.body(self.q(x), self.q, self.pins), self.q)


# ----------------------------------------------------------------------------- RPN
class AnchorGenerator(nn.Module):
    def __init__(self, sizes=((32,), (64,), (128,), (256,), (512,)), aspect_ratios=((0.5, 1.0, 2.0),) * 5):
        super().__init__()
        self.sizes, self.aspect_ratios = sizes, aspect_ratios

    def num_anchors_per_location(self):
        return [len(s) * len(a) for s, a in zip(self.sizes, self.aspect_ratios)]

    @staticmethod
    def base_anchors(scales, ratios):
        scales = torch.as_tensor(scales, dtype=torch.float32)
        ratios = torch.as_tensor(ratios, dtype=torch.float32)
        h_r = torch.sqrt(ratios)
        w_r = 1 / h_r
        ws = (w_r[:, None] * scales[None, :]).view(-1)
        hs = (h_r[:, None] * scales[None, :]).view(-1)
        return (torch.stack([-ws, -hs, ws, hs], dim=1) / 2).round()

    def forward(self, image_list, feature_maps):
        grid_sizes = [fm.shape[-2:] for fm in feature_maps]
        ih, iw = image_list.tensors.shape[-2:]
        per_level = []
        for (gh, gw), s, a in zip(grid_sizes, self.sizes, self.aspect_ratios):
            sh, sw = ih // gh, iw // gw
            sx = torch.arange(0, gw, dtype=torch.int32) * sw
            sy = torch.arange(0, gh, dtype=torch.int32) * sh
            yy, xx = torch.meshgrid(sy, sx, indexing="ij")
            xx, yy = xx.reshape(-1), yy.reshape(-1)
            shifts = torch.stack((xx, yy, xx, yy), dim=1)
            per_level.append((shifts.view(-1, 1, 4) + self.base_anchors(s, a).view(1, -1, 4)).reshape(-1, 4))
        allv = torch.cat(per_level)
        return [allv for _ in range(len(image_list.image_sizes))]


class RPNHead(nn.Module):
    def __init__(self, in_channels=256, num_anchors=3):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, 3, padding=1)
        self.cls_logits = nn.Conv2d(in_channels, num_anchors, 1)
        self.bbox_pred = nn.Conv2d(in_channels, num_anchors * 4, 1)
        for layer in self.children():
            nn.init.normal_(layer.weight, std=0.01)
            nn.init.constant_(layer.bias, 0)
        self.q = lambda t: t
        self.pins = NO_PINS

    def forward(self, x):
        logits, regs = [], []
        for li, f in enumerate(x):
            t = self.q(self.pins.relu(("rpn", li), self.conv(f)))
            logits.append(self.cls_logits(t))
            regs.append(self.bbox_pred(t))
        return logits, regs


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
    def __init__(self, randperm_fn=None):
        super().__init__()
        self.anchor_generator = AnchorGenerator()
        self.head = RPNHead()
        self.box_coder = BoxCoder((1.0, 1.0, 1.0, 1.0))
        self.proposal_matcher = Matcher(0.7, 0.3, allow_low_quality_matches=True)
        self.fg_bg_sampler = BalancedPositiveNegativeSampler(256, 0.5, randperm_fn)
        self._pre_nms_top_n = dict(training=2000, testing=1000)
        self._post_nms_top_n = dict(training=2000, testing=1000)
        self.nms_thresh, self.score_thresh, self.min_size = 0.7, 0.0, 1e-3
        self.pinned_proposals = None
        self.pins_audit = None

    def pre_nms_top_n(self):
        return self._pre_nms_top_n["training" if self.training else "testing"]

    def post_nms_top_n(self):
        return self._post_nms_top_n["training" if self.training else "testing"]

    def assign_targets_to_anchors(self, anchors, targets):
        labels, matched = [], []
        for a, t in zip(anchors, targets):
            gt = t["boxes"]
            if gt.numel() == 0:
                matched.append(torch.zeros(a.shape, dtype=torch.float32))
                labels.append(torch.zeros((a.shape[0],), dtype=torch.float32))
                continue
            mi = self.proposal_matcher(ok.box_iou(gt, a))
            matched.append(gt[mi.clamp(min=0)])
            lab = (mi >= 0).to(torch.float32)
            lab[mi == Matcher.BELOW_LOW_THRESHOLD] = 0.0
            lab[mi == Matcher.BETWEEN_THRESHOLDS] = -1.0
            labels.append(lab)
        return labels, matched

    def _get_top_n_idx(self, objectness, num_anchors_per_level):
        r, offset = [], 0
        for ob in objectness.split(num_anchors_per_level, 1):
            n = ob.shape[1]
            # stable descending sort == topk with deterministic tie order
            idx = torch.sort(ob, dim=1, descending=True, stable=True)[1][:, : min(self.pre_nms_top_n(), n)]
            r.append(idx + offset)
            offset += n
        return torch.cat(r, dim=1)

    def filter_proposals(self, proposals, objectness, image_shapes, num_anchors_per_level):
        if self.pinned_proposals is not None:        # tests: the product's post-NMS proposal sets (Pins docstring)
            fb = [b.clone() for b in self.pinned_proposals]
            if self.pins_audit is not None:          # ... audited against this restatement's own sets
                pinned, self.pinned_proposals = self.pinned_proposals, None
                try:
                    own, _ = self.filter_proposals(proposals, objectness, image_shapes, num_anchors_per_level)
                finally:
                    self.pinned_proposals = pinned
                for i, (pb, ob) in enumerate(zip(fb, own)):
                    cand = clip_boxes_to_image(proposals[i].detach().reshape(-1, 4), image_shapes[i])
                    hit = [0, 0, 0]
                    for lo in range(0, pb.shape[0], 256):               # [256, A] IoU blocks
                        best = ok.box_iou(pb[lo:lo + 256], cand).max(dim=1).values
                        for j, thr in enumerate((0.9, 0.7, 0.5)):
                            hit[j] += int((best >= thr).sum())
                    self.pins_audit[("proposals", i)] = (pb.shape[0], ob.shape[0], hit[0], hit[1], hit[2])
            return fb, [torch.zeros(b.shape[0]) for b in fb]
        n_img = proposals.shape[0]
        objectness = objectness.detach().reshape(n_img, -1)
        levels = torch.cat([torch.full((n,), i, dtype=torch.int64) for i, n in enumerate(num_anchors_per_level)], 0)
        levels = levels.reshape(1, -1).expand_as(objectness)
        top = self._get_top_n_idx(objectness, num_anchors_per_level)
        bidx = torch.arange(n_img)[:, None]
        objectness, levels, proposals = objectness[bidx, top], levels[bidx, top], proposals[bidx, top]
        prob = torch.sigmoid(objectness)
        fb, fs = [], []
        for boxes, scores, lvl, shp in zip(proposals, prob, levels, image_shapes):
            boxes = clip_boxes_to_image(boxes, shp)
            keep = remove_small_boxes(boxes, self.min_size)
            boxes, scores, lvl = boxes[keep], scores[keep], lvl[keep]
            keep = torch.where(scores >= self.score_thresh)[0]
            boxes, scores, lvl = boxes[keep], scores[keep], lvl[keep]
            keep = batched_nms(boxes, scores, lvl, self.nms_thresh)[: self.post_nms_top_n()]
            fb.append(boxes[keep])
            fs.append(scores[keep])
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


# ----------------------------------------------------------------------------- RoI heads
class MultiScaleRoIAlign(nn.Module):
    def __init__(self, featmap_names=("0", "1", "2", "3"), output_size=7, sampling_ratio=2):
        super().__init__()
        self.featmap_names, self.output_size, self.sampling_ratio = list(featmap_names), (output_size, output_size), sampling_ratio
        self.canonical_scale, self.canonical_level, self.eps = 224, 4, 1e-6

    @staticmethod
    def infer_scale(feature, original_size):
        s = [2 ** float(torch.tensor(float(a) / float(b)).log2().round()) for a, b in zip(feature.shape[-2:], original_size)]
        return s[0]

    def level_map(self, boxes, k_min, k_max):
        s = torch.sqrt(torch.cat([box_area(b) for b in boxes]))
        t = torch.floor(self.canonical_level + torch.log2(s / self.canonical_scale) + torch.tensor(self.eps, dtype=s.dtype))
        return (torch.clamp(t, min=k_min, max=k_max).to(torch.int64) - k_min).to(torch.int64)

    def forward(self, x, boxes, image_shapes):
        feats = [v for k, v in x.items() if k in self.featmap_names]
        rois = torch.cat([torch.cat([torch.full_like(b[:, :1], i), b], dim=1) for i, b in enumerate(boxes)], dim=0)
        mh = max(s[0] for s in image_shapes)
        mw = max(s[1] for s in image_shapes)
        scales = [self.infer_scale(f, (mh, mw)) for f in feats]
        k_min = int(-torch.log2(torch.tensor(scales[0], dtype=torch.float32)).item())
        k_max = int(-torch.log2(torch.tensor(scales[-1], dtype=torch.float32)).item())
        levels = self.level_map(boxes, k_min, k_max)
        out = torch.zeros((rois.shape[0], feats[0].shape[1]) + self.output_size, dtype=feats[0].dtype)
        for lvl, (f, sc) in enumerate(zip(feats, scales)):
            idx = torch.where(levels == lvl)[0]
            out[idx] = roi_align_autograd(f, rois[idx], self.output_size[0], sc, self.sampling_ratio)
        return out


def roi_align_autograd(feat, rois, P, scale, sr):
    """Vectorised, differentiable RoIAlign (aligned=False), same arithmetic as oracle.kernels.roi_align_nchw."""
    R = rois.shape[0]
    N, C, H, W = feat.shape
    if R == 0:
        return feat.new_zeros((0, C, P, P))
    n = rois[:, 0].long()
    rs_w, rs_h, re_w, re_h = [rois[:, k] * scale for k in (1, 2, 3, 4)]
    rw = torch.clamp(re_w - rs_w, min=1.0)
    rh = torch.clamp(re_h - rs_h, min=1.0)
    bh, bw = rh / P, rw / P
    ph = torch.arange(P, dtype=torch.float32)
    iy = torch.arange(sr, dtype=torch.float32)
    # y[r, ph, iy], x[r, pw, ix]
    y = rs_h[:, None, None] + ph[None, :, None] * bh[:, None, None] + (iy[None, None, :] + 0.5) * bh[:, None, None] / sr
    x = rs_w[:, None, None] + ph[None, :, None] * bw[:, None, None] + (iy[None, None, :] + 0.5) * bw[:, None, None] / sr
    y = y.reshape(R, P * sr)
    x = x.reshape(R, P * sr)

    def prep(v, size):
        valid = ~((v < -1.0) | (v > size))
        v = v.clamp(min=0)
        lo = v.floor().long()
        top = lo >= size - 1
        lo = torch.where(top, torch.full_like(lo, size - 1), lo)
        hi = torch.where(top, lo, lo + 1)
        v = torch.where(top, lo.to(v.dtype), v)
        l = v - lo.to(v.dtype)
        return lo, hi, l, 1.0 - l, valid

    yl, yh, ly, hy, vy = prep(y, H)
    xl, xh, lx, hx, vx = prep(x, W)
    flat = feat.permute(0, 2, 3, 1).reshape(N * H * W, C)          # gather rows instead of materialising feat[n]
    base = (n * (H * W))[:, None, None]

    def gather(yi, xi):
        idx = base + yi[:, :, None] * W + xi[:, None, :]              # [R, P*sr, P*sr]
        return flat[idx.reshape(-1)].reshape(R, P * sr, P * sr, C).permute(0, 3, 1, 2)

    w1 = (hy[:, :, None] * hx[:, None, :])[:, None]
    w2 = (hy[:, :, None] * lx[:, None, :])[:, None]
    w3 = (ly[:, :, None] * hx[:, None, :])[:, None]
    w4 = (ly[:, :, None] * lx[:, None, :])[:, None]
    val = w1 * gather(yl, xl) + w2 * gather(yl, xh) + w3 * gather(yh, xl) + w4 * gather(yh, xh)
    val = val * (vy[:, :, None] & vx[:, None, :])[:, None].to(val.dtype)
    val = val.reshape(R, C, P, sr, P, sr).sum(dim=(3, 5)) / float(max(sr * sr, 1))
    return val


class TwoMLPHead(nn.Module):
    def __init__(self, in_channels=256 * 7 * 7, rep=1024):
        super().__init__()
        self.fc6 = nn.Linear(in_channels, rep)
        self.fc7 = nn.Linear(rep, rep)
        self.q = lambda t: t
        self.pins = NO_PINS

    def forward(self, x):
        x = self.q(x).flatten(start_dim=1)
        h6 = self.q(self.pins.relu(("fc6",), self.fc6(x)))
        return self.q(self.pins.relu(("fc7",), self.fc7(h6)))


class FastRCNNPredictor(nn.Module):
    def __init__(self, in_channels=1024, num_classes=2):
        super().__init__()
        self.cls_score = nn.Linear(in_channels, num_classes)
        self.bbox_pred = nn.Linear(in_channels, num_classes * 4)

    def forward(self, x):
        x = x.flatten(start_dim=1)
        return self.cls_score(x), self.bbox_pred(x)


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
    def __init__(self, num_classes=2, randperm_fn=None):
        super().__init__()
        self.box_roi_pool = MultiScaleRoIAlign()
        self.box_head = TwoMLPHead()
        self.box_predictor = FastRCNNPredictor(1024, num_classes)
        self.box_coder = BoxCoder((10.0, 10.0, 5.0, 5.0))
        self.proposal_matcher = Matcher(0.5, 0.5, allow_low_quality_matches=False)
        self.fg_bg_sampler = BalancedPositiveNegativeSampler(512, 0.25, randperm_fn)
        self.score_thresh, self.nms_thresh, self.detections_per_img = 0.05, 0.5, 100
        self.keypoint_roi_pool = self.keypoint_head = self.keypoint_predictor = None
        self.mask_roi_pool = self.mask_head = self.mask_predictor = None

    def has_keypoint(self):
        return False

    def has_mask(self):
        return False

    def select_training_samples(self, proposals, targets):
        dtype = proposals[0].dtype
        gt_boxes = [t["boxes"].to(dtype) for t in targets]
        gt_labels = [t["labels"] for t in targets]
        proposals = [torch.cat((p, g)) for p, g in zip(proposals, gt_boxes)]
        matched_idxs, labels = [], []
        for p, g, gl in zip(proposals, gt_boxes, gt_labels):
            if g.numel() == 0:
                matched_idxs.append(torch.zeros((p.shape[0],), dtype=torch.int64))
                labels.append(torch.zeros((p.shape[0],), dtype=torch.int64))
                continue
            mi = self.proposal_matcher(ok.box_iou(g, p))
            lab = gl[mi.clamp(min=0)].to(torch.int64)
            lab[mi == Matcher.BELOW_LOW_THRESHOLD] = 0
            lab[mi == Matcher.BETWEEN_THRESHOLDS] = -1
            matched_idxs.append(mi.clamp(min=0))
            labels.append(lab)
        pos, neg = self.fg_bg_sampler(labels)
        matched_gt = []
        for i in range(len(proposals)):
            s = torch.where(pos[i] | neg[i])[0]
            proposals[i], labels[i], matched_idxs[i] = proposals[i][s], labels[i][s], matched_idxs[i][s]
            g = gt_boxes[i] if gt_boxes[i].numel() else torch.zeros((1, 4), dtype=dtype)
            matched_gt.append(g[matched_idxs[i]])
        return proposals, matched_idxs, labels, self.box_coder.encode(matched_gt, proposals)

    def postprocess_detections(self, class_logits, box_regression, proposals, image_shapes):
        num_classes = class_logits.shape[-1]
        per = [b.shape[0] for b in proposals]
        pred_boxes = self.box_coder.decode(box_regression, proposals).split(per, 0)
        pred_scores = F.softmax(class_logits, -1).split(per, 0)
        ab, as_, al = [], [], []
        for boxes, scores, shp in zip(pred_boxes, pred_scores, image_shapes):
            boxes = clip_boxes_to_image(boxes, shp)
            labels = torch.arange(num_classes).view(1, -1).expand_as(scores)
            boxes, scores, labels = boxes[:, 1:].reshape(-1, 4), scores[:, 1:].reshape(-1), labels[:, 1:].reshape(-1)
            inds = torch.where(scores > self.score_thresh)[0]
            boxes, scores, labels = boxes[inds], scores[inds], labels[inds]
            keep = remove_small_boxes(boxes, 1e-2)
            boxes, scores, labels = boxes[keep], scores[keep], labels[keep]
            keep = batched_nms(boxes, scores, labels, self.nms_thresh)[: self.detections_per_img]
            ab.append(boxes[keep])
            as_.append(scores[keep])
            al.append(labels[keep])
        return ab, as_, al


class FasterRCNN(nn.Module):
    """fasterrcnn_resnet50_fpn re-headed to `num_classes` with the reference's fixed-size transform (detector.py:39-55)."""

    def __init__(self, num_classes=2, size=300, randperm_fn=None):
        super().__init__()
        self.transform = FixedSizeTransform(size)
        self.backbone = BackboneWithFPN()
        self.rpn = RegionProposalNetwork(randperm_fn)
        self.roi_heads = RoIHeads(num_classes, randperm_fn)

    def set_quant(self, q):
        """Apply the product path's fp16 rounding schedule (see oracle.unet.fp16_round)."""
        self.backbone.q = q
        self.rpn.head.q = q
        self.roi_heads.box_head.q = q

    def set_pins(self, pins):
        """See `Pins`.  `None` restores the plain operations."""
        pins = pins or NO_PINS
        self.backbone.pins = self.rpn.head.pins = self.roi_heads.box_head.pins = pins
        self.rpn.pinned_proposals = pins.proposals
        self.rpn.pins_audit = pins.audit if pins.proposals is not None else None


def eval_forward_fasterrcnn(model, images, targets, train_det=False):
    """Oracle statement of src/utils/eval_forward_fasterrcnn.py:13-68 (+ rpn_eval :72-102, roi_heads_eval :105-136)."""
    if not train_det:
        model.eval()
    for t in targets:
        b = t["boxes"]
        torch._assert(isinstance(b, torch.Tensor) and b.dim() == 2 and b.shape[-1] == 4,
                      f"Expected target boxes to be a tensor of shape [N, 4], got {getattr(b, 'shape', type(b))}.")
    original_sizes = [(img.shape[-2], img.shape[-1]) for img in images]
    il, targets = model.transform(images, targets)
    for ti, t in enumerate(targets):
        deg = t["boxes"][:, 2:] <= t["boxes"][:, :2]
        if deg.any():
            bb = t["boxes"][torch.where(deg.any(dim=1))[0][0]].tolist()
            torch._assert(False, "All bounding boxes should have positive height and width."
                          f" Found invalid box {bb} for target at index {ti}.")
    features = model.backbone(il.tensors)
    # --- RPN
    feats = list(features.values())
    objectness, deltas = model.rpn.head(feats)
    anchors = model.rpn.anchor_generator(il, feats)
    n_img = len(anchors)
    napl = [o[0].shape[0] * o[0].shape[1] * o[0].shape[2] for o in objectness]
    objectness, deltas = concat_box_prediction_layers(objectness, deltas)
    proposals = model.rpn.box_coder.decode(deltas.detach(), anchors).view(n_img, -1, 4)
    boxes, _ = model.rpn.filter_proposals(proposals, objectness, il.image_sizes, napl)
    labels, matched_gt = model.rpn.assign_targets_to_anchors(anchors, targets)
    reg_t = model.rpn.box_coder.encode(matched_gt, anchors)
    loss_obj, loss_rpn_box = model.rpn.compute_loss(objectness, deltas, labels, reg_t)
    # --- RoI heads (training-style sampling with GT injected, reference :120)
    props, _, lab, reg_targets = model.roi_heads.select_training_samples(boxes, targets)
    bf = model.roi_heads.box_roi_pool(features, props, il.image_sizes)
    bf = model.roi_heads.box_head(bf)
    logits, box_reg = model.roi_heads.box_predictor(bf)
    loss_cls, loss_box = fastrcnn_loss(logits, box_reg, lab, reg_targets)
    b, s, l = model.roi_heads.postprocess_detections(logits, box_reg, props, il.image_sizes)
    dets = [{"boxes": bi, "labels": li, "scores": si} for bi, si, li in zip(b, s, l)]
    dets = model.transform.postprocess(dets, il.image_sizes, original_sizes)
    losses = {"loss_classifier": loss_cls, "loss_box_reg": loss_box, "loss_objectness": loss_obj, "loss_rpn_box_reg": loss_rpn_box}
    return losses, dets
