"""Loss AND detections in one RetinaNet pass (reference src/utils/eval_forward_retinanet.py: sigmoid_focal_loss :22-50,
box_loss :53-80, eval_forward_retinanet :83-160, compute_retinanet_loss :163-178, compute_loss_classification_head
:181-211, compute_loss_regression_head :215-244): same function names, argument order, assertion messages, returned keys
({'classification', 'bbox_regression'}) and the same call sequence on the model (transform -> backbone -> head ->
anchor_generator -> losses -> postprocess_detections -> transform.postprocess).

The per-image python loops of the loss and the per-image / per-level loops of the post-processing run in padded,
batched form when `model.batched_heads` is set (hallucidet_amd.models.retinanet; same arithmetic -- the GPU tests assert
equality with the list-based functions below, which stay for exactly that purpose and for API compatibility)."""
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from .eval_forward_fasterrcnn import _check_degenerate, _check_targets
from ..models import detection as D
from ..models import retinanet as R


def _sum(x: List[torch.Tensor]) -> torch.Tensor:
    res = x[0]
    for i in x[1:]:
        res = res + i
    return res


def sigmoid_focal_loss(inputs: torch.Tensor, targets: torch.Tensor, alpha: float = 0.25, gamma: float = 2,
                       reduction: str = "none") -> torch.Tensor:
    p = torch.sigmoid(inputs)
    ce_loss = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce_loss * ((1 - p_t) ** gamma)
    if alpha >= 0:
        alpha_t = alpha * targets + (1 - alpha) * (1 - targets)
        loss = alpha_t * loss
    if reduction == "none":
        pass
    elif reduction == "mean":
        loss = loss.mean()
    elif reduction == "sum":
        loss = loss.sum()
    else:
        raise ValueError(f"Invalid Value for arg 'reduction': '{reduction} \n Supported reduction modes: 'none', 'mean', 'sum'")
    return loss


def box_loss(type: str, box_coder, anchors_per_image: torch.Tensor, matched_gt_boxes_per_image: torch.Tensor,
             bbox_regression_per_image: torch.Tensor, cnf: Optional[Dict[str, float]] = None) -> torch.Tensor:
    torch._assert(type in ["l1", "smooth_l1", "ciou", "diou", "giou"], f"Unsupported loss: {type}")
    if type == "l1":
        target_regression = box_coder.encode_single(matched_gt_boxes_per_image, anchors_per_image)
        return F.l1_loss(bbox_regression_per_image, target_regression, reduction="sum")
    if type == "smooth_l1":
        target_regression = box_coder.encode_single(matched_gt_boxes_per_image, anchors_per_image)
        beta = cnf["beta"] if cnf is not None and "beta" in cnf else 1.0
        return F.smooth_l1_loss(bbox_regression_per_image, target_regression, reduction="sum", beta=beta)
    raise NotImplementedError("hallucidet_amd: the reference only ever calls box_loss with its default 'smooth_l1' "
                              "(eval_forward_retinanet.py:215); the IoU-family losses live in the un-vendored torchvision.ops")


def eval_forward_retinanet(model, images, targets, train_det=False, model_name='retinanet'):
    if train_det:
        if not getattr(model.backbone, "train_params", False):
            raise RuntimeError("hallucidet_amd: train_det=True needs detector.set_trainable(True) first (see "
                               "hallucidet_amd.train_detector.DetectorLit); the frozen-detector kernels emit data gradients only")
    else:
        model.eval()
    _check_targets(targets)
    original_image_sizes: List[Tuple[int, int]] = []
    for img in images:
        val = img.shape[-2:]
        torch._assert(len(val) == 2, f"expecting the last two dimensions of the Tensor to be H and W instead got {img.shape[-2:]}")
        original_image_sizes.append((val[0], val[1]))

    images, targets = model.transform(images, targets)
    if targets is not None:
        _check_degenerate(targets)

    features = model.backbone(images.tensors)
    if isinstance(features, torch.Tensor):
        features = OrderedDict([("0", features)])
    features = list(features.values())
    head_outputs = model.head(features)
    anchors = model.anchor_generator(images, features)

    # recover level sizes (features are NHWC here: H, W = size(1), size(2))
    num_anchors_per_level = [x.size(1) * x.size(2) for x in features]
    HW = sum(num_anchors_per_level)
    A = head_outputs["cls_logits"].size(1) // HW
    num_anchors_per_level = [hw * A for hw in num_anchors_per_level]

    if getattr(model, "batched_heads", False):
        gt, glab, gvalid = D.pad_targets(targets, images.tensors.device)
        losses = R.retinanet_loss_batched(model, anchors[0], gt, glab, gvalid, head_outputs["cls_logits"], head_outputs["bbox_regression"])
        detections = _detections_padded(model, head_outputs, anchors[0], num_anchors_per_level, images.image_sizes, original_image_sizes)
        return losses, detections

    losses = compute_retinanet_loss(targets, head_outputs, anchors, model)
    split_head_outputs: Dict[str, List[torch.Tensor]] = {}
    for k in head_outputs:
        split_head_outputs[k] = list(head_outputs[k].split(num_anchors_per_level, dim=1))
    split_anchors = [list(a.split(num_anchors_per_level)) for a in anchors]
    detections = model.postprocess_detections(split_head_outputs, split_anchors, images.image_sizes)
    detections = model.transform.postprocess(detections, images.image_sizes, original_image_sizes)
    return losses, detections


def _detections_padded(model, head_outputs, anchors0, napl, image_sizes, original_image_sizes):
    from ..models.custom_generalized_transform import _ratios
    sb, ss, sl, counts = model.postprocess_detections_padded(head_outputs["cls_logits"], head_outputs["bbox_regression"],
                                                             anchors0, napl, image_sizes[0])
    scale = None
    if not model.transform.training:
        rh, rw = _ratios(image_sizes[0], original_image_sizes[0])
        from .eval_forward_fasterrcnn import _scale_tensor
        scale = _scale_tensor(rw, rh, sb)
    return D.LazyDetections(sb, ss, sl, counts, (lambda b: b * scale) if scale is not None else None)


def eval_forward_retinanet_multi(model, image_batches, target_lists, model_name='retinanet'):
    """The hallucinated / RGB / IR detector passes of one training step (train_hallucidet.py:180,183,186) as ONE
    transform + trunk + head evaluation over the concatenated batch (frozen eval-mode detector: images are independent).
    Only the first batch carries a gradient and a loss (the reference discards the other two).  RetinaNet has no sampler,
    so -- unlike the Faster R-CNN fusion -- the result is exactly that of three separate passes."""
    model.eval()
    for t in target_lists:
        _check_targets(t)
    sizes = [[(img.shape[-2], img.shape[-1]) for img in b] for b in image_batches]
    nb = [len(s) for s in sizes]
    x = torch.cat([b if isinstance(b, torch.Tensor) else torch.stack(list(b)) for b in image_batches], dim=0)
    flat_targets = [t for tl in target_lists for t in tl]
    il, flat_targets = model.transform(x, flat_targets)
    _check_degenerate(flat_targets)
    n0 = nb[0]
    if image_batches[0].requires_grad:
        features = list(model.backbone(il.tensors, n_active=n0).values())
        head_outputs = model.head(features, n_active=n0)
    else:
        with torch.no_grad():
            features = list(model.backbone(il.tensors).values())
            head_outputs = model.head(features)
    anchors = model.anchor_generator(il, features)
    napl = [f.size(1) * f.size(2) for f in features]
    A = head_outputs["cls_logits"].size(1) // sum(napl)
    napl = [n * A for n in napl]
    gt, glab, gvalid = D.pad_targets(flat_targets[:n0], il.tensors.device)
    losses = R.retinanet_loss_batched(model, anchors[0], gt, glab, gvalid, head_outputs["cls_logits"][:n0], head_outputs["bbox_regression"][:n0])
    dets = _detections_padded(model, head_outputs, anchors[0], napl, il.image_sizes, sizes[0]).split(nb)
    return [(losses if k == 0 else {}, d) for k, d in enumerate(dets)]


def compute_retinanet_loss(targets, head_outputs, anchors, model):
    matched_idxs = []
    for anchors_per_image, targets_per_image in zip(anchors, targets):
        if targets_per_image["boxes"].numel() == 0:
            matched_idxs.append(torch.full((anchors_per_image.size(0),), -1, dtype=torch.int64, device=anchors_per_image.device))
            continue
        match_quality_matrix = D.box_iou(targets_per_image["boxes"], anchors_per_image)
        matched_idxs.append(model.proposal_matcher(match_quality_matrix))
    return {"classification": compute_loss_classification_head(targets, head_outputs, matched_idxs, model),
            "bbox_regression": compute_loss_regression_head(targets, head_outputs, anchors, matched_idxs, model)}


def compute_loss_classification_head(targets, head_outputs, matched_idxs, model):
    losses = []
    cls_logits = head_outputs["cls_logits"]
    for targets_per_image, cls_logits_per_image, matched_idxs_per_image in zip(targets, cls_logits, matched_idxs):
        foreground_idxs_per_image = matched_idxs_per_image >= 0
        num_foreground = foreground_idxs_per_image.sum()
        gt_classes_target = torch.zeros_like(cls_logits_per_image)
        gt_classes_target[foreground_idxs_per_image,
                          targets_per_image["labels"][matched_idxs_per_image[foreground_idxs_per_image]]] = 1.0
        valid_idxs_per_image = matched_idxs_per_image != model.head.classification_head.BETWEEN_THRESHOLDS
        losses.append(sigmoid_focal_loss(cls_logits_per_image[valid_idxs_per_image], gt_classes_target[valid_idxs_per_image],
                                         reduction="sum") / max(1, num_foreground))
    return _sum(losses) / len(targets)


def compute_loss_regression_head(targets, head_outputs, anchors, matched_idxs, model, loss_reg='smooth_l1'):
    losses = []
    bbox_regression = head_outputs["bbox_regression"]
    for targets_per_image, bbox_regression_per_image, anchors_per_image, matched_idxs_per_image in zip(
            targets, bbox_regression, anchors, matched_idxs):
        foreground_idxs_per_image = torch.where(matched_idxs_per_image >= 0)[0]
        num_foreground = foreground_idxs_per_image.numel()
        matched_gt_boxes_per_image = targets_per_image["boxes"][matched_idxs_per_image[foreground_idxs_per_image]]
        bbox_regression_per_image = bbox_regression_per_image[foreground_idxs_per_image, :]
        anchors_per_image = anchors_per_image[foreground_idxs_per_image, :]
        losses.append(box_loss(loss_reg, model.box_coder, anchors_per_image, matched_gt_boxes_per_image,
                               bbox_regression_per_image) / max(1, num_foreground))
    return _sum(losses) / max(1, len(targets))
