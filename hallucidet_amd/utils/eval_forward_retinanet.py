"""Loss AND detections in one RetinaNet pass (reference src/utils/eval_forward_retinanet.py: sigmoid_focal_loss :22-50,
box_loss :53-80, eval_forward_retinanet :83-160, compute_retinanet_loss :163-178, compute_loss_classification_head
:181-211, compute_loss_regression_head :215-244): same function names, argument order, assertion messages, returned keys
({'classification', 'bbox_regression'}) and the same call sequence on the model (transform -> backbone -> head ->
anchor_generator -> losses -> postprocess_detections -> transform.postprocess).

The per-image python loops of the loss and the per-image / per-level loops of the post-processing run in padded,
batched form when `model.batched_heads` is set (hallucidet_amd.models.retinanet; same arithmetic -- the GPU tests assert
equality with the list-based functions below, which stay for exactly that purpose and for API compatibility)."""
from typing import Dict, Optional

import torch

from .eval_forward_fasterrcnn import _check_degenerate, _check_targets, check_degenerate_deferred
from ..models import detection as D
from ..models import retinanet as R


class _FocalElementwiseFn(torch.autograd.Function):
    """hd_sigmoid_focal_loss: the per-element loss and its input gradient, one launch each."""

    @staticmethod
    def forward(ctx, inputs, targets, alpha, gamma):
        from .. import _abi
        x, t = inputs.detach().contiguous().float(), targets.detach().contiguous().float()
        out = torch.empty_like(x)
        _abi.check(_abi.load().hd_sigmoid_focal_loss(_abi.ptr(x), _abi.ptr(t), x.numel(), alpha, gamma, None, _abi.ptr(out),
                                                     torch.cuda.current_stream().cuda_stream), "hd_sigmoid_focal_loss")
        ctx.save_for_backward(x, t)
        ctx.consts = (alpha, gamma)
        return out.view(inputs.shape)

    @staticmethod
    def backward(ctx, g):
        from .. import _abi
        x, t = ctx.saved_tensors
        gi = torch.empty_like(x)
        go = g.contiguous().float().expand_as(x).contiguous()
        _abi.check(_abi.load().hd_sigmoid_focal_loss(_abi.ptr(x), _abi.ptr(t), x.numel(), ctx.consts[0], ctx.consts[1], _abi.ptr(go), _abi.ptr(gi),
                                                     torch.cuda.current_stream().cuda_stream), "hd_sigmoid_focal_loss")
        return gi.view(g.shape), None, None, None


_REDUCTIONS = {"none": lambda v: v, "mean": torch.mean, "sum": torch.sum}


def sigmoid_focal_loss(inputs: torch.Tensor, targets: torch.Tensor, alpha: float = 0.25, gamma: float = 2,
                       reduction: str = "none") -> torch.Tensor:
    """Reference signature (:22-50); `targets` are the 0/1 class indicators the reference builds (:199-206).  The arithmetic
    is the HIP kernel's (losses.hip `focal_value` / `focal_grad`)."""
    if reduction not in _REDUCTIONS:
        raise ValueError(f"Invalid Value for arg 'reduction': '{reduction} \n Supported reduction modes: 'none', 'mean', 'sum'")
    if not inputs.is_cuda:
        raise RuntimeError("hallucidet_amd: sigmoid_focal_loss runs on the GPU (no CPU fallback; the CPU statement is oracle/retinanet.py)")
    return _REDUCTIONS[reduction](_FocalElementwiseFn.apply(inputs, targets, float(alpha), float(gamma)))


def box_loss(type: str, box_coder, anchors_per_image: torch.Tensor, matched_gt_boxes_per_image: torch.Tensor,
             bbox_regression_per_image: torch.Tensor, cnf: Optional[Dict[str, float]] = None) -> torch.Tensor:
    """Reference signature (:53-80): the regression loss of ONE image over its foreground anchors (sum reduction).  Evaluated
    by the batched kernel with every row foreground and matched to itself: B = 1, G = A."""
    torch._assert(type in ["l1", "smooth_l1", "ciou", "diou", "giou"], f"Unsupported loss: {type}")
    if type not in ("l1", "smooth_l1"):
        raise NotImplementedError("hallucidet_amd: the reference only ever calls box_loss with its default 'smooth_l1' "
                                  "(eval_forward_retinanet.py:215); the IoU-family losses live in the un-vendored torchvision.ops")
    n = anchors_per_image.shape[0]
    if n == 0:
        return bbox_regression_per_image.sum() * 0.0
    # beta -> 0 turns smooth-L1 into L1 (|d| - beta/2 for |d| >= beta); 1e-12 is exact in fp32 for any representable |d| > 1e-12
    beta = 1e-12 if type == "l1" else float(cnf["beta"] if cnf is not None and "beta" in cnf else 1.0)
    dev = bbox_regression_per_image.device
    own = torch.arange(n, device=dev, dtype=torch.int64)[None]
    zeros = torch.zeros((1, n, 1), dtype=torch.float32, device=dev)
    _, reg = R._RetinaNetLossFn.apply(zeros, bbox_regression_per_image[None], own, matched_gt_boxes_per_image[None].float(),
                                      torch.zeros((1, n), dtype=torch.int64, device=dev), anchors_per_image, tuple(box_coder.weights), 0.25, 2.0, beta)
    return reg * float(n)          # the kernel divides by max(1, #foreground) = n


def eval_forward_retinanet(model, images, targets, train_det=False, model_name='retinanet'):
    if train_det:
        if not getattr(model.backbone, "train_params", False):
            raise RuntimeError("hallucidet_amd: train_det=True needs detector.set_trainable(True) first (see "
                               "hallucidet_amd.train_detector.DetectorLit); the frozen-detector kernels emit data gradients only")
    else:
        model.eval()
    _check_targets(targets)
    original_image_sizes = [_hw_of(img) for img in images]

    images, targets = model.transform(images, targets)
    if targets is not None:
        _check_degenerate(targets)

    features = model.backbone(images.tensors)
    features = [features] if isinstance(features, torch.Tensor) else list(features.values())
    head_outputs, anchors = model.head(features), model.anchor_generator(images, features)
    # anchors per level (the feature maps are NHWC here: H, W = size(1), size(2)); A = anchors per location
    cells = [f.size(1) * f.size(2) for f in features]
    per_cell = head_outputs["cls_logits"].size(1) // sum(cells)
    napl = [c * per_cell for c in cells]

    if getattr(model, "batched_heads", False):
        gt, glab, gvalid = D.pad_targets(targets, images.tensors.device)
        losses = R.retinanet_loss_batched(model, anchors[0], gt, glab, gvalid, head_outputs["cls_logits"], head_outputs["bbox_regression"])
        return losses, _detections_padded(model, head_outputs, anchors[0], napl, images.image_sizes, original_image_sizes)

    # list-based route (API compatibility with the reference's model methods; the GPU tests compare it with the batched one)
    losses = compute_retinanet_loss(targets, head_outputs, anchors, model)
    per_level_outputs = {k: list(v.split(napl, dim=1)) for k, v in head_outputs.items()}
    per_level_anchors = [list(a.split(napl)) for a in anchors]
    raw = model.postprocess_detections(per_level_outputs, per_level_anchors, images.image_sizes)
    return losses, model.transform.postprocess(raw, images.image_sizes, original_image_sizes)


def _hw_of(img):
    hw = img.shape[-2:]
    torch._assert(len(hw) == 2, f"expecting the last two dimensions of the Tensor to be H and W instead got {img.shape[-2:]}")
    return (hw[0], hw[1])


def _detections_padded(model, head_outputs, anchors0, napl, image_sizes, original_image_sizes):
    """Deferred (LazyDetections.deferred): launched by the first access or by the training step's flush() after the backward pass."""
    cls_logits, bbox_regression = head_outputs["cls_logits"].detach(), head_outputs["bbox_regression"].detach()
    training = model.transform.training

    def postprocess():
        from ..models.custom_generalized_transform import _ratios
        from .eval_forward_fasterrcnn import _scale_tensor
        sb, ss, sl, counts = model.postprocess_detections_padded(cls_logits, bbox_regression, anchors0, napl, image_sizes[0])
        scale = None
        if not training:
            rh, rw = _ratios(image_sizes[0], original_image_sizes[0])
            scale = _scale_tensor(rw, rh, sb)
        return sb, ss, sl, counts, ((lambda b: b * scale) if scale is not None else None)
    return D.LazyDetections.deferred(postprocess, cls_logits.shape[0])


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
    flat_targets = [t for tl in target_lists for t in tl]
    il, flat_targets = model.transform.forward_batches(image_batches, flat_targets)       # every batch resized into its slice: no fp32 concat
    check_degenerate_deferred(model, flat_targets)          # no host synchronisation inside the step
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


def _matched_padded(targets, anchors, model):
    """[B, A] matched-GT table of the list inputs (box_iou + proposal_matcher per image, :165-175), via the fused target assignment."""
    gt, glab, gvalid = D.pad_targets(targets, anchors[0].device)
    return R.retinanet_match_batched(model, anchors[0], gt, gvalid), gt, glab


def compute_retinanet_loss(targets, head_outputs, anchors, model):
    """Reference signature (:163-178).  All images of a batch share one anchor set here (fixed 300 x 300 inputs), so the
    per-image loop of the reference is one batched evaluation."""
    m, gt, glab = _matched_padded(targets, anchors, model)
    return R.retinanet_loss_batched(model, anchors[0], gt, glab, None, head_outputs["cls_logits"], head_outputs["bbox_regression"], matched=m)


def compute_loss_classification_head(targets, head_outputs, matched_idxs, model):
    """Reference signature (:181-211): `matched_idxs` is the per-image list the reference's matcher returns."""
    dev = head_outputs["cls_logits"].device
    gt, glab, _ = D.pad_targets(targets, dev)
    m = torch.stack([v.to(dev) for v in matched_idxs])
    a0 = torch.tensor([[0.0, 0.0, 1.0, 1.0]], device=dev).expand(m.shape[1], 4).contiguous()      # boxes are not read by this head
    reg0 = torch.zeros(m.shape + (4,), dtype=torch.float32, device=dev)
    return R._RetinaNetLossFn.apply(head_outputs["cls_logits"], reg0, m, gt, glab, a0, (1.0, 1.0, 1.0, 1.0), 0.25, 2.0, 1.0)[0]


def compute_loss_regression_head(targets, head_outputs, anchors, matched_idxs, model, loss_reg='smooth_l1'):
    """Reference signature (:215-244)."""
    if loss_reg != 'smooth_l1':
        raise NotImplementedError("hallucidet_amd: the reference trains with its default smooth-L1 (beta 1) regression loss")
    dev = head_outputs["bbox_regression"].device
    gt, glab, _ = D.pad_targets(targets, dev)
    m = torch.stack([v.to(dev) for v in matched_idxs])
    cls0 = torch.zeros(m.shape + (1,), dtype=torch.float32, device=dev)
    return R._RetinaNetLossFn.apply(cls0, head_outputs["bbox_regression"], m, gt, torch.zeros_like(glab), anchors[0], tuple(model.box_coder.weights),
                                    0.25, 2.0, 1.0)[1]
