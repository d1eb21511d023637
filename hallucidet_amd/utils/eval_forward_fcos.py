"""Loss AND detections in one FCOS pass (reference src/utils/eval_forward_fcos.py:11-83): same function name, argument order,
assertion messages, returned keys ({'classification', 'bbox_regression', 'bbox_ctrness'}) and the same call sequence on the
model (transform -> backbone -> head -> anchor_generator -> compute_loss -> postprocess_detections -> transform.postprocess).

With `model.batched_heads` (the default of hallucidet_amd.models.fcos.FCOS) the per-image / per-level python loops of the
post-processing run in padded, batched form (same arithmetic; the GPU tests assert equality with the list route below)."""
import torch

from .eval_forward_fasterrcnn import _check_degenerate, _check_targets, check_degenerate_deferred, _scale_tensor
from .eval_forward_retinanet import _hw_of
from ..models import detection as D
from ..models import fcos as F_


def eval_forward_fcos(model, images, targets, train_det=False, model_name='fcos'):
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
    head_outputs = model.head(features)
    anchors = model.anchor_generator(images, features)
    # locations per level (the feature maps are NHWC here: H, W = size(1), size(2))
    num_anchors_per_level = [f.size(1) * f.size(2) for f in features]

    losses = model.compute_loss(targets, head_outputs, anchors, num_anchors_per_level)

    if getattr(model, "batched_heads", False):
        return losses, _detections_padded(model, head_outputs, anchors[0], num_anchors_per_level, images.image_sizes, original_image_sizes)
    per_level_outputs = {k: list(v.split(num_anchors_per_level, dim=1)) for k, v in head_outputs.items()}
    per_level_anchors = [list(a.split(num_anchors_per_level)) for a in anchors]
    raw = model.postprocess_detections(per_level_outputs, per_level_anchors, images.image_sizes)
    return losses, model.transform.postprocess(raw, images.image_sizes, original_image_sizes)


def _detections_padded(model, head_outputs, anchors0, napl, image_sizes, original_image_sizes):
    """Deferred (LazyDetections.deferred): launched by the first access or by the training step's flush() after the backward pass."""
    ho = {k: v.detach() for k, v in head_outputs.items()}
    training = model.transform.training

    def postprocess():
        from ..models.custom_generalized_transform import _ratios
        sb, ss, sl, counts = model.postprocess_detections_padded(ho, anchors0, napl, image_sizes[0])
        scale = None
        if not training:
            rh, rw = _ratios(image_sizes[0], original_image_sizes[0])
            scale = _scale_tensor(rw, rh, sb)
        return sb, ss, sl, counts, ((lambda b: b * scale) if scale is not None else None)
    return D.LazyDetections.deferred(postprocess, ho["cls_logits"].shape[0])


def eval_forward_fcos_multi(model, image_batches, target_lists, model_name='fcos'):
    """The hallucinated / RGB / IR detector passes of one training step (train_hallucidet.py:180,183,186) as ONE transform +
    trunk + head evaluation over the concatenated batch (frozen eval-mode detector: images are independent; GroupNorm
    normalises per image).  Only the first batch carries a gradient and a loss (the reference discards the other two).  FCOS has
    no sampler, so the result is exactly that of three separate passes."""
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
    gt, glab, gvalid = D.pad_targets(flat_targets[:n0], il.tensors.device)
    m = model.match_batched(anchors[0], gt, gvalid, napl)
    losses = F_.fcos_loss_batched(anchors[0], gt, glab, {k: v[:n0] for k, v in head_outputs.items()}, m)
    dets = _detections_padded(model, head_outputs, anchors[0], napl, il.image_sizes, sizes[0]).split(nb)
    return [(losses if k == 0 else {}, d) for k, d in enumerate(dets)]
