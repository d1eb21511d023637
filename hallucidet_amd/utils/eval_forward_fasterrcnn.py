"""Loss AND detections in one detector pass (reference src/utils/eval_forward_fasterrcnn.py:13-68, rpn_eval :72-102,
roi_heads_eval :105-185): same function names, argument order, assertions/messages, returned keys and the same
torchvision call sequence (including `select_training_samples` at evaluation time)."""
from collections import OrderedDict
from typing import Dict, List, Tuple

import os

import torch

from ..models.detection import concat_box_prediction_layers, fastrcnn_loss


def _check_targets(targets):
    """Shape / type test of the target boxes with the reference's messages (:24-31)."""
    for t in targets:
        b = t["boxes"]
        if not isinstance(b, torch.Tensor):
            torch._assert(False, f"Expected target boxes to be of type Tensor, got {type(b)}.")
        torch._assert(b.dim() == 2 and b.shape[-1] == 4, f"Expected target boxes to be a tensor of shape [N, 4], got {b.shape}.")


def _degenerate_flag(targets):
    """Device-side flag of the reference's degenerate-box check (:41-53); read at the step's first natural sync."""
    if targets and "_rows" in targets[0]:
        # staged targets (det_graph.py): every image carries the same number of rows, `_rows` marks the real ones (the rest is
        # zero padding, which must not trip the check)
        if "_flag" in targets[0]:
            return targets[0]["_flag"]          # computed when the rows were staged (det_graph.py)
        from ..models.detection import stack_rows
        allb = stack_rows([t["boxes"] for t in targets])
        live = stack_rows([t["_rows"] for t in targets])
        return ((allb[..., 2:] <= allb[..., :2]).any(dim=-1) & live).any()
    allb = torch.cat([t["boxes"] for t in targets], dim=0)
    return (allb[:, 2:] <= allb[:, :2]).any() if allb.numel() else None


# While det_graph.py captures the detector half this is a list: the batch's degenerate-box flag is appended to it (a uint8 [1]
# tensor, an OUTPUT of the graph) instead of being sent to the host from inside the capture; its trip to pinned memory is issued
# after every replay.
_GRAPH_FLAGS = None


def _defer_flag(model, flag, targets):
    if _GRAPH_FLAGS is not None:
        if flag is not None:
            _GRAPH_FLAGS.append(flag.reshape(1).to(torch.uint8))
        return
    pend = model.__dict__.pop("_pending_degenerate", None)
    if pend is not None:
        pend.raise_if_set()
    if flag is not None:
        model.__dict__["_pending_degenerate"] = _AsyncFlag(flag, targets)


def check_degenerate_deferred(model, targets):
    """The reference's degenerate-box check (:41-53) without a host synchronisation inside the step: this batch's device flag
    travels to pinned memory now and is read when the NEXT batch arrives (a bad box still raises, one call later, with the same
    message); the flag left by the previous call is read first."""
    flag = _degenerate_flag(targets)
    if flag is not None and not flag.is_cuda:
        pend = model.__dict__.pop("_pending_degenerate", None)
        if pend is not None:
            pend.raise_if_set()
        _raise_if_degenerate(flag, targets)
    else:
        _defer_flag(model, flag, targets)


def flush_degenerate(model, block=True):
    """Read the degenerate-box flag the last deferred check left behind (fit_step: right before the optimizer step, without
    waiting -- a bad batch whose flag has already arrived never updates the parameters; epoch ends / validation: blocking, so the
    flag of a run's last batch is never lost).  Raises the reference's assertion (:41-53) if it is set."""
    pend = model.__dict__.get("_pending_degenerate")
    if pend is not None and (block or pend.ready()):
        model.__dict__.pop("_pending_degenerate", None)
        pend.raise_if_set()


_PAD_ROIS = os.environ.get("HD_PAD_ROIS", "1") != "0"      # fixed-size RoI stage of the fused three-pass evaluation (A/B knob)


class _AsyncFlag:
    """A device bool on its way to pinned host memory; `raise_if_set()` waits for THAT copy only (an .item() on the tensor would
    drain everything queued on the stream)."""

    def __init__(self, flag, targets):
        self.targets = targets
        self.host = torch.empty(1, dtype=torch.uint8).pin_memory()
        self.host.copy_(flag if flag.dtype == torch.uint8 else flag.reshape(1).to(torch.uint8), non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()

    def ready(self):
        return self.event.query()

    def raise_if_set(self):
        self.event.synchronize()
        if int(self.host[0]):
            _check_degenerate(self.targets)


def _raise_if_degenerate(flag, targets):
    if flag is not None and bool(flag):
        _check_degenerate(targets)


def _check_degenerate(targets):
    """One fused test (a single host sync) instead of one `.any()` per image; on a hit, the reference's message (:41-53) for the first
    offending box."""
    allb = torch.cat([t["boxes"] for t in targets], dim=0)
    if not (allb.numel() and bool((allb[:, 2:] <= allb[:, :2]).any())):
        return
    for ti, t in enumerate(targets):
        bad = (t["boxes"][:, 2:] <= t["boxes"][:, :2]).any(dim=1)
        if bad.any():
            box = t["boxes"][torch.where(bad)[0][0]].tolist()
            torch._assert(False, "All bounding boxes should have positive height and width."
                          f" Found invalid box {box} for target at index {ti}.")



class _ImageSlice:
    """ImageList view of a sub-batch (what the RPN / anchor code needs: image_sizes and the layout tag)."""

    def __init__(self, il, lo, hi):
        self.tensors = il.tensors[lo:hi]
        self.image_sizes = il.image_sizes[lo:hi]
        self.layout = il.layout


def eval_forward_fasterrcnn_multi(model, image_batches, target_lists, model_name='fasterrcnn', fused=None):
    """The hallucinated / RGB / IR detector passes of one training step (train_hallucidet.py:180,183,186) with ONE
    transform + ResNet-50-FPN + RPN-head evaluation over the concatenated batch (the detector is frozen and in eval
    mode, so every image is independent), followed by the reference's per-pass logic -- proposals, target assignment,
    sampling (same `randperm` call order as three separate passes), RoI heads, losses, detections -- on the slices.
    Only the first batch carries a gradient.  Returns [(losses, detections)] per pass."""
    model.eval()
    for t in target_lists:
        _check_targets(t)
    sizes = [[(img.shape[-2], img.shape[-1]) for img in b] for b in image_batches]
    nb = [len(s) for s in sizes]
    flat_targets = [t for tl in target_lists for t in tl]
    il, flat_targets = model.transform.forward_batches(image_batches, flat_targets)       # every batch resized into its slice: no fp32 concat
    fused = getattr(model, "fused_passes", False) if fused is None else fused
    if fused and getattr(model, "batched_heads", False):
        return _multi_fused(model, il, flat_targets, nb, sizes, image_batches[0].requires_grad)
    _check_degenerate(flat_targets)
    n_active = nb[0] if image_batches[0].requires_grad else 0
    if n_active:
        features = model.backbone(il.tensors, n_active=n_active)
        objectness, deltas = model.rpn.head(list(features.values()), n_active=n_active)
    else:
        with torch.no_grad():
            features = model.backbone(il.tensors)
            objectness, deltas = model.rpn.head(list(features.values()))
    out, lo = [], 0
    for k, n in enumerate(nb):
        hi = lo + n
        ctx = torch.enable_grad() if (k == 0 and n_active) else torch.no_grad()
        with ctx:
            # the first pass's slice is the [n_active] view the backbone hands out for differentiation (detection._active_views)
            f_k = OrderedDict((name, v._hd_active if (k == 0 and getattr(v, "_hd_active", None) is not None and hi == v._hd_active.shape[0]) else v[lo:hi])
                              for name, v in features.items())
            o_k = [o[lo:hi] for o in objectness]
            d_k = [d[lo:hi] for d in deltas]
            il_k = _ImageSlice(il, lo, hi)
            t_k = flat_targets[lo:hi]
            if getattr(model, "batched_heads", False):
                proposal_losses, detector_losses, detections = _heads_batched(model, il_k, f_k, o_k, d_k, t_k)
            else:
                proposals, proposal_losses = rpn_eval(model, il_k, f_k, t_k, head_out=(o_k, d_k))
                detections, detector_losses = roi_heads_eval(model, f_k, proposals, il_k.image_sizes, t_k)
            detections = model.transform.postprocess(detections, il_k.image_sizes, sizes[k])
        out.append(({**detector_losses, **proposal_losses}, detections))
        lo = hi
    return out


def _multi_fused(model, il, targets, nb, sizes, need_grad):
    """All passes in ONE head evaluation over the concatenated batch: 2 host syncs per step instead of 9, one third of the
    launches.  The sampler draws RPN(image 0..N-1) then RoI(image 0..N-1) -- every draw has the reference's distribution,
    but the interleaving differs from three separate passes (RPN, RoI, RPN, RoI, ...); `fused_passes=False` keeps the
    reference's interleaving.  Losses are taken over the first pass's images only."""
    from ..models import detection as D
    n0 = nb[0]
    flag = _degenerate_flag(targets)
    _target_dtypes(targets)
    dev = il.tensors.device
    gt, glab, gvalid = D.pad_targets(targets, dev)
    shape = il.image_sizes[0]
    # RPN target assignment + sampling depend on targets and anchors only: do them (and the sampler's host sync) BEFORE
    # the trunk is launched whenever the anchors of this image size are already known (every step but the first)
    cache = model.rpn.__dict__.setdefault("_anchor_by_size", {})
    key = (tuple(shape), len(targets) > 0, str(dev))
    rpn_state = None
    if key in cache:
        rpn_state = D.rpn_targets_sample_batched(model.rpn, cache[key], gt, gvalid, n_loss=n0)
    if need_grad:
        features = model.backbone(il.tensors, n_active=n0)
        objectness, deltas = model.rpn.head(list(features.values()), n_active=n0)
    else:
        with torch.no_grad():
            features = model.backbone(il.tensors)
            objectness, deltas = model.rpn.head(list(features.values()))
    feats = list(features.values())
    anchors = model.rpn.anchor_generator(il, feats)
    cache[key] = anchors[0]
    n_img = len(anchors)
    napl = [o[0].shape[0] * o[0].shape[1] * o[0].shape[2] for o in objectness]
    obj, dl = concat_box_prediction_layers(objectness, deltas)
    pb, _, pc = D.filter_proposals_padded(model.rpn, None, obj, shape, napl, deltas=dl, anchors0=anchors[0])
    if rpn_state is None:
        rpn_state = D.rpn_targets_sample_batched(model.rpn, anchors[0], gt, gvalid, n_loss=n0)
    loss_objectness, loss_rpn_box_reg = D.rpn_loss_from_samples(rpn_state, obj, dl)
    rh = model.roi_heads
    padded = _PAD_ROIS and pb.is_cuda and rh.fg_bg_sampler.randperm_fn is None
    pool, head, pred = rh.box_roi_pool, rh.box_head, rh.box_predictor
    if padded:
        # Fixed-size RoI stage: S rows per image, padding rows carry label -1, counts stay on the device -> the step has NO host
        # synchronisation left (the degenerate-box flag of this batch is read, from pinned memory, when the next batch arrives).
        _defer_flag(model, flag, targets)
        rois, labels, reg_t, per_dev = D.select_training_samples_padded(rh, pb, pc, gt, glab, gvalid)
        S = rh.fg_bg_sampler.batch_size_per_image
        r0 = S * n0
        bf0 = D.roi_pool_rois(pool, features, rois[:r0], shape, n_images=n0)
        cl0, br0 = pred(head(bf0))
        with torch.no_grad():                          # the other passes' RoIs: forward only
            f_ng = OrderedDict((k, v.detach()) for k, v in features.items())
            cl1, br1 = pred(head(D.roi_pool_rois(pool, f_ng, rois[r0:], shape)))
        loss_classifier, loss_box_reg = D.fastrcnn_loss_flat(cl0, br0, labels[:r0], reg_t[:r0], n_valid=per_dev[:n0].sum())
        cl0d, br0d, training = cl0.detach(), br0.detach(), model.transform.training

        def postprocess_padded():
            from ..models.custom_generalized_transform import _ratios
            class_logits, box_regression = torch.cat([cl0d, cl1]), torch.cat([br0d, br1])
            sb, ss, sl, counts = D.postprocess_detections_padded_rois(rh, class_logits, box_regression, rois, per_dev, S, shape)
            rh_, rw_ = _ratios(shape, sizes[0][0])
            scale = _scale_tensor(rw_, rh_, sb) if not training else None
            return sb, ss, sl, counts, ((lambda b: b * scale) if scale is not None else None)
        dets = D.LazyDetections.deferred(postprocess_padded, sum(nb)).split(nb)
        losses = {"loss_classifier": loss_classifier, "loss_box_reg": loss_box_reg,
                  "loss_objectness": loss_objectness, "loss_rpn_box_reg": loss_rpn_box_reg}
        return [(losses if k == 0 else {}, d) for k, d in enumerate(dets)]
    rois, labels, reg_t, per = D.select_training_samples_batched(model.roi_heads, pb, pc, gt, glab, gvalid)
    _raise_if_degenerate(flag, targets)            # read at the step's ONE host sync (the RoI sampler's counts, just above)
    r0 = sum(per[:n0])
    bf0 = D.roi_pool_rois(pool, features, rois[:r0], shape, n_images=n0)
    cl0, br0 = pred(head(bf0))
    with torch.no_grad():                          # the other passes' RoIs: forward only
        f_ng = OrderedDict((k, v.detach()) for k, v in features.items())
        cl1, br1 = pred(head(D.roi_pool_rois(pool, f_ng, rois[r0:], shape)))
    loss_classifier, loss_box_reg = D.fastrcnn_loss_flat(cl0, br0, labels[:r0], reg_t[:r0])
    cl0d, br0d, training = cl0.detach(), br0.detach(), model.transform.training

    def postprocess():
        from ..models.custom_generalized_transform import _ratios
        class_logits, box_regression = torch.cat([cl0d, cl1]), torch.cat([br0d, br1])
        sb, ss, sl, counts = D.postprocess_detections_flat(model.roi_heads, class_logits, box_regression, rois, per, shape)
        rh, rw = _ratios(shape, sizes[0][0])
        scale = _scale_tensor(rw, rh, sb) if not training else None
        return sb, ss, sl, counts, ((lambda b: b * scale) if scale is not None else None)
    # launched by the first access or by the training step's flush() after it has issued the backward pass (LazyDetections.deferred)
    dets = D.LazyDetections.deferred(postprocess, sum(nb)).split(nb)
    losses = {"loss_classifier": loss_classifier, "loss_box_reg": loss_box_reg,
              "loss_objectness": loss_objectness, "loss_rpn_box_reg": loss_rpn_box_reg}
    return [(losses if k == 0 else {}, d) for k, d in enumerate(dets)]


_SCALE_CACHE = {}


def _scale_tensor(rw, rh, like):
    """[rw, rh, rw, rh] on the device, built once per (ratio, device): a fresh host->device copy every step is a
    blocking call that drains the launch queue."""
    key = (float(rw), float(rh), str(like.device), like.dtype)
    t = _SCALE_CACHE.get(key)
    if t is None:
        t = _SCALE_CACHE[key] = torch.tensor([rw, rh, rw, rh], dtype=like.dtype).to(like.device)
    return t


def _target_dtypes(targets):
    for t in targets:
        if not t["boxes"].dtype in (torch.float, torch.double, torch.half):
            raise TypeError(f"target boxes must of float type, instead got {t['boxes'].dtype}")
        if not t["labels"].dtype == torch.int64:
            raise TypeError(f"target labels must of int64 type, instead got {t['labels'].dtype}")


def eval_forward_fasterrcnn(model, images, targets, train_det=False, model_name='fasterrcnn'):
    """One detector pass that returns the training losses AND the detections (reference :13-68; the error texts are the reference's).
    `model.batched_heads` (default) runs the two stages in padded, batched form (`_heads_batched`); otherwise the list forms below."""
    if train_det:
        # train_detector.py:159 -- the module's mode is left as the caller set it (Lightning: train()); parameter
        # gradients need FasterRCNN.set_trainable(True) (DetectorLit does it), otherwise nothing would be learned
        if not getattr(model.backbone, "train_params", False):
            raise RuntimeError("hallucidet_amd: train_det=True needs detector.set_trainable(True) first (see "
                               "hallucidet_amd.train_detector.DetectorLit); the frozen-detector kernels emit data gradients only")
    else:
        model.eval()
    _check_targets(targets)
    sizes_in = []
    for img in images:
        hw = img.shape[-2:]
        torch._assert(len(hw) == 2, f"expecting the last two dimensions of the Tensor to be H and W instead got {img.shape[-2:]}")
        sizes_in.append((hw[0], hw[1]))
    images, targets = model.transform(images, targets)
    if targets is not None:
        _check_degenerate(targets)               # (:41-53) one fused test, the reference's message
    features = model.backbone(images.tensors)
    if isinstance(features, torch.Tensor):
        features = OrderedDict([("0", features)])
    if getattr(model, "batched_heads", False):
        rpn_losses, roi_losses, detections = _heads_batched(model, images, features, *model.rpn.head(list(features.values())), targets)
    else:
        proposals, rpn_losses = rpn_eval(model, images, features, targets)
        detections, roi_losses = roi_heads_eval(model, features, proposals, images.image_sizes, targets)
    detections = model.transform.postprocess(detections, images.image_sizes, sizes_in)
    return {**roi_losses, **rpn_losses}, detections


def _heads_batched(model, images, features, objectness, deltas, targets):
    """rpn_eval + roi_heads_eval with the per-image torchvision loops in padded, batched form
    (hallucidet_amd.models.detection: same arithmetic, same sampler call order -- asserted equal by the tests)."""
    from ..models import detection as D
    feats = list(features.values())
    anchors = model.rpn.anchor_generator(images, feats)
    napl = [o[0].numel() for o in objectness]
    obj, dl = concat_box_prediction_layers(objectness, deltas)
    shape = images.image_sizes[0]
    pb, _, pc = D.filter_proposals_padded(model.rpn, None, obj, shape, napl, deltas=dl, anchors0=anchors[0])
    if targets is None:
        raise ValueError("targets should not be None")
    _target_dtypes(targets)
    gt, glab, gvalid = D.pad_targets(targets, obj.device)
    loss_objectness, loss_rpn_box_reg = D.rpn_targets_loss_batched(model.rpn, anchors[0], gt, gvalid, obj, dl)
    rh = model.roi_heads
    rois, labels, reg_t, per = D.select_training_samples_batched(rh, pb, pc, gt, glab, gvalid)
    class_logits, box_regression = rh.box_predictor(rh.box_head(D.roi_pool_rois(rh.box_roi_pool, features, rois, shape)))
    loss_classifier, loss_box_reg = D.fastrcnn_loss_flat(class_logits, box_regression, labels, reg_t)
    sb, ss, sl, counts = D.postprocess_detections_flat(rh, class_logits, box_regression, rois, per, shape)
    dets = [{"boxes": sb[i, :c], "labels": sl[i, :c], "scores": ss[i, :c]} for i, c in enumerate(counts.tolist())]
    return ({"loss_objectness": loss_objectness, "loss_rpn_box_reg": loss_rpn_box_reg},
            {"loss_classifier": loss_classifier, "loss_box_reg": loss_box_reg}, dets)


def rpn_eval(model, images, features, targets, head_out=None):
    """List form of the RPN stage (reference :72-102) over the module's torchvision-named methods: proposals and the two RPN losses."""
    rpn = model.rpn
    feats = list(features.values())
    obj_l, dl_l = rpn.head(feats) if head_out is None else head_out
    anchors = rpn.anchor_generator(images, feats)
    per_level = [o[0].numel() for o in obj_l]
    obj, dl = concat_box_prediction_layers(obj_l, dl_l)
    decoded = rpn.box_coder.decode(dl.detach(), anchors).view(len(anchors), -1, 4)
    boxes, _ = rpn.filter_proposals(decoded, obj, images.image_sizes, per_level)
    if targets is None:
        raise ValueError("targets should not be None")
    labels, matched = rpn.assign_targets_to_anchors(anchors, targets)
    l_obj, l_reg = rpn.compute_loss(obj, dl, labels, rpn.box_coder.encode(matched, anchors))
    return boxes, {"loss_objectness": l_obj, "loss_rpn_box_reg": l_reg}


def roi_heads_eval(model, features, proposals, image_shapes, targets=None, train_det=False):
    """List form of the RoI stage (reference :105-185): the training sampler also runs at evaluation time, the box head
    sees the SAMPLED proposals, and those are what is post-processed into detections."""
    rh = model.roi_heads
    if targets is not None:
        _target_dtypes(targets)
    proposals, _, labels, reg_targets = rh.select_training_samples(proposals, targets)
    logits, regression = rh.box_predictor(rh.box_head(rh.box_roi_pool(features, proposals, image_shapes)))
    if labels is None:
        raise ValueError("labels cannot be None")
    if reg_targets is None:
        raise ValueError("regression_targets cannot be None")
    l_cls, l_box = fastrcnn_loss(logits, regression, labels, reg_targets)
    b, sc, lb = rh.postprocess_detections(logits, regression, proposals, image_shapes)
    if rh.has_keypoint():
        raise NotImplementedError("keypoint branch is dead code for fasterrcnn_resnet50_fpn (SURVEY #7)")
    return [{"boxes": x, "labels": y, "scores": z} for x, y, z in zip(b, lb, sc)], {"loss_classifier": l_cls, "loss_box_reg": l_box}
