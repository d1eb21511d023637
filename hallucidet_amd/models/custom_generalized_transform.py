"""Detector input transform (reference src/models/custom_generalized_transform.py:103-304).

Same class name, constructor keywords, `forward(images, targets) -> (ImageList, targets)`, `postprocess`,
`resize_boxes` semantics -- but the per-image python loop (normalize -> F.interpolate(mode default = 'nearest') ->
batch_images) is ONE fused kernel: NCHW fp32 -> nearest resize -> NHWC fp16 (hd_nchw_to_nhwc_resize), differentiable
through hd_nchw_to_nhwc_resize_bwd so the detector loss reaches the hallucination network.

`ImageList.tensors` holds the batched detector input in the layout the HIP backbone consumes
(NHWC fp16, channels padded 3 -> 8); `ImageList.image_sizes` is the reference's list of (H, W).
"""
from typing import Any, Dict, List, Optional, Tuple

import torch
from torch import nn, Tensor

from .. import ops


class ImageList:
    def __init__(self, tensors: Tensor, image_sizes: List[Tuple[int, int]]) -> None:
        self.tensors = tensors            # [N, H, W, 8] NHWC in the storage type (float16; float32 with precision=32), 3 real channels
        self.image_sizes = image_sizes
        self.layout = "nhwc8_f16"

    def to(self, device) -> "ImageList":
        return ImageList(self.tensors.to(device), self.image_sizes)


class _ResizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo):
        ctx.shape = tuple(x.shape)
        return ops.nchw_to_nhwc_resize(x, Ho, Wo, 8)

    @staticmethod
    def backward(ctx, dy):
        N, C, H, W = ctx.shape
        return ops.nchw_to_nhwc_resize_bwd(dy.contiguous(), N, C, H, W, 1.0), None, None


class _ResizeManyFn(torch.autograd.Function):
    """Several image batches -> ONE [sum(n), Ho, Wo, 8] fp16 NHWC tensor, each batch resized straight into its slice (the fused
    three-pass evaluation used to concatenate the fp32 batches first: 94 MB read + written per step); only batches that require a
    gradient get one."""

    @staticmethod
    def forward(ctx, Ho, Wo, *xs):
        ctx.shapes = [tuple(x.shape) for x in xs]
        n = sum(s[0] for s in ctx.shapes)
        y = torch.empty((n, Ho, Wo, 8), dtype=ops.act_dtype(), device=xs[0].device)
        lo = 0
        for x in xs:
            ops.nchw_to_nhwc_resize(x, Ho, Wo, 8, out=y[lo:lo + x.shape[0]])
            lo += x.shape[0]
        return y

    @staticmethod
    def backward(ctx, dy):
        grads, lo = [], 0
        for i, (N, C, H, W) in enumerate(ctx.shapes):
            grads.append(ops.nchw_to_nhwc_resize_bwd(dy[lo:lo + N].contiguous(), N, C, H, W, 1.0) if ctx.needs_input_grad[2 + i] else None)
            lo += N
        return (None, None) + tuple(grads)


_RATIO_CACHE = {}
_SCALE_VEC = {}


def _ratios(original_size, new_size):
    """(ratio_height, ratio_width) exactly as the reference computes them -- fp32(new) / fp32(orig) (:326-330) -- but as
    cached python floats: the reference builds two 0-dim device tensors per image per call (a blocking host->device copy
    each; 15 ms of host time per training step at batch 8)."""
    key = (int(original_size[0]), int(original_size[1]), int(new_size[0]), int(new_size[1]))
    r = _RATIO_CACHE.get(key)
    if r is None:
        r = tuple(float(torch.tensor(s, dtype=torch.float32) / torch.tensor(o, dtype=torch.float32))
                  for s, o in zip(key[2:], key[:2]))
        _RATIO_CACHE[key] = r
    return r


def resize_boxes(boxes: Tensor, original_size: List[int], new_size: List[int]) -> Tensor:
    """:325-338.  A python scalar holding the fp32 ratio multiplies fp32 boxes in fp32 and float64 boxes in float64 --
    the same values the reference's 0-dim fp32 tensor gives (float64 boxes stay float64)."""
    ratio_height, ratio_width = _ratios(original_size, new_size)
    if boxes.dtype == torch.float32 and boxes.is_cuda:
        # one launch: fp32 boxes times the fp32 ratios is the same arithmetic as the four scalar multiplies + stack below
        key = (ratio_width, ratio_height, str(boxes.device))
        sc = _SCALE_VEC.get(key)
        if sc is None:
            sc = _SCALE_VEC[key] = torch.tensor([ratio_width, ratio_height, ratio_width, ratio_height], dtype=torch.float32, device=boxes.device)
        return boxes * sc
    xmin, ymin, xmax, ymax = boxes.unbind(1)
    return torch.stack((xmin * ratio_width, ymin * ratio_height, xmax * ratio_width, ymax * ratio_height), dim=1)


def resize_boxes_many(box_list, original_size, new_size):
    """resize_boxes over a list of per-image tensors that share sizes: one scaling, then views."""
    if len(box_list) == 0:
        return []
    if len({b.dtype for b in box_list}) != 1:
        return [resize_boxes(b, original_size, new_size) for b in box_list]
    n = [int(b.shape[0]) for b in box_list]
    out = resize_boxes(torch.cat([b.reshape(-1, 4) for b in box_list], dim=0), original_size, new_size)
    return list(out.split(n, 0))


class CustomGeneralizedRCNNTransform(nn.Module):
    def __init__(self, min_size: int, max_size: int, image_mean: List[float], image_std: List[float],
                 size_divisible: int = 32, fixed_size: Optional[Tuple[int, int]] = None, **kwargs: Any):
        super().__init__()
        if not isinstance(min_size, (list, tuple)):
            min_size = (min_size,)
        self.min_size, self.max_size = min_size, max_size
        self.image_mean, self.image_std = image_mean, image_std
        self.size_divisible, self.fixed_size = size_divisible, fixed_size
        self._skip_resize = kwargs.pop("_skip_resize", False)
        if fixed_size is None:
            raise NotImplementedError("hallucidet_amd: the hot path always passes fixed_size (detector.py:43-48)")
        if any(float(m) != 0.0 for m in image_mean) or any(float(s) != 1.0 for s in image_std):
            raise NotImplementedError("hallucidet_amd: fused transform implements mean 0 / std 1 (detector.py:45-46), "
                                      "for which normalize is the identity")

    def forward(self, images, targets: Optional[List[Dict[str, Tensor]]] = None):
        imgs = [img for img in images]
        for image in imgs:
            if image.dim() != 3:
                raise ValueError(f"images is expected to be a list of 3d tensors of shape [C, H, W], got {image.shape}")
            if not image.is_floating_point():
                raise TypeError(f"Expected input images to be of floating type (in range [0, 1]), "
                                f"but found type {image.dtype} instead")
        if targets is not None:
            targets = [{k: v for k, v in t.items()} for t in targets]
        same = all(im.shape == imgs[0].shape for im in imgs)
        if not same:
            raise NotImplementedError("hallucidet_amd: fused transform needs equally sized images in a batch "
                                      "(LLVIP / FLIR batches are; Utils.stack_images already requires it)")
        x = images if isinstance(images, Tensor) and images.dim() == 4 else torch.stack(imgs)
        x = x.float().contiguous()
        h, w = x.shape[-2:]
        Wo, Ho = self.fixed_size[0], self.fixed_size[1]           # size = [fixed_size[1], fixed_size[0]] (:65-66)
        y = _ResizeFn.apply(x, Ho, Wo)
        if targets is not None:
            for t, b in zip(targets, resize_boxes_many([t["boxes"] for t in targets], (h, w), (Ho, Wo))):
                t["boxes"] = b
        image_sizes = [(Ho, Wo) for _ in imgs]
        return ImageList(y, image_sizes), targets

    def forward_batches(self, image_batches, targets):
        """`forward` for several equally sized [n_k, C, H, W] batches at once (the hallucinated / RGB / IR passes of a step) without
        concatenating them: -> (ImageList over sum(n_k) images, resized targets)."""
        xs = []
        for b in image_batches:
            b = b if isinstance(b, Tensor) and b.dim() == 4 else torch.stack(list(b))
            if not b.is_floating_point():
                raise TypeError(f"Expected input images to be of floating type (in range [0, 1]), but found type {b.dtype} instead")
            if b.shape[1:] != (image_batches[0].shape[1:] if isinstance(image_batches[0], Tensor) else b.shape[1:]):
                raise NotImplementedError("hallucidet_amd: fused transform needs equally sized images in a batch")
            xs.append(ops.as_dense_planes_f32(b))
        h, w = xs[0].shape[-2:]
        Wo, Ho = self.fixed_size[0], self.fixed_size[1]
        y = _ResizeManyFn.apply(Ho, Wo, *xs)
        if targets is not None:
            targets = [{k: v for k, v in t.items()} for t in targets]
            for t, b in zip(targets, resize_boxes_many([t["boxes"] for t in targets], (h, w), (Ho, Wo))):
                t["boxes"] = b
        return ImageList(y, [(Ho, Wo)] * y.shape[0]), targets

    def postprocess(self, result, image_shapes, original_image_sizes):
        if self.training:
            return result
        if len(set(map(tuple, image_shapes))) == 1 and len(set(map(tuple, original_image_sizes))) == 1 and len(result):
            for r, b in zip(result, resize_boxes_many([p["boxes"] for p in result], image_shapes[0], original_image_sizes[0])):
                r["boxes"] = b
            return result
        for i, (pred, im_s, o_im_s) in enumerate(zip(result, image_shapes, original_image_sizes)):
            result[i]["boxes"] = resize_boxes(pred["boxes"], im_s, o_im_s)
        return result

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(\n    Normalize(mean={self.image_mean}, std={self.image_std})"
                f"\n    Resize(fixed_size={self.fixed_size}, mode='nearest')\n)")
