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
        self.tensors = tensors            # [N, H, W, 8] float16 NHWC (3 real channels)
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


def resize_boxes(boxes: Tensor, original_size: List[int], new_size: List[int]) -> Tensor:
    """:325-338 -- ratios are fp32 0-dim tensors, so float64 boxes stay float64."""
    ratios = [
        torch.tensor(s, dtype=torch.float32, device=boxes.device) / torch.tensor(s_orig, dtype=torch.float32, device=boxes.device)
        for s, s_orig in zip(new_size, original_size)
    ]
    ratio_height, ratio_width = ratios
    xmin, ymin, xmax, ymax = boxes.unbind(1)
    return torch.stack((xmin * ratio_width, ymin * ratio_height, xmax * ratio_width, ymax * ratio_height), dim=1)


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
            for i in range(len(imgs)):
                targets[i]["boxes"] = resize_boxes(targets[i]["boxes"], (h, w), (Ho, Wo))
        image_sizes = [(Ho, Wo) for _ in imgs]
        return ImageList(y, image_sizes), targets

    def postprocess(self, result, image_shapes, original_image_sizes):
        if self.training:
            return result
        for i, (pred, im_s, o_im_s) in enumerate(zip(result, image_shapes, original_image_sizes)):
            result[i]["boxes"] = resize_boxes(pred["boxes"], im_s, o_im_s)
        return result

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(\n    Normalize(mean={self.image_mean}, std={self.image_std})"
                f"\n    Resize(fixed_size={self.fixed_size}, mode='nearest')\n)")
