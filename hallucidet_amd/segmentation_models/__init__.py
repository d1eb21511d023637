"""Mirror of the reference's `src.segmentation_models` package for the hot path (Unet only)."""
from .unet import Unet, UnetDecoder, SegmentationHead, ResNetEncoder, initialize_decoder, initialize_head  # noqa: F401
