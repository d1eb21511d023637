"""Hot-path helpers of the reference's `Utils` (src/utils/utils.py:13-53,237-254,333-338): same names and semantics.
Plotting / VOC-XML helpers of that class are outside the hot path (SURVEY #8)."""
import torch


_IR_VIEW = __import__("os").environ.get("HD_IR_VIEW", "1") != "0"      # A/B knob: 0 = materialise the repeat


class Utils():

    @staticmethod
    def stack_images(imgs, device='cpu', ablation_flag=False):
        if isinstance(imgs, torch.Tensor) and imgs.dim() == 4:        # already batched (synthetic / pre-staged input)
            return imgs.to(device, dtype=torch.float) if ablation_flag else imgs.to(device)
        if ablation_flag:
            return torch.stack(list(image.to(device, dtype=torch.float) for image in imgs))
        return torch.stack(list(image.to(device) for image in imgs))

    @staticmethod
    def batch_images_for_encoder_decoder(imgs, device='cpu', ablation_flag=False):
        return Utils.stack_images(imgs=imgs, device=device, ablation_flag=ablation_flag)

    @staticmethod
    def list_targets(targets, device='cpu', detach=False, detector_name='fasterrcnn'):
        """utils.py:25-42: every tensor of every target dict moves to `device` (detached first when asked), strings pass through;
        FCOS takes its boxes as fp32 (the reference casts before the move, so does this)."""
        boxes_as_float = 'fcos' in detector_name

        def _to(key, value):
            if isinstance(value, str):
                return value
            if boxes_as_float and key == 'boxes':
                value = value.float()
            if detach:
                value = value.detach()
            return value.to(device)

        return [{key: _to(key, value) for key, value in target.items()} for target in targets]

    @staticmethod
    def batch_targets_for_detector(targets, device='cpu', detach=False, detector_name='fasterrcnn'):
        return Utils.list_targets(targets=targets, device=device, detach=detach, detector_name=detector_name)

    @staticmethod
    def expand_one_channel_to_output_channels(imgs, output_channels=3):
        """utils.py:52-53 `imgs.repeat(1, C, 1, 1)`.  A single-channel batch is returned as its stride-0 channel VIEW (same shape, same
        values, read-only use downstream): the U-Net's and the detector transform's first kernel read the one plane three times
        instead of 31 MB being written and read back every step."""
        if imgs.dim() == 4 and imgs.shape[1] == 1 and _IR_VIEW:
            return imgs.expand(-1, output_channels, -1, -1)
        return imgs.repeat(1, output_channels, 1, 1)

    @staticmethod
    def concat_modalities(img_rgb, img_ir):
        return torch.cat([img_rgb, img_ir], dim=0)

    @staticmethod
    def collate_fn(batch):
        return tuple(zip(*batch))

    @staticmethod
    def normalize_image(image):
        """Per-channel min-max to [0,1] (constant channels -> 0), in place, as utils.py:237-248 -- vectorised: no host
        round trip per channel (the reference's `if` on a device scalar syncs 3x per image)."""
        mins = image.amin(dim=(1, 2), keepdim=True)
        maxs = image.amax(dim=(1, 2), keepdim=True)
        rng = maxs - mins
        out = torch.where(rng != 0, (image - mins) / torch.where(rng != 0, rng, torch.ones_like(rng)), torch.zeros_like(image))
        image.copy_(out)
        return image

    @staticmethod
    def normalize_batch_images(images):
        mins = images.amin(dim=(2, 3), keepdim=True)
        maxs = images.amax(dim=(2, 3), keepdim=True)
        rng = maxs - mins
        images.copy_(torch.where(rng != 0, (images - mins) / torch.where(rng != 0, rng, torch.ones_like(rng)), torch.zeros_like(images)))
        return images

    @staticmethod
    def filter_dictionary(input_dict, filter_keys):
        return {key: value for key, value in input_dict.items() if key in filter_keys}
