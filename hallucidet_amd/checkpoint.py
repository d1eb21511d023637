"""Checkpoint compatibility: the reference stores / restores whole LightningModules
(`EncoderDecoderLit.load_from_checkpoint`, train_hallucidet.py:467-481; `detectorLit.load_from_checkpoint(...).detector`,
:107-115 and eval_hallucidet.py:102-110,199; `trainer.save_checkpoint`, :353-356,544-545) and bare detector state dicts
(`.bin`, src/models/detector.py:69-79).  A Lightning checkpoint is a pickled dict whose 'state_dict' holds the module
tree under the attribute names: `encoder_decoder.<smp keys>` and `detector.<torchvision keys>` for EncoderDecoderLit,
`detector.<torchvision keys>` for DetectorLit.  This module reads that layout into the MI355X modules (same key names by
construction; torchvision >= 0.13 renames are mapped by the detectors' load_state_dict) and writes it back."""
from collections import OrderedDict

import torch

LIGHTNING_VERSION = "1.5.10"      # requirements.txt:65 of the reference


def read_state_dict(path):
    """-> flat state_dict of a Lightning `.ckpt` (its 'state_dict' entry) or of a bare `.bin` / `.pth` state dict."""
    try:
        blob = torch.load(path, map_location="cpu", weights_only=False)
    except TypeError:                                   # older torch without the keyword
        blob = torch.load(path, map_location="cpu")
    if isinstance(blob, dict) and "state_dict" in blob and isinstance(blob["state_dict"], dict):
        return blob["state_dict"]
    if not isinstance(blob, dict):
        raise ValueError(f"{path}: not a state dict or Lightning checkpoint")
    return blob


def sub_state_dict(sd, prefix):
    """Entries of `sd` under `prefix`, with the prefix removed (empty if none)."""
    return OrderedDict((k[len(prefix):], v) for k, v in sd.items() if k.startswith(prefix))


def _load(module, sd, what, strict):
    if not sd:
        if strict:
            raise KeyError(f"checkpoint holds no '{what}.*' entries")
        return
    own = module.state_dict()
    if hasattr(module, "remap_state_dict_keys"):      # torchvision >= 0.13 names first, THEN the name / shape filter
        sd = module.remap_state_dict_keys(sd)
    if not strict:
        kept = OrderedDict((k, v) for k, v in sd.items() if k in own and tuple(own[k].shape) == tuple(v.shape))
        dropped = [k for k in sd if k not in kept]
        missing = [k for k in own if k not in kept]
        if dropped or missing:
            import warnings
            warnings.warn("%s: strict=False skipped %d checkpoint entries (%s%s) and left %d module entries unset (%s%s)" % (
                what, len(dropped), ", ".join(dropped[:4]), " ..." if len(dropped) > 4 else "",
                len(missing), ", ".join(missing[:4]), " ..." if len(missing) > 4 else ""))
        sd = kept
    module.load_state_dict(sd, strict=strict)


def load_encoder_decoder_lit(lit, path, strict=True):
    """EncoderDecoderLit.load_from_checkpoint semantics: U-Net from `encoder_decoder.*`, detector from `detector.*`
    (`strict=False`, as the reference passes at :480, skips what is missing or mis-shaped)."""
    sd = read_state_dict(path)
    _load(lit.encoder_decoder, sub_state_dict(sd, "encoder_decoder."), "encoder_decoder", strict)
    det = sub_state_dict(sd, "detector.")
    if det or strict:
        _load(lit.detector, det, "detector", strict)
        if hasattr(lit.detector, "invalidate_packs"):
            lit.detector.invalidate_packs()
    return lit


def load_detector(detector, path, strict=True):
    """`detectorLit.load_from_checkpoint(path).detector` (Lightning .ckpt: keys `detector.*`) or a bare state dict."""
    sd = read_state_dict(path)
    det = sub_state_dict(sd, "detector.")
    _load(detector, det if det else sd, "detector", strict)
    if hasattr(detector, "invalidate_packs"):
        detector.invalidate_packs()
    return detector


def save_lightning_checkpoint(path, modules, epoch=0, global_step=0):
    """`trainer.save_checkpoint` subset: {'state_dict': {'<attr>.<key>': tensor}, 'epoch', 'global_step',
    'pytorch-lightning_version'} -- what load_from_checkpoint (here and in the reference) consumes.  `modules`: dict
    attribute name -> nn.Module (e.g. {'encoder_decoder': unet, 'detector': det})."""
    sd = OrderedDict()
    for name, m in modules.items():
        for k, v in m.state_dict().items():
            sd[f"{name}.{k}"] = v.detach().cpu().clone()
    torch.save({"state_dict": sd, "epoch": int(epoch), "global_step": int(global_step),
                "pytorch-lightning_version": LIGHTNING_VERSION}, path)
    return path
