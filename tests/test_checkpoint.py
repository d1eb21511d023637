"""Checkpoint compatibility: Lightning `.ckpt` layout in and out, bare `.bin` detector dicts, strict=False."""
import torch

from hallucidet_amd import checkpoint as ck


def _perturb(m, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for t in list(m.parameters()) + list(m.buffers()):
            if t.is_floating_point():
                t.copy_(torch.randn(t.shape, generator=g) * 0.01)


def test_encoder_decoder_lit_roundtrip_and_reference_key_layout(tmp_path):
    from hallucidet_amd.train_hallucidet import EncoderDecoderLit
    a = EncoderDecoderLit(batch_size=2, device="cpu")
    _perturb(a.encoder_decoder, 1)
    _perturb(a.detector, 2)
    path = a.save_checkpoint(str(tmp_path / "best_encoder_decoder_pl.ckpt"), epoch=3, global_step=77)
    blob = torch.load(path, map_location="cpu", weights_only=False)
    assert set(blob) >= {"state_dict", "epoch", "global_step", "pytorch-lightning_version"} and blob["epoch"] == 3
    keys = list(blob["state_dict"])
    for k in ("encoder_decoder.encoder.conv1.weight", "encoder_decoder.encoder.layer4.2.bn2.running_var",
              "encoder_decoder.decoder.blocks.0.conv1.0.weight", "encoder_decoder.decoder.blocks.4.conv2.1.bias",
              "encoder_decoder.segmentation_head.0.bias", "detector.backbone.body.layer1.0.conv1.weight",
              "detector.backbone.fpn.inner_blocks.0.weight", "detector.rpn.head.conv.weight",
              "detector.roi_heads.box_predictor.cls_score.weight"):
        assert k in keys, k
    b = EncoderDecoderLit.load_from_checkpoint(path, batch_size=2, device="cpu")
    for (k, v), (k2, v2) in zip(a.encoder_decoder.state_dict().items(), b.encoder_decoder.state_dict().items()):
        assert k == k2 and torch.equal(v, v2), k
    for (k, v), (k2, v2) in zip(a.detector.state_dict().items(), b.detector.state_dict().items()):
        assert k == k2 and torch.equal(v, v2), k


def test_detector_lit_checkpoint_bare_bin_and_torchvision_013_names(tmp_path):
    from hallucidet_amd.models.detector import Detector
    from hallucidet_amd.train_detector import DetectorLit
    a = DetectorLit(batch_size=2, pretrained=False, device="cpu")
    _perturb(a.detector, 3)
    path = a.save_checkpoint(str(tmp_path / "best.ckpt"))
    b = DetectorLit.load_from_checkpoint(path, batch_size=2, pretrained=False, device="cpu")
    assert all(torch.equal(v, b.detector.state_dict()[k]) for k, v in a.detector.state_dict().items())
    # `.bin`: bare state dict with torchvision >= 0.13 FPN / RPN key spelling (detector.py:69-79 loads such files)
    sd = {k.replace("fpn.inner_blocks.1.", "fpn.inner_blocks.1.0.").replace("rpn.head.conv.", "rpn.head.conv.0.0."): v
          for k, v in a.detector.state_dict().items()}
    binp = str(tmp_path / "detector.bin")
    torch.save(sd, binp)
    d = Detector(name="fasterrcnn", pretrained=False, n_classes=2, size=300).detector
    ck.load_detector(d, binp)
    assert all(torch.equal(v, d.state_dict()[k]) for k, v in a.detector.state_dict().items())
    # strict=False (what train_hallucidet.py / eval_hallucidet.py pass) must apply the renames BEFORE its name / shape filter
    d3 = Detector(name="fasterrcnn", pretrained=False, n_classes=2, size=300).detector
    ck.load_detector(d3, binp, strict=False)
    assert all(torch.equal(v, d3.state_dict()[k]) for k, v in a.detector.state_dict().items())
    d2 = Detector(name="fasterrcnn", pretrained=False, n_classes=2, size=300, eval_path=binp).detector     # the reference's own route
    assert torch.equal(d2.state_dict()["rpn.head.conv.weight"], a.detector.state_dict()["rpn.head.conv.weight"])


def test_strict_false_skips_missing_and_misshaped(tmp_path):
    from hallucidet_amd.train_hallucidet import EncoderDecoderLit
    a = EncoderDecoderLit(batch_size=2, device="cpu")
    _perturb(a.encoder_decoder, 5)
    sd = {"encoder_decoder." + k: v.clone() for k, v in a.encoder_decoder.state_dict().items()}
    sd["encoder_decoder.segmentation_head.0.weight"] = torch.zeros(5, 16, 3, 3)          # 5-class head from another run
    del sd["encoder_decoder.encoder.conv1.weight"]
    sd["loss_perceptual.net.weight"] = torch.zeros(3)                                     # foreign entries are ignored
    path = str(tmp_path / "partial.ckpt")
    torch.save({"state_dict": sd}, path)
    b = EncoderDecoderLit(batch_size=2, device="cpu")
    keep_head = b.encoder_decoder.segmentation_head[0].weight.detach().clone()
    keep_conv1 = b.encoder_decoder.encoder.conv1.weight.detach().clone()
    ck.load_encoder_decoder_lit(b, path, strict=False)
    assert torch.equal(b.encoder_decoder.segmentation_head[0].weight, keep_head) and torch.equal(b.encoder_decoder.encoder.conv1.weight, keep_conv1)
    assert torch.equal(b.encoder_decoder.encoder.layer1[0].conv1.weight, a.encoder_decoder.encoder.layer1[0].conv1.weight)
    import pytest
    with pytest.raises((RuntimeError, KeyError)):
        ck.load_encoder_decoder_lit(EncoderDecoderLit(batch_size=2, device="cpu"), path, strict=True)
