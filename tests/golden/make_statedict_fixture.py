"""Writes tests/golden/state_dict_layouts.json: the parameter / buffer NAMES and SHAPES of the four networks whose checkpoints the
reference exchanges, and the top-level keys of a PyTorch-Lightning 1.5.10 checkpoint -- stated here from the PUBLISHED
definitions of the third-party packages the reference pins (requirements.txt: torchvision 0.12.0, segmentation-models-pytorch
(vendored under src/segmentation_models), pytorch-lightning 1.5.10), NOT from this repository's modules, so that the fixture is an
independent statement the product's `state_dict()` can be held against (reference call sites:
train_hallucidet.py:107-115,353-356,467-481,544-545, eval_hallucidet.py:102-110,199, src/models/detector.py:51-79).

    python tests/golden/make_statedict_fixture.py

Sources restated (no torchvision / smp / lightning import: none of them is installed here):
  * torchvision.models.resnet.ResNet(Bottleneck, [3,4,6,3]) with norm_layer=FrozenBatchNorm2d (buffers weight, bias, running_mean,
    running_var -- no num_batches_tracked), v1.5 stride placement; torchvision.ops.FeaturePyramidNetwork (0.12 names:
    inner_blocks.{i}.weight, layer_blocks.{i}.weight; >= 0.13 wrap them in Conv2dNormActivation: inner_blocks.{i}.0.weight);
    torchvision.models.detection.{faster_rcnn, rpn, roi_heads, retinanet, fcos} module trees; the reference re-heads every detector
    to n_classes = 2 (src/models/detector.py:51-66);
  * segmentation_models_pytorch.Unet('resnet34'): ResNetEncoder = torchvision ResNet(BasicBlock, [3,4,6,3]) without fc, UnetDecoder
    blocks conv1 / conv2 = Conv2dReLU = Sequential(Conv2d(bias=False), BatchNorm2d, ReLU), SegmentationHead = Sequential(Conv2d,
    Identity upsampling, activation) (src/segmentation_models/decoders/unet/{model,decoder}.py, base/{modules,heads}.py);
  * pytorch_lightning 1.5.10 Trainer.save_checkpoint -> CheckpointConnector.dump_checkpoint: top-level keys of a full checkpoint of
    a module trained with precision=16 (native AMP) whose __init__ calls save_hyperparameters() or not (the reference's modules do
    not: no 'hyper_parameters').
"""
import json
import os
from collections import OrderedDict


def frozen_bn(prefix, c, out):
    for k in ("weight", "bias", "running_mean", "running_var"):
        out[prefix + "." + k] = [c]


def batch_norm(prefix, c, out):
    for k in ("weight", "bias", "running_mean", "running_var"):
        out[prefix + "." + k] = [c]
    out[prefix + ".num_batches_tracked"] = []


def conv(prefix, cout, cin, k, out, bias=False):
    out[prefix + ".weight"] = [cout, cin, k, k]
    if bias:
        out[prefix + ".bias"] = [cout]


def resnet50_body(prefix, out):
    conv(prefix + "conv1", 64, 3, 7, out)
    frozen_bn(prefix + "bn1", 64, out)
    cin = 64
    for li, (n, w) in enumerate(zip((3, 4, 6, 3), (64, 128, 256, 512)), start=1):
        for b in range(n):
            p = "%slayer%d.%d." % (prefix, li, b)
            conv(p + "conv1", w, cin, 1, out); frozen_bn(p + "bn1", w, out)
            conv(p + "conv2", w, w, 3, out); frozen_bn(p + "bn2", w, out)
            conv(p + "conv3", 4 * w, w, 1, out); frozen_bn(p + "bn3", 4 * w, out)
            if b == 0:                                   # stride != 1 or cin != 4w: true for the first block of every stage
                conv(p + "downsample.0", 4 * w, cin, 1, out); frozen_bn(p + "downsample.1", 4 * w, out)
            cin = 4 * w


def fpn(prefix, in_channels, out, extra=None):
    for i, c in enumerate(in_channels):
        conv("%sinner_blocks.%d" % (prefix, i), 256, c, 1, out, bias=True)
    for i, _ in enumerate(in_channels):
        conv("%slayer_blocks.%d" % (prefix, i), 256, 256, 3, out, bias=True)
    if extra == "p6p7":
        conv(prefix + "extra_blocks.p6", 256, 256, 3, out, bias=True)
        conv(prefix + "extra_blocks.p7", 256, 256, 3, out, bias=True)


def fasterrcnn_resnet50_fpn(num_classes=2):
    o = OrderedDict()
    resnet50_body("backbone.body.", o)
    fpn("backbone.fpn.", (256, 512, 1024, 2048), o)                # LastLevelMaxPool has no parameters
    conv("rpn.head.conv", 256, 256, 3, o, bias=True)
    conv("rpn.head.cls_logits", 3, 256, 1, o, bias=True)
    conv("rpn.head.bbox_pred", 12, 256, 1, o, bias=True)
    o["roi_heads.box_head.fc6.weight"], o["roi_heads.box_head.fc6.bias"] = [1024, 256 * 7 * 7], [1024]
    o["roi_heads.box_head.fc7.weight"], o["roi_heads.box_head.fc7.bias"] = [1024, 1024], [1024]
    o["roi_heads.box_predictor.cls_score.weight"], o["roi_heads.box_predictor.cls_score.bias"] = [num_classes, 1024], [num_classes]
    o["roi_heads.box_predictor.bbox_pred.weight"], o["roi_heads.box_predictor.bbox_pred.bias"] = [4 * num_classes, 1024], [4 * num_classes]
    return o


def retinanet_resnet50_fpn(num_classes=2):
    o = OrderedDict()
    resnet50_body("backbone.body.", o)
    fpn("backbone.fpn.", (512, 1024, 2048), o, extra="p6p7")
    for i in (0, 2, 4, 6):                                           # Sequential(conv, ReLU) x 4
        conv("head.classification_head.conv.%d" % i, 256, 256, 3, o, bias=True)
    conv("head.classification_head.cls_logits", 9 * num_classes, 256, 3, o, bias=True)
    for i in (0, 2, 4, 6):
        conv("head.regression_head.conv.%d" % i, 256, 256, 3, o, bias=True)
    conv("head.regression_head.bbox_reg", 36, 256, 3, o, bias=True)
    return o


def fcos_resnet50_fpn(num_classes=2):
    o = OrderedDict()
    resnet50_body("backbone.body.", o)
    fpn("backbone.fpn.", (512, 1024, 2048), o, extra="p6p7")

    def tower(prefix):
        for i in range(4):                                           # Sequential(conv, GroupNorm(32, 256), ReLU) x 4
            conv("%sconv.%d" % (prefix, 3 * i), 256, 256, 3, o, bias=True)
            o["%sconv.%d.weight" % (prefix, 3 * i + 1)], o["%sconv.%d.bias" % (prefix, 3 * i + 1)] = [256], [256]
    tower("head.classification_head.")
    conv("head.classification_head.cls_logits", num_classes, 256, 3, o, bias=True)
    tower("head.regression_head.")
    conv("head.regression_head.bbox_reg", 4, 256, 3, o, bias=True)
    conv("head.regression_head.bbox_ctrness", 1, 256, 3, o, bias=True)
    return o


def smp_unet_resnet34(classes=3):
    o = OrderedDict()
    conv("encoder.conv1", 64, 3, 7, o)
    batch_norm("encoder.bn1", 64, o)
    cin = 64
    for li, (n, w) in enumerate(zip((3, 4, 6, 3), (64, 128, 256, 512)), start=1):
        for b in range(n):
            p = "encoder.layer%d.%d." % (li, b)
            conv(p + "conv1", w, cin, 3, o); batch_norm(p + "bn1", w, o)
            conv(p + "conv2", w, w, 3, o); batch_norm(p + "bn2", w, o)
            if b == 0 and li > 1:                        # BasicBlock: downsample only where the stride is 2 (cin != w)
                conv(p + "downsample.0", w, cin, 1, o); batch_norm(p + "downsample.1", w, o)
            cin = w
    enc = (512, 256, 128, 64, 64)                        # encoder_channels[1:][::-1]
    dec = (256, 128, 64, 32, 16)
    cins = (enc[0],) + dec[:-1]
    skips = enc[1:] + (0,)
    for i, (ci, cs, co) in enumerate(zip(cins, skips, dec)):
        p = "decoder.blocks.%d." % i
        conv(p + "conv1.0", co, ci + cs, 3, o); batch_norm(p + "conv1.1", co, o)
        conv(p + "conv2.0", co, co, 3, o); batch_norm(p + "conv2.1", co, o)
    conv("segmentation_head.0", classes, 16, 3, o, bias=True)
    return o


LIGHTNING_1_5_10 = {
    # CheckpointConnector.dump_checkpoint(weights_only=False), Trainer(precision=16, amp_backend='native'), one optimizer
    "always": ["epoch", "global_step", "pytorch-lightning_version", "state_dict", "loops"],
    "full_checkpoint": ["callbacks", "optimizer_states", "lr_schedulers", "native_amp_scaling_state"],
    "with_save_hyperparameters": ["hparams_name", "hyper_parameters"],
    "version": "1.5.10",
    "module_prefixes": {"EncoderDecoderLit": ["encoder_decoder.", "detector."], "DetectorLit": ["detector."]},
}


def main():
    out = {"torchvision_0_12": {"fasterrcnn_resnet50_fpn": fasterrcnn_resnet50_fpn(), "retinanet_resnet50_fpn": retinanet_resnet50_fpn(),
                                "fcos_resnet50_fpn": fcos_resnet50_fpn()},
           "smp_unet_resnet34": smp_unet_resnet34(), "lightning": LIGHTNING_1_5_10}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "state_dict_layouts.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print(path, {k: len(v) for k, v in out["torchvision_0_12"].items()}, len(out["smp_unet_resnet34"]))


if __name__ == "__main__":
    main()
