"""Generate the golden fixtures in tests/golden/ by IMPORTING THE REFERENCE (only possible in the build container,
where /root/reference exists).  Only data (inputs / expected outputs) is written; no reference source is copied.

    python tests/golden/make_golden.py

The reference's package __init__ chains pull timm / pretrainedmodels / torchvision, none of which is installed, so the
few torch-only files on the hot path are loaded BY PATH with stub parent packages.
"""
import importlib.util
import json
import os
import sys
from types import ModuleType

import numpy as np
import torch

REF = os.environ.get("HALLUCIDET_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.path.insert(0, ROOT)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    """Returns a namespace with the reference modules that import with torch alone."""
    sm_dir = os.path.join(REF, "src", "segmentation_models")
    sm = ModuleType("segmentation_models")
    sm.__path__ = []
    base = ModuleType("segmentation_models.base")
    base.__path__ = []
    sys.modules.update({"segmentation_models": sm, "segmentation_models.base": base})
    modules = _load("segmentation_models.base.modules", os.path.join(sm_dir, "base", "modules.py"))
    heads = _load("segmentation_models.base.heads", os.path.join(sm_dir, "base", "heads.py"))
    init = _load("segmentation_models.base.initialization", os.path.join(sm_dir, "base", "initialization.py"))
    model = _load("segmentation_models.base.model", os.path.join(sm_dir, "base", "model.py"))
    base.modules, base.initialization = modules, init
    base.SegmentationHead, base.ClassificationHead, base.SegmentationModel = heads.SegmentationHead, heads.ClassificationHead, model.SegmentationModel
    sm.base = base
    dec = _load("ref_unet_decoder", os.path.join(sm_dir, "decoders", "unet", "decoder.py"))
    tv = ModuleType("torchvision")
    tv._is_tracing = lambda: False
    sys.modules.setdefault("torchvision", tv)
    tr = _load("ref_transform", os.path.join(REF, "src", "models", "custom_generalized_transform.py"))
    # eval_forward_fasterrcnn imports two torchvision names; supply the oracle's restatements as stand-ins
    from oracle import detection as od
    m1 = ModuleType("torchvision.models"); m2 = ModuleType("torchvision.models.detection")
    m3 = ModuleType("torchvision.models.detection.roi_heads"); m4 = ModuleType("torchvision.models.detection.rpn")
    m3.fastrcnn_loss = od.fastrcnn_loss
    m4.concat_box_prediction_layers = od.concat_box_prediction_layers
    sys.modules.update({"torchvision.models": m1, "torchvision.models.detection": m2,
                        "torchvision.models.detection.roi_heads": m3, "torchvision.models.detection.rpn": m4})
    glue = _load("ref_eval_forward_fasterrcnn", os.path.join(REF, "src", "utils", "eval_forward_fasterrcnn.py"))
    # eval_forward_retinanet names torchvision.ops.box_iou and (as an annotation) torchvision.models.detection._utils.BoxCoder
    from oracle import kernels as ok
    tvm = sys.modules["torchvision"]
    tvm.ops = ModuleType("torchvision.ops"); tvm.ops.box_iou = ok.box_iou
    m5 = ModuleType("torchvision.models.detection._utils"); m5.BoxCoder = od.BoxCoder
    m2._utils = m5; m1.detection = m2; tvm.models = m1
    sys.modules.update({"torchvision.ops": tvm.ops, "torchvision.models.detection._utils": m5})
    glue_retina = _load("ref_eval_forward_retinanet", os.path.join(REF, "src", "utils", "eval_forward_retinanet.py"))
    cfg = _load("ref_config", os.path.join(REF, "src", "config", "config.py"))
    ns = ModuleType("ref")
    ns.modules, ns.heads, ns.init, ns.model, ns.decoder, ns.transform, ns.glue, ns.config = modules, heads, init, model, dec, tr, glue, cfg
    ns.glue_retina = glue_retina
    ns.glue_fcos = _load("ref_eval_forward_fcos", os.path.join(REF, "src", "utils", "eval_forward_fcos.py"))     # torch-only imports
    return ns


ENC_SMALL = (3, 8, 8, 16, 24, 32)
DEC_SMALL = (24, 16, 8, 8, 8)


def gen_decoder(ref):
    """Small-channel decoder + head, fwd and bwd (train-mode BatchNorm), from the reference's own classes."""
    torch.manual_seed(123)
    d = ref.decoder.UnetDecoder(encoder_channels=ENC_SMALL, decoder_channels=DEC_SMALL, n_blocks=5, use_batchnorm=True,
                                center=False, attention_type=None)
    h = ref.heads.SegmentationHead(in_channels=DEC_SMALL[-1], out_channels=3, activation=None, kernel_size=3)
    ref.init.initialize_decoder(d)
    ref.init.initialize_head(h)
    h[-1] = torch.nn.Sigmoid()   # what encoder_decoder.py:29-30 does
    d.train()
    H, W = 64, 96
    feats = [torch.randn(2, c, H // s, W // s, requires_grad=True) for c, s in zip(ENC_SMALL, (1, 2, 4, 8, 16, 32))]
    out = h(d(*feats))
    gout = torch.randn_like(out)
    out.backward(gout)
    blob = {"out": out.detach(), "gout": gout}
    for i, f in enumerate(feats):
        blob["feat%d" % i] = f.detach()
        if i > 0:
            blob["gfeat%d" % i] = f.grad
    for k, v in d.state_dict().items():
        blob["sd.decoder." + k] = v
    for k, v in h.state_dict().items():
        blob["sd.segmentation_head." + k] = v
    for n, p in list(d.named_parameters()):
        blob["grad.decoder." + n] = p.grad
    for n, p in list(h.named_parameters()):
        blob["grad.segmentation_head." + n] = p.grad
    np.savez_compressed(os.path.join(OUT, "decoder_small.npz"), **{k: v.numpy() for k, v in blob.items()})
    # upsample_deterministic
    x = torch.arange(24.0).view(1, 2, 3, 4)
    np.savez_compressed(os.path.join(OUT, "upsample.npz"), x=x.numpy(), y=ref.decoder.upsample_deterministic(x, 2).numpy())


def gen_init_checksums(ref):
    """Full-size decoder/head initialisation under seed 123: per-parameter checksums (tiny)."""
    torch.manual_seed(123)
    d = ref.decoder.UnetDecoder(encoder_channels=(3, 64, 64, 128, 256, 512), decoder_channels=(256, 128, 64, 32, 16), n_blocks=5,
                                use_batchnorm=True, center=False, attention_type=None)
    h = ref.heads.SegmentationHead(in_channels=16, out_channels=3, activation=None, kernel_size=3)
    ref.init.initialize_decoder(d)
    ref.init.initialize_head(h)
    rec = {}
    for prefix, m in (("decoder.", d), ("segmentation_head.", h)):
        for k, v in m.state_dict().items():
            v = v.double()
            rec[prefix + k] = {"shape": list(v.shape), "sum": float(v.sum()), "abssum": float(v.abs().sum()),
                               "head": [float(t) for t in v.flatten()[:4]]}
    rec["_n_params"] = {"decoder": sum(p.numel() for p in d.parameters()), "head": sum(p.numel() for p in h.parameters())}
    json.dump(rec, open(os.path.join(OUT, "init_checksums.json"), "w"), indent=1, sort_keys=True)


def gen_shape_error(ref):
    class M(ref.model.SegmentationModel):
        pass
    m = M()
    class E:
        output_stride = 32
    m.encoder = E()
    try:
        m.check_input_shape(torch.zeros(1, 3, 500, 640))
    except RuntimeError as e:
        json.dump({"shape": [500, 640], "message": str(e)}, open(os.path.join(OUT, "shape_error.json"), "w"))


def gen_transform(ref):
    T = ref.transform.CustomGeneralizedRCNNTransform
    t = T(min_size=300, max_size=300, image_mean=[0.0], image_std=[1.0], size_divisible=1, fixed_size=(300, 300)).eval()
    # index image: value = linear source index (exact in fp32)
    idx = (torch.arange(512 * 640, dtype=torch.float32).view(1, 512, 640)).repeat(3, 1, 1)
    boxes = torch.tensor([[10.0, 20.0, 110.0, 220.0], [300.5, 17.25, 639.0, 511.0], [0.0, 0.0, 640.0, 512.0]])
    il, tg = t([idx, idx.flip(-1)], [{"boxes": boxes, "labels": torch.ones(3, dtype=torch.int64)},
                                     {"boxes": boxes[:1], "labels": torch.ones(1, dtype=torch.int64)}])
    post = t.postprocess([{"boxes": tg[0]["boxes"].clone()}, {"boxes": tg[1]["boxes"].clone()}], il.image_sizes, [(512, 640), (512, 640)])
    # small value case with another fixed size and float64 boxes (test-loader dtype quirk)
    t2 = T(min_size=24, max_size=24, image_mean=[0.0], image_std=[1.0], size_divisible=1, fixed_size=(24, 24)).eval()
    torch.manual_seed(5)
    img = torch.rand(3, 40, 56)
    b64 = torch.tensor([[1.5, 2.5, 30.0, 33.0]], dtype=torch.float64)
    il2, tg2 = t2([img], [{"boxes": b64, "labels": torch.ones(1, dtype=torch.int64)}])
    np.savez_compressed(os.path.join(OUT, "transform.npz"),
                        src_index=il.tensors[0, 0].to(torch.int32).numpy(), src_index_flipped=il.tensors[1, 0].to(torch.int32).numpy(),
                        image_sizes=np.array(il.image_sizes), boxes_in=boxes.numpy(), boxes_out0=tg[0]["boxes"].numpy(),
                        boxes_out1=tg[1]["boxes"].numpy(), post0=post[0]["boxes"].numpy(), post1=post[1]["boxes"].numpy(),
                        small_img=img.numpy(), small_out=il2.tensors.numpy(), small_boxes_in=b64.numpy(),
                        small_boxes_out=tg2[0]["boxes"].numpy(), small_boxes_out_dtype=str(tg2[0]["boxes"].dtype))


def gen_config(ref):
    import argparse
    old = sys.argv
    sys.argv = ["x"]
    try:
        args = ref.config.Config.argument_parser()
    finally:
        sys.argv = old
    C = ref.config.Config
    rec = {"args": {k: (v if isinstance(v, (int, float, str, bool, type(None))) else str(v)) for k, v in vars(args).items()},
           "loss_weights": dict(C.Losses.hparams_losses_weights), "optimizer_name": C.Optimizer.name,
           "n_gpus": C.Environment.N_GPUS, "decoder_head": C.EncoderDecoder.decoder_head}
    json.dump(rec, open(os.path.join(OUT, "config_defaults.json"), "w"), indent=1, sort_keys=True)


def make_detector_case(seed=7, n_img=2, H=96, W=128):
    """Shared by the generator and the tests: oracle detector (seeded), inputs and the injected sampler permutations."""
    from oracle import detection as od
    torch.manual_seed(seed)
    perm_log = []
    g = torch.Generator().manual_seed(seed + 1)

    def randperm_fn(n):
        p = torch.randperm(n, generator=g)
        perm_log.append(p)
        return p

    model = od.FasterRCNN(num_classes=2, size=300, randperm_fn=randperm_fn)
    tame_detector_(model)
    images = torch.rand(n_img, 3, H, W)
    targets = []
    for i in range(n_img):
        k = 1 + i
        xy = torch.rand(k, 2) * torch.tensor([W * 0.5, H * 0.5])
        wh = torch.rand(k, 2) * torch.tensor([W * 0.3, H * 0.4]) + 8.0
        targets.append({"boxes": torch.cat([xy, xy + wh], 1), "labels": torch.ones(k, dtype=torch.int64)})
    return model, images, targets, perm_log, g


def tame_detector_(model):
    """Random FrozenBN (weight 1, mean 0, var 1) lets activations grow through 16 residual blocks; damp the last BN of
    every bottleneck so the synthetic detector stays in a numerically ordinary range (deterministic, seed-free)."""
    from oracle import detection as od
    for m in model.modules():
        if isinstance(m, od.Bottleneck):
            m.bn3.weight.fill_(0.25)
            m.bn1.weight.fill_(0.9)


def gen_glue(ref):
    """Drive the REFERENCE's eval_forward_fasterrcnn.py over the oracle's duck-typed detector: pins call order,
    argument contract, loss keys and detection post-processing of the reference glue."""
    model, images, targets, perm_log, g = make_detector_case()
    losses, dets = ref.glue.eval_forward_fasterrcnn(model, images, targets, train_det=False)
    blob = {"images": images, "n_perm": torch.tensor(len(perm_log))}
    for i, t in enumerate(targets):
        blob["t%d.boxes" % i] = t["boxes"]
    for k, v in losses.items():
        blob["loss." + k] = v.detach()
    for i, d in enumerate(dets):
        for k, v in d.items():
            blob["det%d.%s" % (i, k)] = v.detach()
    np.savez_compressed(os.path.join(OUT, "glue_fasterrcnn.npz"), **{k: v.numpy() for k, v in blob.items()})


def make_retinanet_case(seed=13, n_img=2, H=96, W=128):
    """Oracle RetinaNet (seeded) + inputs; image 1 has NO boxes (the reference's empty-target branch, :169-172)."""
    from oracle import retinanet as orn
    torch.manual_seed(seed)
    model = orn.RetinaNet(num_classes=2, size=300)
    tame_detector_(model)
    with torch.no_grad():   # prior bias -log(99) keeps every score < 0.05: spread the logits so post-processing has work
        model.head.classification_head.cls_logits.weight.normal_(0, 0.05)
        model.head.classification_head.cls_logits.bias.fill_(-2.0)
        model.head.regression_head.bbox_reg.weight.normal_(0, 0.02)
    images = torch.rand(n_img, 3, H, W)
    targets = []
    for i in range(n_img):
        k = 2 if i == 0 else 0
        xy = torch.rand(k, 2) * torch.tensor([W * 0.5, H * 0.5])
        wh = torch.rand(k, 2) * torch.tensor([W * 0.3, H * 0.4]) + 8.0
        targets.append({"boxes": torch.cat([xy, xy + wh], 1).reshape(-1, 4), "labels": torch.ones(k, dtype=torch.int64)})
    return model, images, targets


def gen_glue_retinanet(ref):
    """The REFERENCE's eval_forward_retinanet.py (its own focal loss / box loss / compute_retinanet_loss) driving the
    oracle's duck-typed RetinaNet; plus direct known-answer vectors of the two loss functions."""
    model, images, targets = make_retinanet_case()
    losses, dets = ref.glue_retina.eval_forward_retinanet(model, images, targets, train_det=False)
    blob = {"images": images}
    for k, v in losses.items():
        blob["loss." + k] = v.detach()
    for i, d in enumerate(dets):
        for k, v in d.items():
            blob["det%d.%s" % (i, k)] = v.detach()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(37, 2, generator=g) * 3
    t = (torch.rand(37, 2, generator=g) > 0.7).float()
    blob["focal.x"], blob["focal.t"] = x, t
    blob["focal.none"] = ref.glue_retina.sigmoid_focal_loss(x, t)
    blob["focal.sum"] = ref.glue_retina.sigmoid_focal_loss(x, t, reduction="sum")
    blob["focal.mean_a-1_g0"] = ref.glue_retina.sigmoid_focal_loss(x, t, alpha=-1, gamma=0, reduction="mean")
    from oracle import detection as od
    anchors = torch.tensor([[0., 0., 20., 30.], [5., 5., 45., 25.], [10., 2., 30., 42.]])
    gts = torch.tensor([[1., 2., 22., 28.], [0., 9., 50., 20.], [12., 0., 27., 47.]])
    breg = torch.randn(3, 4, generator=g)
    blob["boxloss.anchors"], blob["boxloss.gts"], blob["boxloss.breg"] = anchors, gts, breg
    blob["boxloss.l1"] = ref.glue_retina.box_loss("l1", od.BoxCoder((1.0,) * 4), anchors, gts, breg)
    blob["boxloss.smooth_l1"] = ref.glue_retina.box_loss("smooth_l1", od.BoxCoder((1.0,) * 4), anchors, gts, breg)
    np.savez_compressed(os.path.join(OUT, "glue_retinanet.npz"), **{k: v.numpy() for k, v in blob.items()})


def make_fcos_case(seed=17, n_img=3, H=96, W=128):
    """Oracle FCOS (seeded) + inputs; image 1 has NO boxes (FCOS.compute_loss's all -1 branch), image 2 has two nested boxes (the
    smallest-area rule decides)."""
    from oracle import fcos as ofc
    torch.manual_seed(seed)
    model = ofc.FCOS(num_classes=2, size=300)
    tame_detector_(model)
    with torch.no_grad():   # the prior bias -log(99) keeps every score below 0.2: spread the outputs so post-processing has work
        model.head.classification_head.cls_logits.weight.normal_(0, 0.05)
        model.head.classification_head.cls_logits.bias.fill_(0.5)
        model.head.regression_head.bbox_reg.weight.normal_(0, 0.03)
        model.head.regression_head.bbox_reg.bias.fill_(0.8)
        model.head.regression_head.bbox_ctrness.weight.normal_(0, 0.05)
    images = torch.rand(n_img, 3, H, W)
    targets = []
    for i in range(n_img):
        if i == 1:
            boxes = torch.zeros(0, 4)
        elif i == 2:
            boxes = torch.tensor([[10.0, 8.0, 118.0, 90.0], [40.0, 30.0, 90.0, 70.0]])
        else:
            xy = torch.rand(3, 2) * torch.tensor([W * 0.5, H * 0.5])
            wh = torch.rand(3, 2) * torch.tensor([W * 0.4, H * 0.4]) + 10.0
            boxes = torch.cat([xy, xy + wh], 1)
        targets.append({"boxes": boxes.reshape(-1, 4), "labels": torch.ones(boxes.shape[0], dtype=torch.int64)})
    return model, images, targets


def gen_glue_fcos(ref):
    """The REFERENCE's eval_forward_fcos.py driving the oracle's duck-typed FCOS: pins the call sequence, the argument
    contract of compute_loss / postprocess_detections, the per-level split and the returned keys."""
    model, images, targets = make_fcos_case()
    losses, dets = ref.glue_fcos.eval_forward_fcos(model, images, targets, train_det=False)
    blob = {"images": images}
    for k, v in losses.items():
        blob["loss." + k] = v.detach()
    for i, d in enumerate(dets):
        for k, v in d.items():
            blob["det%d.%s" % (i, k)] = v.detach()
    np.savez_compressed(os.path.join(OUT, "glue_fcos.npz"), **{k: v.numpy() for k, v in blob.items()})


def main():
    ref = load_reference()
    gen_decoder(ref)
    gen_init_checksums(ref)
    gen_shape_error(ref)
    gen_transform(ref)
    gen_config(ref)
    gen_glue(ref)
    gen_glue_retinanet(ref)
    gen_glue_fcos(ref)
    for f in sorted(os.listdir(OUT)):
        if f.endswith((".npz", ".json")):
            print("%-28s %8d bytes" % (f, os.path.getsize(os.path.join(OUT, f))))


if __name__ == "__main__":
    main()
