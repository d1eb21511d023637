"""Detector fine-tuning path (reference train_detector.py:147-203 -> Detector.calculate_loss(..., train_det=True); SURVEY
a20, BASELINE configs[4]): the hand-written backward chains must emit the PARAMETER gradients torch autograd gives on the
CPU oracle, for exactly the parameters torchvision leaves trainable, and DetectorLit.fit_step must learn with them."""
import pytest
import torch

from oracle import detection as od
from oracle import unet as ou
from test_detector_gpu import nchw, _t2d

pytestmark = pytest.mark.gpu

TRAINABLE_PREFIXES = ("backbone.body.layer2", "backbone.body.layer3", "backbone.body.layer4", "backbone.fpn", "rpn", "roi_heads")


def _make_case(dev, N, H, W):
    from hallucidet_amd.models.detector import Detector
    torch.manual_seed(31)
    det = Detector(name="fasterrcnn", pretrained=False, n_classes=2, size=300).detector.to(dev)
    images = torch.rand(N, 3, H, W)
    targets = []
    for i in range(N):
        k = 1 + i
        xy = torch.rand(k, 2) * torch.tensor([W * 0.5, H * 0.5])
        wh = torch.rand(k, 2) * torch.tensor([W * 0.3, H * 0.4]) + 8.0
        targets.append({"boxes": torch.cat([xy, xy + wh], 1), "labels": torch.ones(k, dtype=torch.int64)})
    il, _ = det.transform(images.to(dev), None)
    det.backbone.calibrate_(il.tensors)
    oracle = od.FasterRCNN(num_classes=2, size=300)           # NOT folded: conv weights and FrozenBN stay separate tensors
    oracle.load_state_dict({k: v.cpu() for k, v in det.state_dict().items()})
    from _pins import product_weight_numerics_
    product_weight_numerics_(oracle)
    oracle.set_quant(ou.fp16_round)
    return det, oracle, images, targets


@pytest.fixture(scope="module")
def case(dev):
    return _make_case(dev, 2, 96, 128)


def test_trainable_set_matches_torchvision_rule(dev, case):
    det, _, _, _ = case
    det.set_trainable(True)
    names = {n for n, p in det.named_parameters() if p.requires_grad}
    all_names = {n for n, _ in det.named_parameters()}
    assert names == {n for n in all_names if n.startswith(TRAINABLE_PREFIXES)}
    assert "backbone.body.conv1.weight" not in names and not any(n.startswith("backbone.body.layer1") for n in names)
    assert len(det.trainable_parameters()) == len(names)
    det.set_trainable(False)
    assert not any(p.requires_grad for p in det.parameters())


def test_parameter_gradients_match_oracle_autograd(dev, case):
    """Linear probe loss on RPN outputs and box-head outputs over FIXED proposals (no sampler): every trainable tensor's
    gradient against torch autograd on the oracle (fp16 activation rounding on both sides)."""
    _check_parameter_gradients(dev, case)


def test_parameter_gradients_match_oracle_autograd_at_full_size(dev):
    """The same at the size of BASELINE configs[4] (train_detector.py, batch 16 per GPU, 512x640 images -> 16 x 300 x 300): every
    trainable tensor of the detector (heads, FPN, layer2-4) to rel-L2 <= 3 % / cosine >= 0.999 of the oracle's autograd."""
    _check_parameter_gradients(dev, _make_case(dev, 16, 512, 640))


def _check_parameter_gradients(dev, case):
    from hallucidet_amd.optim import ParamArena
    from _pins import record
    det, oracle, images, targets = case
    S = 256.0
    det.train()
    det.set_trainable(True, grad_scale=S)
    arena = ParamArena(det.trainable_parameters())
    det.invalidate_packs()
    g = torch.Generator().manual_seed(5)
    props = []
    for i in range(images.shape[0]):
        xy = torch.rand(40, 2, generator=g) * 200
        wh = torch.rand(40, 2, generator=g) * torch.tensor([90.0, 90.0]) + 8
        props.append(torch.cat([xy, xy + wh], 1))
    with record(det) as rec:
        il, _ = det.transform(images.to(dev), None)
        f = det.backbone(il.tensors)
        obj, reg = det.rpn.head(list(f.values()))
        bf = det.roi_heads.box_roi_pool(f, [p.to(dev) for p in props], il.image_sizes)
        logits, regs = det.roi_heads.box_predictor(det.roi_heads.box_head(bf))
    w_obj = [torch.randn(o.shape, generator=g) for o in obj]
    w_reg = [torch.randn(o.shape, generator=g) for o in reg]
    w_l, w_r = torch.randn(logits.shape, generator=g), torch.randn(regs.shape, generator=g)
    loss = sum((o * w.to(dev)).sum() for o, w in zip(obj, w_obj)) + sum((o * w.to(dev)).sum() for o, w in zip(reg, w_reg))
    loss = loss + (logits * w_l.to(dev)).sum() + (regs * w_r.to(dev)).sum()
    arena.flat_grads.zero_()
    (loss * S).backward()
    got = {n: p.grad.detach().cpu().clone() for n, p in det.named_parameters() if p.requires_grad}

    oracle.train()
    for n, p in oracle.named_parameters():
        p.requires_grad_(n.startswith(TRAINABLE_PREFIXES))
        p.grad = None
    # end to end with the product's ReLU / max-pool decisions (tests/_pins.py): same piecewise-linear network on both sides
    pins = rec.pins()
    oracle.set_pins(pins)
    try:
        ol, _ = oracle.transform(images, None)
        of = oracle.backbone(ol.tensors)
        oobj, oreg = oracle.rpn.head(list(of.values()))
        obf = oracle.roi_heads.box_roi_pool(of, props, ol.image_sizes)
        ologits, oregs = oracle.roi_heads.box_predictor(oracle.roi_heads.box_head(obf))
    finally:
        oracle.set_pins(None)
    assert pins.used == set(pins.masks)
    from _pins import assert_borrowed_decisions_are_noise
    assert_borrowed_decisions_are_noise(pins, "train_detector")
    oloss = sum((o * w).sum() for o, w in zip(oobj, w_obj)) + sum((o * w).sum() for o, w in zip(oreg, w_reg))
    oloss = oloss + (ologits * w_l).sum() + (oregs * w_r).sum()
    oloss.backward()
    want = {n: p.grad for n, p in oracle.named_parameters() if p.requires_grad}
    assert set(got) == set(want)

    def agree(n, ref):
        a, b = got[n].flatten().double(), ref.flatten().double()
        assert torch.isfinite(a).all(), n
        return float((a * b).sum() / (a.norm() * b.norm() + 1e-30)), float((a - b).norm() / (b.norm() + 1e-30))

    # (A) stage-wise on IDENTICAL inputs (the product's own feature maps / pooled RoI features fed to the oracle heads):
    # every ReLU decision is then taken on the same numbers, what remains is fp16 storage of activations and gradients
    for p in oracle.parameters():
        p.grad = None
    pf = [nchw(t).detach() for t in f.values()]
    sobj, sreg = oracle.rpn.head(pf)
    (sum((o * w).sum() for o, w in zip(sobj, w_obj)) + sum((o * w).sum() for o, w in zip(sreg, w_reg))).backward()
    slog, sregs = oracle.roi_heads.box_predictor(oracle.roi_heads.box_head(bf.detach().float().cpu().permute(0, 3, 1, 2)))
    ((slog * w_l).sum() + (sregs * w_r).sum()).backward()
    for n, p in oracle.named_parameters():
        if n.startswith(("rpn.head", "roi_heads.box_head", "roi_heads.box_predictor")):
            cos, rel = agree(n, p.grad)
            print("   stage  %-42s cos %.6f rel %.4f" % (n, cos, rel))
            assert cos > 0.9995 and rel < 0.03, (n, cos, rel)

    # (B) end to end, every trainable tensor (heads, FPN, layer2-4), decisions shared: rel-L2 <= 3 %, cosine >= 0.999 per tensor
    worst = {}
    for n in sorted(want):
        cos, rel = agree(n, want[n])
        grp = ("heads" if n.startswith(("roi_heads", "rpn")) else "fpn" if "fpn" in n else n.split(".")[2])
        if rel > worst.setdefault(grp, [1.0, 0.0, n])[1]:
            worst[grp] = [cos, rel, n]
        assert cos >= 0.999 and rel <= 0.03, (n, cos, rel)
    print({k: (round(v[0], 5), round(v[1], 4), v[2]) for k, v in worst.items()})
    assert set(worst) == {"heads", "fpn", "layer4", "layer3", "layer2"}
    # frozen tensors received nothing
    assert det.backbone.body.conv1.weight.grad is None and det.backbone.body.layer1[0].conv1.weight.grad is None
    det.set_trainable(False)
    det.eval()


def test_fit_step_learns_and_keeps_frozen_parts_fixed(dev):
    from hallucidet_amd import synthetic
    from hallucidet_amd.models.detector import Detector
    from hallucidet_amd.train_detector import DetectorLit
    torch.manual_seed(41)
    det = Detector(name="fasterrcnn", pretrained=False, n_classes=2, size=300).detector.to(dev)
    rgb, trgb, _, _ = synthetic.make_batch(2, 128, 160, seed=9, device=str(dev))
    il, _ = det.transform(rgb, None)
    det.backbone.calibrate_(il.tensors)
    lit = DetectorLit(batch_size=2, lr=1e-4, detector_name="fasterrcnn", pretrained=False, detector=det, device=str(dev)).prepare()
    with pytest.raises(RuntimeError, match="set_trainable"):
        det.set_trainable(False)
        Detector.calculate_loss(det, rgb, trgb, train_det=True, model_name="fasterrcnn")
    det.set_trainable(True, grad_scale=lit.loss_scale)
    before = {n: p.detach().clone() for n, p in det.named_parameters()}
    v0 = float(lit.validation_step((rgb, trgb), 0))
    losses = [float(lit.fit_step((rgb, trgb))) for _ in range(12)]
    assert all(map(lambda v: v == v and abs(v) < 1e4, losses)), losses
    assert float(lit.optimizer.found_inf) == 0.0
    moved = [n for n, p in det.named_parameters() if not torch.equal(p.detach(), before[n])]
    frozen = [n for n, p in det.named_parameters() if not p.requires_grad]
    assert frozen and not set(moved) & set(frozen)
    assert len(moved) == len(det.trainable_parameters()), (len(moved), len(det.trainable_parameters()))
    # Adam: |update| <= lr * (1-b1)/sqrt(1-b2) ~ 3.2 lr in the worst case (a gradient spike), ~lr typically
    assert max(float((p.detach() - before[n]).abs().max()) for n, p in det.named_parameters()) <= 12 * 1e-4 * 3.2
    v1 = float(lit.validation_step((rgb, trgb), 0))
    print("train losses", [round(v, 4) for v in losses], "val (unweighted sum)", round(v0, 4), "->", round(v1, 4))
    assert v1 < v0, (v0, v1)             # 12 Adam steps on one batch reduce its own loss
    dets = lit.test_step((rgb, trgb), 0)
    assert len(dets) == 2 and set(dets[0]) == {"boxes", "labels", "scores"}
    m = lit.on_test_epoch_end()
    assert set(m) == {"map", "map_50", "map_75", "map_per_class"} and -1.0 <= float(m["map_50"]) <= 1.0
    mv = lit.on_validation_epoch_end()                       # two validation_step calls above fed it
    assert -1.0 <= float(mv["map"]) <= 1.0


# ---------------------------------------------------------------------------------------------------------------------
# RetinaNet fine-tuning (train_detector.py with detector_name='retinanet': towers shared over 5 levels, P6/P7 convs)
# ---------------------------------------------------------------------------------------------------------------------
def test_retinanet_parameter_gradients_and_fit_step(dev):
    from hallucidet_amd import synthetic
    from hallucidet_amd.models.detector import Detector
    from hallucidet_amd.optim import ParamArena
    from hallucidet_amd.train_detector import DetectorLit
    from oracle import retinanet as orn
    torch.manual_seed(51)
    det = Detector(name="retinanet", pretrained=False, n_classes=2, size=300).detector.to(dev)
    with torch.no_grad():
        for t in (det.head.classification_head, det.head.regression_head):      # N(0,0.01) towers shrink the signal 5x per layer
            for l in t.conv:
                if isinstance(l, torch.nn.Conv2d):
                    l.weight.normal_(0, 0.03)
        det.head.classification_head.cls_logits.bias.fill_(-2.0)
    rgb, trgb, _, _ = synthetic.make_batch(2, 128, 160, seed=9, device=str(dev))
    il, _ = det.transform(rgb, None)
    det.backbone.calibrate_(il.tensors)
    oracle = orn.RetinaNet(num_classes=2, size=300)
    oracle.load_state_dict({k: v.cpu() for k, v in det.state_dict().items()})
    from _pins import product_weight_numerics_
    product_weight_numerics_(oracle)
    oracle.set_quant(ou.fp16_round)

    S = 256.0
    det.train()
    det.set_trainable(True, grad_scale=S)
    arena = ParamArena(det.trainable_parameters())
    det.invalidate_packs()
    from _pins import record
    with record(det) as rec:
        feats = list(det.backbone(il.tensors).values())
        ho = det.head(feats)
    g = torch.Generator().manual_seed(6)
    w_c, w_r = torch.randn(ho["cls_logits"].shape, generator=g), torch.randn(ho["bbox_regression"].shape, generator=g)
    arena.flat_grads.zero_()
    (((ho["cls_logits"] * w_c.to(dev)).sum() + (ho["bbox_regression"] * w_r.to(dev)).sum()) * S).backward()
    got = {n: p.grad.detach().cpu().clone() for n, p in det.named_parameters() if p.requires_grad}

    prefixes = ("backbone.body.layer2", "backbone.body.layer3", "backbone.body.layer4", "backbone.fpn", "head")
    assert set(got) == {n for n, _ in det.named_parameters() if n.startswith(prefixes)}
    oracle.train()
    for n, p in oracle.named_parameters():
        p.requires_grad_(n.startswith(prefixes))
    # stage-wise: oracle head on the product's own feature maps -> every ReLU decision on identical numbers
    oho = oracle.head([nchw(t).detach() for t in feats])
    ((oho["cls_logits"] * w_c).sum() + (oho["bbox_regression"] * w_r).sum()).backward()
    for n, p in oracle.named_parameters():
        if n.startswith("head."):
            a, b = got[n].flatten().double(), p.grad.flatten().double()
            cos, rel = float((a * b).sum() / (a.norm() * b.norm() + 1e-30)), float((a - b).norm() / (b.norm() + 1e-30))
            assert cos > 0.999 and rel < 0.05, (n, cos, rel)
    # end to end: trunk / FPN / P6 / P7 with the product's ReLU / max-pool decisions (tests/_pins.py): every tensor tight
    for p in oracle.parameters():
        p.grad = None
    pins = rec.pins()
    oracle.set_pins(pins)
    try:
        ol, _ = oracle.transform(rgb.cpu(), None)
        oho = oracle.head(list(oracle.backbone(ol.tensors).values()))
    finally:
        oracle.set_pins(None)
    assert pins.used == set(pins.masks)
    from _pins import assert_borrowed_decisions_are_noise
    assert_borrowed_decisions_are_noise(pins, "train_detector")
    ((oho["cls_logits"] * w_c).sum() + (oho["bbox_regression"] * w_r).sum()).backward()
    worst = {}
    for n, p in oracle.named_parameters():
        if p.grad is None:
            continue
        a, b = got[n].flatten().double(), p.grad.flatten().double()
        cos, rel = float((a * b).sum() / (a.norm() * b.norm() + 1e-30)), float((a - b).norm() / (b.norm() + 1e-30))
        grp = "head" if n.startswith("head") else ("extra" if "extra_blocks" in n else "fpn" if "fpn" in n else n.split(".")[2])
        if rel > worst.get(grp, (1.0, 0.0))[1]:
            worst[grp] = (cos, rel, n)
        assert cos >= 0.999 and rel <= 0.03, (n, cos, rel)
    print({k: (round(v[0], 5), round(v[1], 4), v[2]) for k, v in worst.items()})
    assert set(worst) == {"head", "fpn", "extra", "layer4", "layer3", "layer2"}
    det.set_trainable(False)

    # ---- DetectorLit on RetinaNet: learns on its own batch, frozen parts stay fixed
    lit = DetectorLit(batch_size=2, lr=1e-4, detector_name="retinanet", pretrained=False, detector=det, device=str(dev)).prepare()
    before = {n: p.detach().clone() for n, p in det.named_parameters()}
    v0 = float(lit.validation_step((rgb, trgb), 0))
    losses = [float(lit.fit_step((rgb, trgb))) for _ in range(12)]
    v1 = float(lit.validation_step((rgb, trgb), 0))
    print("retinanet train losses", [round(v, 4) for v in losses], "val", round(v0, 4), "->", round(v1, 4))
    assert all(v == v for v in losses) and float(lit.optimizer.found_inf) == 0.0 and v1 < v0
    moved = {n for n, p in det.named_parameters() if not torch.equal(p.detach(), before[n])}
    assert moved == {n for n, p in det.named_parameters() if p.requires_grad}
    assert set(lit._last_losses) >= {"classification", "bbox_regression"} and lit._last_losses["loss_objectness"] == 0.0
