"""GPU parity of the frozen FCOS path (reference src/utils/eval_forward_fcos.py:11-83) against the CPU oracle
(oracle/fcos.py).  Same methodology as test_retinanet_gpu.py: each stage of the product is compared with the oracle ON THE SAME
INPUTS, so integer outputs (matched indices, labels, kept candidates) are IDENTICAL and fp32 maths agrees to 1e-5; the conv /
GroupNorm towers (fp16 storage) are compared with the oracle on the product's rounding schedule."""
import pytest
import torch

from oracle import detection as od
from oracle import fcos as ofc
from oracle import unet as ou
from test_detector_gpu import fold_oracle_, nchw, _t2d

pytestmark = pytest.mark.gpu
A_TOTAL = 38 * 38 + 19 * 19 + 100 + 25 + 9


@pytest.fixture(scope="module")
def case(dev):
    from hallucidet_amd.models.detector import Detector
    torch.manual_seed(23)
    det = Detector(name="fcos", pretrained=False, n_classes=2, size=300).detector
    with torch.no_grad():
        # the prior bias -log(99) would leave no candidate above score_thresh 0.2 on random features: spread the outputs
        det.head.classification_head.cls_logits.weight.normal_(0, 0.05)
        det.head.classification_head.cls_logits.bias.fill_(0.5)
        det.head.regression_head.bbox_reg.weight.normal_(0, 0.03)
        det.head.regression_head.bbox_reg.bias.fill_(0.8)
        det.head.regression_head.bbox_ctrness.weight.normal_(0, 0.05)
        for t in (det.head.classification_head, det.head.regression_head):
            for l in t.conv:
                if isinstance(l, torch.nn.Conv2d):
                    l.weight.normal_(0, 0.03)
                if isinstance(l, torch.nn.GroupNorm):          # non-trivial affine
                    l.weight.uniform_(0.5, 1.5)
                    l.bias.normal_(0, 0.2)
        for mod in det.modules():
            if isinstance(mod, torch.nn.Conv2d) and mod.bias is not None:
                mod.weight.copy_(mod.weight.half().float())
    det.head.invalidate()
    det = det.to(dev).eval()
    N, H, W = 3, 96, 128
    images = torch.rand(N, 3, H, W)
    targets = []
    for i in range(N):
        if i == 1:
            boxes = torch.zeros(0, 4)                          # FCOS.compute_loss's all -1 branch
        elif i == 2:
            boxes = torch.tensor([[10.0, 8.0, 118.0, 90.0], [40.0, 30.0, 90.0, 70.0]])      # nested: the smaller box wins inside it
        else:
            xy = torch.rand(3, 2) * torch.tensor([W * 0.5, H * 0.5])
            wh = torch.rand(3, 2) * torch.tensor([W * 0.4, H * 0.4]) + 10.0
            boxes = torch.cat([xy, xy + wh], 1)
        targets.append({"boxes": boxes.reshape(-1, 4), "labels": torch.ones(boxes.shape[0], dtype=torch.int64)})
    il, _ = det.transform(images.to(dev), None)
    det.backbone.calibrate_(il.tensors)
    oracle = ofc.FCOS(num_classes=2, size=300)
    oracle.load_state_dict({k: v.cpu() for k, v in det.state_dict().items()})
    fold_oracle_(oracle)
    oracle.eval()
    oracle.set_quant(ou.fp16_round)
    return det, oracle, images, targets


def test_factory_surface_and_state_dict_keys(case):
    det, oracle, _, _ = case
    assert list(det.state_dict().keys()) == list(oracle.state_dict().keys())
    assert det.head.classification_head.cls_logits.weight.shape == (2, 256, 3, 3)
    assert abs(float(det.head.classification_head.cls_logits.bias[0]) - 0.5) < 1e-6          # overwritten by the fixture ...
    from hallucidet_amd.models.detector import Detector
    fresh = Detector(name="fcos_resnet50_fpn", pretrained=False, n_classes=2, size=300).detector
    import math
    assert torch.allclose(fresh.head.classification_head.cls_logits.bias, torch.full((2,), -math.log(99.0)))   # ... from the reference's re-heading
    keys = set(fresh.state_dict().keys())
    for k in ("head.classification_head.conv.1.weight", "head.classification_head.conv.10.bias", "head.regression_head.conv.9.weight",
              "head.regression_head.bbox_ctrness.bias", "head.regression_head.bbox_reg.weight", "backbone.fpn.extra_blocks.p7.weight"):
        assert k in keys, k
    assert (fresh.score_thresh, fresh.nms_thresh, fresh.detections_per_img, fresh.topk_candidates, fresh.center_sampling_radius) == (0.2, 0.6, 100, 1000, 1.5)
    fresh.set_trainable(True)
    names = {n for n, p_ in fresh.named_parameters() if p_.requires_grad}
    assert "head.regression_head.conv.1.weight" in names and "backbone.fpn.extra_blocks.p6.weight" in names and "backbone.body.layer2.0.conv1.weight" in names
    assert not any(n.startswith(("backbone.body.conv1", "backbone.body.layer1", "backbone.body.bn1")) for n in names)
    fresh.set_trainable(False)
    assert not any(p_.requires_grad for p_ in fresh.parameters())


def test_groupnorm_kernel_forward_backward(dev):
    """hd_groupnorm8_relu / _bwd against torch.nn.functional.group_norm + relu in fp32 (fp16 storage of input, output and
    gradients: 2e-3 relative), at every pyramid level size incl. a single pixel."""
    from hallucidet_amd import ops
    g = torch.Generator().manual_seed(2)
    for (N, H, W) in ((3, 38, 38), (2, 19, 19), (2, 10, 10), (4, 5, 5), (2, 3, 3), (1, 1, 1)):
        C = 256
        x = (torch.randn(N, H, W, C, generator=g) * 1.5 + 0.3).half()
        ga, be = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
        dy = (torch.randn(N, H, W, C, generator=g) * 0.1).half()
        y, stat = ops.groupnorm8_relu(x.to(dev), ga.to(dev), be.to(dev), 1e-5)
        xr = x.float().permute(0, 3, 1, 2).clone().requires_grad_(True)
        yr = torch.relu(torch.nn.functional.group_norm(xr, 32, ga, be, 1e-5))
        yr.backward(dy.float().permute(0, 3, 1, 2))
        got = y.float().cpu().permute(0, 3, 1, 2)
        assert (got - yr.detach()).abs().max() < 2e-3 * max(1.0, float(yr.abs().max())), (N, H, W)
        mean = x.float().reshape(N, H * W, 32, 8).permute(0, 2, 1, 3).reshape(N, 32, -1).mean(-1)
        assert torch.allclose(stat[:, :, 0].cpu(), mean, atol=1e-4)
        dx = ops.groupnorm8_relu_bwd(dy.to(dev), x.to(dev), y, ga.to(dev), stat)
        want = xr.grad
        # the product's ReLU mask comes from ITS rounded output: compare where both sides agree on the mask (all but ulp cases)
        e = (dx.float().cpu().permute(0, 3, 1, 2) - want).abs()
        assert float(e.mean()) < 2e-3 * float(want.abs().mean()) + 1e-6 and float(e.max()) < 3e-2 * float(want.abs().max()) + 1e-4, (N, H, W, float(e.max()))


def _head_io(det, il):
    with torch.no_grad():
        feats = list(det.backbone(il.tensors).values())
        ho = det.head(feats)
    return feats, ho


def test_head_outputs_and_anchors(dev, case):
    det, oracle, images, targets = case
    il, _ = det.transform(images.to(dev), None)
    feats, ho = _head_io(det, il)
    assert [tuple(f.shape[1:3]) for f in feats] == [(38, 38), (19, 19), (10, 10), (5, 5), (3, 3)]
    ofeats = [nchw(t) for t in feats]
    with torch.no_grad():
        oho = oracle.head(ofeats)
    assert ho["cls_logits"].shape == (3, A_TOTAL, 2) and ho["bbox_regression"].shape == (3, A_TOTAL, 4) and ho["bbox_ctrness"].shape == (3, A_TOTAL, 1)
    assert float(ho["bbox_regression"].min()) >= 0.0                     # the regression branch ends in a ReLU
    for k in ho:
        a, b = ho[k].cpu(), oho[k]
        # 4 x (conv, GroupNorm, ReLU) + conv in fp16 storage, fp32 accumulate on both sides
        assert (a - b).abs().max() < 1e-2 * max(1.0, float(b.abs().max())), (k, float((a - b).abs().max()), float(b.abs().max()))
        assert (a - b).abs().mean() < 1e-3 * max(1.0, float(b.abs().mean())), k
    anchors = det.anchor_generator(il, feats)
    oanchors = oracle.anchor_generator(od.ImageList(torch.zeros(3, 3, 300, 300), [(300, 300)] * 3), ofeats)
    assert torch.equal(anchors[0].cpu(), oanchors[0]) and anchors[0].shape == (A_TOTAL, 4)
    # one stride-sized square per location: strides 300 // (38, 19, 10, 5, 3), sizes 8 .. 128
    a0 = anchors[0].cpu()
    assert torch.equal(a0[0], torch.tensor([-4.0, -4.0, 4.0, 4.0])) and torch.equal(a0[1], torch.tensor([3.0, -4.0, 11.0, 4.0]))
    assert torch.equal(a0[-1], torch.tensor([200.0 - 64, 200.0 - 64, 200.0 + 64, 200.0 + 64]))


def test_matching_and_losses_exact_on_same_head_outputs(dev, case):
    """hd_fcos_match / hd_fcos_loss on IDENTICAL fp32 head outputs: matched indices are identical to the oracle's, the three
    losses agree to fp32 summation-order accuracy, through FCOS.compute_loss and through FCOSHead.compute_loss."""
    det, oracle, images, targets = case
    il, tg = det.transform(images.to(dev), _t2d(targets, dev))
    _, otg = oracle.transform(images, targets)
    feats, ho = _head_io(det, il)
    anchors = det.anchor_generator(il, feats)
    napl = [f.shape[1] * f.shape[2] for f in feats]
    oho = {k: v.cpu() for k, v in ho.items()}
    oanchors = [a.cpu() for a in anchors]
    want = oracle.compute_loss(otg, oho, oanchors, napl)
    from hallucidet_amd.models import detection as D
    gt, glab, gvalid = D.pad_targets(tg, dev)
    m = det.match_batched(anchors[0], gt, gvalid, napl)
    n_fg = 0
    for i, t in enumerate(otg):
        om = oracle.match(oanchors[i], t, napl)
        assert torch.equal(m[i].cpu(), om), (i, int((m[i].cpu() != om).sum()))
        n_fg += int((om >= 0).sum())
    assert n_fg > 20 and int((m[1] >= 0).sum()) == 0
    inner = m[2].cpu() == 1                                                 # locations given to the nested (smaller) box
    assert int(inner.sum()) > 0
    got = det.compute_loss(tg, ho, anchors, napl)
    got_head = det.head.compute_loss(tg, ho, anchors, [m[i] for i in range(3)])
    assert set(got) == {"classification", "bbox_regression", "bbox_ctrness"}
    for k in want:
        assert torch.allclose(got[k].cpu(), want[k], rtol=1e-5, atol=1e-6), (k, float(got[k]), float(want[k]))
        assert torch.allclose(got_head[k].cpu(), want[k], rtol=1e-5, atol=1e-6), k


def test_loss_gradients_against_oracle_autograd(dev):
    """hd_fcos_loss_bwd (focal, generalized-IoU incl. disjoint and containing boxes, centre-ness BCE) against autograd through the
    oracle's loss on a synthetic case with every geometric relation between prediction and target."""
    from hallucidet_amd.models import fcos as F_
    g = torch.Generator().manual_seed(9)
    B, K = 2, 3
    cx, cy = torch.meshgrid(torch.arange(4.0, 100.0, 8.0), torch.arange(4.0, 100.0, 8.0), indexing="ij")
    anchors = torch.stack([cx.flatten() - 4, cy.flatten() - 4, cx.flatten() + 4, cy.flatten() + 4], 1)
    A = anchors.shape[0]
    gt = torch.tensor([[[10.0, 12.0, 70.0, 80.0], [30.0, 30.0, 50.0, 44.0]], [[5.0, 5.0, 95.0, 60.0], [0.0, 0.0, 0.0, 0.0]]])
    glab = torch.tensor([[1, 2], [0, 0]])
    matched = torch.full((B, A), -1, dtype=torch.int64)
    ac = (anchors[:, :2] + anchors[:, 2:]) / 2
    for b in range(B):
        for j in range(2):
            if gt[b, j, 2] > gt[b, j, 0]:
                ins = (ac[:, 0] > gt[b, j, 0]) & (ac[:, 0] < gt[b, j, 2]) & (ac[:, 1] > gt[b, j, 1]) & (ac[:, 1] < gt[b, j, 3])
                matched[b, ins] = j
    assert int((matched >= 0).sum()) > 40
    cls = torch.randn(B, A, K, generator=g) * 2
    reg = torch.rand(B, A, 4, generator=g) * 6          # ltrb in anchor sizes: from tiny boxes (disjoint from the target) to huge ones (containing it)
    reg[0, :5] = 0.0                                      # degenerate zero-area predictions
    ctr = torch.randn(B, A, 1, generator=g)
    w = torch.tensor([0.7, 1.3, 2.1])

    def oracle_losses(c, r, t):
        head = ofc.FCOSHead(256, 1, K)
        tg = [{"boxes": gt[b][: (2 if b == 0 else 1)], "labels": glab[b][: (2 if b == 0 else 1)]} for b in range(B)]
        return head.compute_loss(tg, {"cls_logits": c, "bbox_regression": r, "bbox_ctrness": t}, [anchors] * B, [matched[b] for b in range(B)])
    co, ro, to = (t.clone().requires_grad_(True) for t in (cls, reg, ctr))
    lo = oracle_losses(co, ro, to)
    (w[0] * lo["classification"] + w[1] * lo["bbox_regression"] + w[2] * lo["bbox_ctrness"]).backward()
    cg, rg, tg_ = (t.to(dev).requires_grad_(True) for t in (cls, reg, ctr))
    lg = F_.fcos_loss_batched(anchors.to(dev), gt.to(dev), glab.to(dev), {"cls_logits": cg, "bbox_regression": rg, "bbox_ctrness": tg_}, matched.to(dev))
    (w[0] * lg["classification"] + w[1] * lg["bbox_regression"] + w[2] * lg["bbox_ctrness"]).backward()
    for k in lo:
        assert torch.allclose(lg[k].detach().cpu(), lo[k].detach(), rtol=1e-5, atol=1e-6), (k, float(lg[k]), float(lo[k]))
    assert torch.allclose(cg.grad.cpu(), co.grad, rtol=2e-4, atol=1e-7)
    assert torch.allclose(tg_.grad.cpu(), to.grad, rtol=1e-4, atol=1e-7)
    assert torch.allclose(rg.grad.cpu(), ro.grad, rtol=1e-3, atol=1e-6), float((rg.grad.cpu() - ro.grad).abs().max())
    assert float(ro.grad.abs().max()) > 0


def test_postprocess_exact_on_same_head_outputs(dev, case):
    det, oracle, images, targets = case
    il, _ = det.transform(images.to(dev), None)
    feats, ho = _head_io(det, il)
    anchors = det.anchor_generator(il, feats)
    napl = [f.shape[1] * f.shape[2] for f in feats]
    split = {k: list(v.split(napl, dim=1)) for k, v in ho.items()}
    got = det.postprocess_detections(split, [list(a.split(napl)) for a in anchors], il.image_sizes)
    osplit = {k: [t.cpu() for t in v] for k, v in split.items()}
    want = oracle.postprocess_detections(osplit, [list(a.cpu().split(napl)) for a in anchors], [(300, 300)] * 3)
    sb, ss, sl, counts = det.postprocess_detections_padded(ho, anchors[0], napl, (300, 300))
    n_nontrivial = 0
    for i, (g, w) in enumerate(zip(got, want)):
        assert torch.equal(g["labels"].cpu(), w["labels"]), i
        assert torch.allclose(g["scores"].cpu(), w["scores"], rtol=1e-6, atol=1e-7) and torch.allclose(g["boxes"].cpu(), w["boxes"], atol=1e-4)
        c = int(counts[i])
        assert c == w["labels"].numel() <= 100 and torch.equal(sl[i, :c].cpu(), w["labels"])
        assert torch.allclose(ss[i, :c].cpu(), w["scores"], rtol=1e-6, atol=1e-7) and torch.allclose(sb[i, :c].cpu(), w["boxes"], atol=1e-4)
        n_nontrivial += int(0 < c)
    assert n_nontrivial == 3


def test_end_to_end_losses_detections_and_image_gradient(dev, case):
    """Detector.calculate_loss(model_name='fcos') vs the oracle end to end, and the data gradient w.r.t. the input images against
    the oracle's autograd with shared rounding (fp16 storage => statistical tolerance, as for the other two detectors)."""
    from hallucidet_amd.models.detector import Detector
    from _pins import record, grad_agreement
    det, oracle, images, targets = case
    x = images.to(dev).requires_grad_(True)
    with record(det) as rec:
        losses, dets = Detector.calculate_loss(det, x, _t2d(targets, dev), train_det=False, model_name="fcos")
    assert set(losses) == {"classification", "bbox_regression", "bbox_ctrness"}
    (losses["classification"] + losses["bbox_regression"] + losses["bbox_ctrness"]).backward()
    # decisions (ReLU after GroupNorm / in the trunk, ReLU on the regression outputs, max-pool winners) shared with the oracle
    pins = rec.pins()
    assert len(pins.masks) == 1 + 3 * 16 + 1 + 2 * 5 * 4 + 5
    oracle.set_pins(pins)
    try:
        ox = images.clone().requires_grad_(True)
        olosses, odets = ofc.eval_forward_fcos(oracle, ox, targets, train_det=False)
    finally:
        oracle.set_pins(None)
    assert pins.used == set(pins.masks)
    from _pins import assert_borrowed_decisions_are_noise
    assert_borrowed_decisions_are_noise(pins, "fcos")
    (olosses["classification"] + olosses["bbox_regression"] + olosses["bbox_ctrness"]).backward()
    for k in losses:
        print("fcos %s: product %.6f oracle %.6f" % (k, float(losses[k]), float(olosses[k])))
        assert abs(float(losses[k]) - float(olosses[k])) < 5e-3 * abs(float(olosses[k])) + 1e-5, (k, float(losses[k]), float(olosses[k]))
    assert len(dets) == 3 and all(d["boxes"].shape[0] <= 100 for d in dets)
    for d in dets:
        assert d["boxes"].shape[1] == 4 and d["labels"].dtype == torch.int64 and float(d["boxes"].max()) <= 128.0 + 1e-3
    g, og = x.grad.cpu(), ox.grad
    assert torch.isfinite(g).all() and float(og.abs().max()) > 0
    cos, rel = grad_agreement(g, og)
    print("fcos image gradient: rel-L2 %.4f cosine %.5f" % (rel, cos))
    assert cos >= 0.999 and rel <= 0.03, (rel, cos)
    det.batched_heads = False                      # list-based (reference-shaped) route gives the same numbers
    try:
        with torch.no_grad():
            l2, d2 = Detector.calculate_loss(det, images.to(dev), _t2d(targets, dev), train_det=False, model_name="fcos")
    finally:
        det.batched_heads = True
    for k in losses:
        assert torch.allclose(l2[k], losses[k].detach(), rtol=1e-5, atol=1e-6), k
    for a, b in zip(d2, dets):
        assert torch.equal(a["labels"], b["labels"]) and torch.allclose(a["boxes"], b["boxes"], atol=1e-4)
    with pytest.raises(RuntimeError, match="set_trainable"):
        Detector.calculate_loss(det, images.to(dev), _t2d(targets, dev), train_det=True, model_name="fcos")


def test_three_pass_fusion_equals_three_single_passes(dev, case, pinned_tiles):
    from hallucidet_amd.utils.eval_forward_fcos import eval_forward_fcos, eval_forward_fcos_multi
    det, _, images, targets = case
    tg = _t2d(targets, dev)
    a = images.to(dev).requires_grad_(True)
    b, c = torch.rand_like(images).to(dev), torch.rand_like(images).to(dev)
    out = eval_forward_fcos_multi(det, [a, b, c], [tg, tg, tg])
    sum(out[0][0].values()).backward()
    g_multi = a.grad.clone()
    a2 = images.to(dev).requires_grad_(True)
    l1, d1 = eval_forward_fcos(det, a2, tg)
    sum(l1.values()).backward()
    for k in l1:
        assert torch.allclose(l1[k], out[0][0][k], rtol=1e-5, atol=1e-6), k
    assert torch.allclose(g_multi, a2.grad, rtol=1e-3, atol=1e-6 + 1e-3 * float(a2.grad.abs().max()))
    with torch.no_grad():
        singles = [d1, eval_forward_fcos(det, b, tg)[1], eval_forward_fcos(det, c, tg)[1]]
    for (_, dm), ds in zip(out, singles):
        for x, y in zip(dm, ds):
            assert torch.equal(x["labels"], y["labels"]) and torch.equal(x["scores"], y["scores"]) and torch.equal(x["boxes"], y["boxes"])


def test_training_step_with_fcos_detector(dev):
    """One train_hallucidet step with detector_name='fcos': the 11-key loss dict carries det_bbox_ctrness (weight 0.1,
    train_hallucidet.py:201-205), the loss is finite, the U-Net parameters move, and a rerun from the same seed is bit-identical."""
    from hallucidet_amd import synthetic
    outs = []
    for _ in range(2):
        lit = synthetic.make_module(seed=41, device=str(dev), precision=16, detector_name="fcos")
        with torch.no_grad():
            lit.detector.head.classification_head.cls_logits.bias.fill_(-1.0)
        lit.detector.invalidate_packs()
        batch = synthetic.make_batch(2, 128, 160, seed=6, device=str(dev))
        lit.encoder_decoder.train()
        out = lit.forward_step(*batch, 0, step="train")
        assert float(out["loss"]["det_bbox_ctrness"]) > 0 and out["loss"]["det_objectness"] == 0.0 and out["loss"]["det_rpn_box_reg"] == 0.0
        tot = out["loss"]["det_regression"] + out["loss"]["det_classification"] + out["loss"]["det_bbox_ctrness"]
        assert abs(float(out["loss"]["det_total"]) - float(tot)) < 1e-6
        p0 = lit.encoder_decoder.runner.flat_params.clone()
        loss = lit.fit_step(batch)
        assert torch.isfinite(loss) and not torch.equal(p0, lit.encoder_decoder.runner.flat_params)
        outs.append((float(loss), lit.encoder_decoder.runner.flat_grads.clone()))
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])


def test_fcos_parameter_gradients_and_fit_step(dev):
    """Detector fine-tuning with FCOS (train_detector.py, detector_name='fcos'): parameter gradients of the head (convs, GroupNorm
    affines, the three output convs: stage-wise on the product's own feature maps, so every ReLU decision is taken on identical
    numbers) and of the trunk (end to end, statistical) against the oracle's autograd; then DetectorLit learns on its own batch and
    leaves conv1 / layer1 / FrozenBN untouched."""
    from hallucidet_amd import synthetic
    from hallucidet_amd.models.detector import Detector
    from hallucidet_amd.optim import ParamArena
    from hallucidet_amd.train_detector import DetectorLit
    torch.manual_seed(53)
    det = Detector(name="fcos", pretrained=False, n_classes=2, size=300).detector.to(dev)
    with torch.no_grad():
        for t in (det.head.classification_head, det.head.regression_head):
            for l in t.conv:
                if isinstance(l, torch.nn.Conv2d):
                    l.weight.normal_(0, 0.03)
                if isinstance(l, torch.nn.GroupNorm):
                    l.weight.uniform_(0.5, 1.5)
                    l.bias.normal_(0, 0.2)
        det.head.classification_head.cls_logits.bias.fill_(-2.0)
        det.head.regression_head.bbox_reg.bias.fill_(0.5)
    rgb, trgb, _, _ = synthetic.make_batch(2, 128, 160, seed=9, device=str(dev))
    il, _ = det.transform(rgb, None)
    det.backbone.calibrate_(il.tensors)
    oracle = ofc.FCOS(num_classes=2, size=300)
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
    ws = {k: torch.randn(v.shape, generator=g) for k, v in ho.items()}
    arena.flat_grads.zero_()
    (sum((ho[k] * ws[k].to(dev)).sum() for k in ho) * S).backward()
    got = {n: p.grad.detach().cpu().clone() for n, p in det.named_parameters() if p.requires_grad}
    prefixes = ("backbone.body.layer2", "backbone.body.layer3", "backbone.body.layer4", "backbone.fpn", "head")
    assert set(got) == {n for n, _ in det.named_parameters() if n.startswith(prefixes)}
    oracle.train()
    for n, p in oracle.named_parameters():
        p.requires_grad_(n.startswith(prefixes))
    oho = oracle.head([nchw(t).detach() for t in feats])
    sum((oho[k] * ws[k]).sum() for k in oho).backward()
    for n, p in oracle.named_parameters():
        if n.startswith("head."):
            a, b = got[n].flatten().double(), p.grad.flatten().double()
            cos, rel = float((a * b).sum() / (a.norm() * b.norm() + 1e-30)), float((a - b).norm() / (b.norm() + 1e-30))
            assert cos > 0.999 and rel < 0.05, (n, cos, rel)
    # end to end with the product's ReLU / max-pool decisions (tests/_pins.py): every trainable tensor tight
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
    assert_borrowed_decisions_are_noise(pins, "fcos")
    sum((oho[k] * ws[k]).sum() for k in oho).backward()
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

    lit = DetectorLit(batch_size=2, lr=1e-4, detector_name="fcos", pretrained=False, detector=det, device=str(dev)).prepare()
    before = {n: p.detach().clone() for n, p in det.named_parameters()}
    v0 = float(lit.validation_step((rgb, trgb), 0))
    losses = [float(lit.fit_step((rgb, trgb))) for _ in range(12)]
    v1 = float(lit.validation_step((rgb, trgb), 0))
    print("fcos train losses", [round(v, 4) for v in losses], "val", round(v0, 4), "->", round(v1, 4))
    assert all(v == v for v in losses) and float(lit.optimizer.found_inf) == 0.0 and v1 < v0
    moved = {n for n, p in det.named_parameters() if not torch.equal(p.detach(), before[n])}
    assert moved == {n for n, p in det.named_parameters() if p.requires_grad}
    assert float(lit._last_losses["bbox_ctrness"]) > 0 and lit._last_losses["loss_objectness"] == 0.0
