"""GPU parity of the frozen RetinaNet path (BASELINE config 4) against the CPU oracle (oracle/retinanet.py).

Same methodology as test_detector_gpu.py: each stage of the product is compared with the oracle ON THE SAME INPUTS, so
integer outputs (matcher codes, labels, kept candidates) are IDENTICAL and fp32 maths agrees to 1e-5; conv stacks (fp16
storage) are compared with the oracle on the product's rounding schedule."""
import pytest
import torch

from oracle import detection as od
from oracle import retinanet as orn
from oracle import unet as ou
from test_detector_gpu import fold_oracle_, nchw, _t2d

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def case(dev):
    from hallucidet_amd.models.detector import Detector
    torch.manual_seed(21)
    det = Detector(name="retinanet", pretrained=False, n_classes=2, size=300).detector
    with torch.no_grad():
        # spread the logits (prior bias -log(99) would leave no candidate above score_thresh on random features)
        det.head.classification_head.cls_logits.weight.normal_(0, 0.05)
        det.head.classification_head.cls_logits.bias.fill_(-2.0)
        det.head.regression_head.bbox_reg.weight.normal_(0, 0.02)
        for t in (det.head.classification_head, det.head.regression_head):   # N(0,0.01) towers shrink the signal 5x per layer
            for l in t.conv:
                if isinstance(l, torch.nn.Conv2d):
                    l.weight.normal_(0, 0.03)
        for mod in det.modules():
            if isinstance(mod, torch.nn.Conv2d) and mod.bias is not None:
                mod.weight.copy_(mod.weight.half().float())
    det.head.invalidate()
    det = det.to(dev).eval()
    N, H, W = 3, 96, 128
    images = torch.rand(N, 3, H, W)
    targets = []
    for i in range(N):
        k = (2, 0, 1)[i]                                  # image 1 has no boxes (reference's empty-target branch)
        xy = torch.rand(k, 2) * torch.tensor([W * 0.5, H * 0.5])
        wh = torch.rand(k, 2) * torch.tensor([W * 0.3, H * 0.4]) + 8.0
        targets.append({"boxes": torch.cat([xy, xy + wh], 1).reshape(-1, 4), "labels": torch.ones(k, dtype=torch.int64)})
    il, _ = det.transform(images.to(dev), None)
    det.backbone.calibrate_(il.tensors)
    oracle = orn.RetinaNet(num_classes=2, size=300)
    missing = oracle.load_state_dict({k: v.cpu() for k, v in det.state_dict().items()})
    fold_oracle_(oracle)
    oracle.eval()
    oracle.set_quant(ou.fp16_round)
    return det, oracle, images, targets


def test_state_dict_keys_match_torchvision_tree(case):
    det, oracle, _, _ = case
    assert list(det.state_dict().keys()) == list(oracle.state_dict().keys())
    assert det.head.classification_head.cls_logits.weight.shape == (18, 256, 3, 3)


def test_trunk_p6_p7_features(dev, case):
    det, oracle, images, targets = case
    il, _ = det.transform(images.to(dev), None)
    ol, _ = oracle.transform(images, None)
    with torch.no_grad():
        f = det.backbone(il.tensors)
        of = oracle.backbone(ol.tensors)
    assert list(f.keys()) == ["0", "1", "2", "p6", "p7"]
    assert [tuple(v.shape[1:3]) for v in f.values()] == [(38, 38), (19, 19), (10, 10), (5, 5), (3, 3)]
    for k in f:
        a, b = nchw(f[k]), of[k]
        assert a.shape == b.shape, k
        e = (a - b).abs()
        assert e.mean() < 1.5e-2 * b.abs().mean() + 1e-4 and e.max() < 0.08 * b.abs().max() + 1e-2, (k, float(e.mean()), float(e.max()))


def _head_io(det, il):
    with torch.no_grad():
        feats = list(det.backbone(il.tensors).values())
        ho = det.head(feats)
    return feats, ho


def test_head_outputs_and_anchors(dev, case):
    det, oracle, images, targets = case
    il, _ = det.transform(images.to(dev), None)
    feats, ho = _head_io(det, il)
    ofeats = [nchw(t) for t in feats]
    with torch.no_grad():
        oho = oracle.head(ofeats)
    A = 9 * (38 * 38 + 19 * 19 + 100 + 25 + 9)
    assert ho["cls_logits"].shape == (3, A, 2) and ho["bbox_regression"].shape == (3, A, 4)
    for k in ho:
        a, b = ho[k].cpu(), oho[k]
        # 5 convs deep in fp16 storage, fp32 accumulate on both sides
        assert (a - b).abs().max() < 4e-3 * max(1.0, float(b.abs().max())), (k, float((a - b).abs().max()), float(b.abs().max()))
    anchors = det.anchor_generator(il, feats)
    oanchors = oracle.anchor_generator(od.ImageList(torch.zeros(3, 3, 300, 300), [(300, 300)] * 3), ofeats)
    assert torch.equal(anchors[0].cpu(), oanchors[0]) and anchors[0].shape == (A, 4)


def test_losses_exact_on_same_head_outputs(dev, case):
    """Matching, focal loss and smooth-L1 on IDENTICAL fp32 head outputs: list-based (reference-shaped) functions and the
    batched form both agree with the oracle to fp32 summation-order accuracy; matcher codes are identical."""
    from hallucidet_amd.models import detection as D, retinanet as R
    from hallucidet_amd.utils import eval_forward_retinanet as G
    det, oracle, images, targets = case
    il, tg = det.transform(images.to(dev), _t2d(targets, dev))
    _, otg = oracle.transform(images, targets)
    feats, ho = _head_io(det, il)
    anchors = det.anchor_generator(il, feats)
    oho = {k: v.cpu() for k, v in ho.items()}
    oanchors = [a.cpu() for a in anchors]
    want = orn.compute_retinanet_loss(otg, oho, oanchors, oracle)
    got_list = G.compute_retinanet_loss(tg, ho, anchors, det)
    gt, glab, gvalid = D.pad_targets(tg, dev)
    m = R.retinanet_match_batched(det, anchors[0], gt, gvalid)
    for i, t in enumerate(otg):
        om = torch.full((oanchors[0].shape[0],), -1, dtype=torch.int64) if t["boxes"].numel() == 0 else \
            oracle.proposal_matcher(od.ok.box_iou(t["boxes"], oanchors[0]))
        assert torch.equal(m[i].cpu(), om), i
        assert (om >= 0).sum() > 0 or t["boxes"].numel() == 0
    got_b = R.retinanet_loss_batched(det, anchors[0], gt, glab, gvalid, ho["cls_logits"], ho["bbox_regression"])
    for k in ("classification", "bbox_regression"):
        assert torch.allclose(got_list[k].cpu(), want[k], rtol=1e-5, atol=1e-6), (k, float(got_list[k]), float(want[k]))
        assert torch.allclose(got_b[k].cpu(), want[k], rtol=1e-5, atol=1e-6), (k, float(got_b[k]), float(want[k]))


def test_postprocess_exact_on_same_head_outputs(dev, case):
    det, oracle, images, targets = case
    il, _ = det.transform(images.to(dev), None)
    feats, ho = _head_io(det, il)
    anchors = det.anchor_generator(il, feats)
    napl = [9 * f.shape[1] * f.shape[2] for f in feats]
    split = {k: list(v.split(napl, dim=1)) for k, v in ho.items()}
    got = det.postprocess_detections(split, [list(a.split(napl)) for a in anchors], il.image_sizes)
    osplit = {k: [t.cpu() for t in v] for k, v in split.items()}
    want = oracle.postprocess_detections(osplit, [list(a.cpu().split(napl)) for a in anchors], [(300, 300)] * 3)
    sb, ss, sl, counts = det.postprocess_detections_padded(ho["cls_logits"], ho["bbox_regression"], anchors[0], napl, (300, 300))
    n_nontrivial = 0
    for i, (g, w) in enumerate(zip(got, want)):
        assert torch.equal(g["labels"].cpu(), w["labels"]), i
        assert torch.equal(g["scores"].cpu(), w["scores"]) and torch.allclose(g["boxes"].cpu(), w["boxes"], atol=1e-4)
        c = int(counts[i])
        assert c == w["labels"].numel() and torch.equal(sl[i, :c].cpu(), w["labels"]) and torch.equal(ss[i, :c].cpu(), w["scores"])
        assert torch.allclose(sb[i, :c].cpu(), w["boxes"], atol=1e-4)
        n_nontrivial += int(0 < c)
    assert n_nontrivial == 3


def test_end_to_end_losses_detections_and_image_gradient(dev, case):
    """eval_forward_retinanet (batched heads) vs the oracle end to end, and the data gradient w.r.t. the input images
    against the oracle's autograd with shared rounding (fp16 storage => statistical tolerance, see test_unet_gpu)."""
    from hallucidet_amd.models.detector import Detector
    from _pins import record, grad_agreement
    det, oracle, images, targets = case
    x = images.to(dev).requires_grad_(True)
    with record(det) as rec:
        losses, dets = Detector.calculate_loss(det, x, _t2d(targets, dev), train_det=False, model_name="retinanet")
    assert set(losses) == {"classification", "bbox_regression"}
    (losses["classification"] + losses["bbox_regression"]).backward()
    # the oracle takes every ReLU / max-pool decision from the product's activations (tests/_pins.py): same piecewise-linear
    # network on both sides, so losses and the image gradient differ by fp16 storage and summation order only
    pins = rec.pins()
    assert len(pins.masks) == 1 + 3 * 16 + 1 + 2 * 5 * 4           # stem, bottlenecks, ReLU(P6), two towers x 5 levels x 4 convs
    oracle.set_pins(pins)
    try:
        ox = images.clone().requires_grad_(True)
        olosses, odets = orn.eval_forward_retinanet(oracle, ox, targets, train_det=False)
    finally:
        oracle.set_pins(None)
    assert pins.used == set(pins.masks)
    from _pins import assert_borrowed_decisions_are_noise
    assert_borrowed_decisions_are_noise(pins, "retinanet")
    (olosses["classification"] + olosses["bbox_regression"]).backward()
    for k in losses:
        print("retinanet %s: product %.6f oracle %.6f" % (k, float(losses[k]), float(olosses[k])))
        assert abs(float(losses[k]) - float(olosses[k])) < 5e-3 * abs(float(olosses[k])) + 1e-5, (k, float(losses[k]), float(olosses[k]))
    assert len(dets) == 3 and all(d["boxes"].shape[0] <= 300 for d in dets)
    for d in dets:
        assert d["boxes"].shape[1] == 4 and d["labels"].dtype == torch.int64 and float(d["boxes"].max()) <= 128.0 + 1e-3
    g, og = x.grad.cpu(), ox.grad
    assert torch.isfinite(g).all() and float(og.abs().max()) > 0
    cos, rel = grad_agreement(g, og)
    print("retinanet image gradient: rel-L2 %.4f cosine %.5f" % (rel, cos))
    assert cos >= 0.999 and rel <= 0.03, (rel, cos)
    # list-based (reference-shaped) path gives the same numbers as the batched one
    det.batched_heads = False
    try:
        with torch.no_grad():
            l2, d2 = Detector.calculate_loss(det, images.to(dev), _t2d(targets, dev), train_det=False, model_name="retinanet")
    finally:
        det.batched_heads = True
    for k in losses:
        assert torch.allclose(l2[k], losses[k].detach(), rtol=1e-5, atol=1e-6), k
    for a, b in zip(d2, dets):
        assert torch.equal(a["labels"], b["labels"]) and torch.allclose(a["boxes"], b["boxes"], atol=1e-4)


def test_three_pass_fusion_equals_three_single_passes(dev, case, pinned_tiles):
    from hallucidet_amd.utils.eval_forward_retinanet import eval_forward_retinanet, eval_forward_retinanet_multi
    det, _, images, targets = case
    tg = _t2d(targets, dev)
    a = images.to(dev).requires_grad_(True)
    b, c = torch.rand_like(images).to(dev), torch.rand_like(images).to(dev)
    out = eval_forward_retinanet_multi(det, [a, b, c], [tg, tg, tg])
    (out[0][0]["classification"] + out[0][0]["bbox_regression"]).backward()
    g_multi = a.grad.clone()
    a2 = images.to(dev).requires_grad_(True)
    l1, d1 = eval_forward_retinanet(det, a2, tg)
    (l1["classification"] + l1["bbox_regression"]).backward()
    for k in l1:
        assert torch.allclose(l1[k], out[0][0][k], rtol=1e-5, atol=1e-6), k
    assert torch.allclose(g_multi, a2.grad, rtol=1e-3, atol=1e-6 + 1e-3 * float(a2.grad.abs().max()))
    with torch.no_grad():
        singles = [d1, eval_forward_retinanet(det, b, tg)[1], eval_forward_retinanet(det, c, tg)[1]]
    for (_, dm), ds in zip(out, singles):
        for x, y in zip(dm, ds):
            assert torch.equal(x["labels"], y["labels"]) and torch.equal(x["scores"], y["scores"]) and torch.equal(x["boxes"], y["boxes"])


def test_loss_kernels_against_reference_vectors_and_oracle_gradients(dev):
    """hd_sigmoid_focal_loss / hd_retinanet_loss behind the reference-shaped `sigmoid_focal_loss` and `box_loss`:
    values against vectors produced by the REFERENCE's own functions (tests/golden/glue_retinanet.npz, made by
    tests/golden/make_golden.py), gradients against the oracle's autograd."""
    import os
    import numpy as np
    from hallucidet_amd.utils import eval_forward_retinanet as G
    z = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(os.path.dirname(__file__), "golden", "glue_retinanet.npz")).items()}
    x, t = z["focal.x"].to(dev), z["focal.t"].to(dev)
    assert torch.allclose(G.sigmoid_focal_loss(x, t).cpu(), z["focal.none"], rtol=1e-5, atol=1e-7)
    assert torch.allclose(G.sigmoid_focal_loss(x, t, reduction="sum").cpu(), z["focal.sum"], rtol=1e-5)
    assert torch.allclose(G.sigmoid_focal_loss(x, t, alpha=-1, gamma=0, reduction="mean").cpu(), z["focal.mean_a-1_g0"], rtol=1e-5)
    with pytest.raises(ValueError, match="Invalid Value for arg 'reduction'"):
        G.sigmoid_focal_loss(x, t, reduction="max")
    # gradients: a larger random case incl. saturated logits, two gammas
    g = torch.Generator().manual_seed(3)
    xx = torch.randn(257, 3, generator=g) * 4
    tt = (torch.rand(257, 3, generator=g) > 0.8).float()
    for gamma, alpha in ((2.0, 0.25), (1.5, 0.6), (0.0, -1.0)):
        xo = xx.clone().requires_grad_(True)
        (orn.sigmoid_focal_loss(xo, tt, alpha=alpha, gamma=gamma, reduction="none") * torch.arange(1, 4)).sum().backward()
        xg = xx.to(dev).requires_grad_(True)
        (G.sigmoid_focal_loss(xg, tt.to(dev), alpha=alpha, gamma=gamma, reduction="none") * torch.arange(1, 4, device=dev)).sum().backward()
        assert torch.allclose(xg.grad.cpu(), xo.grad, rtol=2e-4, atol=1e-6), (gamma, alpha)
    bc = od.BoxCoder((1.0,) * 4)
    a, gts, br = z["boxloss.anchors"].to(dev), z["boxloss.gts"].to(dev), z["boxloss.breg"].to(dev)
    assert torch.allclose(G.box_loss("smooth_l1", bc, a, gts, br).cpu(), z["boxloss.smooth_l1"], rtol=1e-5)
    assert torch.allclose(G.box_loss("l1", bc, a, gts, br).cpu(), z["boxloss.l1"], rtol=1e-5)
    assert torch.allclose(G.box_loss("smooth_l1", bc, a, gts, br, cnf={"beta": 0.3}).cpu(),
                          torch.nn.functional.smooth_l1_loss(z["boxloss.breg"], bc.encode_single(z["boxloss.gts"], z["boxloss.anchors"]), reduction="sum", beta=0.3), rtol=1e-5)
    bo = z["boxloss.breg"].clone().requires_grad_(True)
    torch.nn.functional.smooth_l1_loss(bo, bc.encode_single(z["boxloss.gts"], z["boxloss.anchors"]), reduction="sum", beta=1.0).backward()
    bg = br.clone().requires_grad_(True)
    G.box_loss("smooth_l1", bc, a, gts, bg).backward()
    assert torch.allclose(bg.grad.cpu(), bo.grad, rtol=1e-5, atol=1e-7)
    with pytest.raises(Exception, match="Unsupported loss"):
        G.box_loss("huber", bc, a, gts, br)


def test_degenerate_box_raises_one_call_later_in_the_fused_path(dev, case):
    """The fused three-pass evaluation keeps the reference's degenerate-box assertion (eval_forward_retinanet.py:108-120) but reads
    the device flag without a host synchronisation inside the step: the call that carries the bad box completes, the NEXT call
    raises with the reference's message."""
    from hallucidet_amd.utils.eval_forward_retinanet import eval_forward_retinanet_multi
    det, _, images, targets = case
    det.__dict__.pop("_pending_degenerate", None)
    tg = _t2d(targets, dev)
    bad = [dict(t) for t in tg]
    bad[0] = {"boxes": torch.tensor([[10.0, 10.0, 10.0, 40.0]], device=dev), "labels": torch.ones(1, dtype=torch.int64, device=dev)}
    x = images.to(dev)
    with torch.no_grad():
        eval_forward_retinanet_multi(det, [x, x, x], [bad, tg, tg])              # carries the flag, does not raise
        with pytest.raises(AssertionError, match="All bounding boxes should have positive height and width"):
            eval_forward_retinanet_multi(det, [x, x, x], [tg, tg, tg])
        eval_forward_retinanet_multi(det, [x, x, x], [tg, tg, tg])               # the flag was consumed
    det.__dict__.pop("_pending_degenerate", None)
