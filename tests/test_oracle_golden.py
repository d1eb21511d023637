"""Pin the CPU oracle (oracle/) against fixtures produced by importing the reference's own files
(tests/golden/make_golden.py) and against hand-worked known answers for the torchvision-side pieces that cannot be
imported anywhere in this environment."""
import importlib.util
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import detection as od
from oracle import kernels as ok
from oracle import unet as ou

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _mg():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(G, "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def npz(name):
    return {k: torch.from_numpy(v) if v.dtype.kind in "fiub" else v for k, v in np.load(os.path.join(G, name), allow_pickle=False).items()}


# ---------------------------------------------------------------------------------- U-Net decoder / head
def test_decoder_head_forward_backward_matches_reference():
    mg = _mg()
    z = npz("decoder_small.npz")
    dec = ou.UnetDecoder(mg.ENC_SMALL, mg.DEC_SMALL)
    head = torch.nn.Sequential(torch.nn.Conv2d(mg.DEC_SMALL[-1], 3, 3, padding=1), torch.nn.Identity(), torch.nn.Sigmoid())
    dec.load_state_dict({k[len("sd.decoder."):]: v for k, v in z.items() if k.startswith("sd.decoder.")})
    head.load_state_dict({k[len("sd.segmentation_head."):]: v for k, v in z.items() if k.startswith("sd.segmentation_head.")})
    dec.train()
    feats = [z["feat%d" % i].clone().requires_grad_(True) for i in range(6)]
    out = head(dec(*feats))
    assert torch.allclose(out, z["out"], atol=1e-6, rtol=1e-5)
    out.backward(z["gout"])
    for i in range(1, 6):
        assert torch.allclose(feats[i].grad, z["gfeat%d" % i], atol=1e-6, rtol=1e-4), i
    for n, p in dec.named_parameters():
        assert torch.allclose(p.grad, z["grad.decoder." + n], atol=1e-5, rtol=1e-4), n
    for n, p in head.named_parameters():
        assert torch.allclose(p.grad, z["grad.segmentation_head." + n], atol=1e-5, rtol=1e-4), n


def test_upsample_matches_reference():
    z = npz("upsample.npz")
    assert torch.equal(ou.upsample2(z["x"]), z["y"])
    assert torch.equal(ok.upsample_deterministic(z["x"], 2), z["y"])


def test_initialisation_matches_reference_checksums():
    rec = json.load(open(os.path.join(G, "init_checksums.json")))
    torch.manual_seed(123)
    dec = ou.UnetDecoder()
    head = torch.nn.Sequential(torch.nn.Conv2d(16, 3, 3, padding=1), torch.nn.Identity(), torch.nn.Identity())
    ou.initialize_decoder(dec)
    ou.initialize_head(head)
    assert sum(p.numel() for p in dec.parameters()) == rec["_n_params"]["decoder"] == 3151552
    assert sum(p.numel() for p in head.parameters()) == rec["_n_params"]["head"] == 435
    for prefix, m in (("decoder.", dec), ("segmentation_head.", head)):
        for k, v in m.state_dict().items():
            r = rec[prefix + k]
            assert list(v.shape) == r["shape"], k
            assert math.isclose(float(v.double().sum()), r["sum"], rel_tol=1e-9, abs_tol=1e-9), k
            assert math.isclose(float(v.double().abs().sum()), r["abssum"], rel_tol=1e-9, abs_tol=1e-9), k


def test_shape_check_message_matches_reference():
    rec = json.load(open(os.path.join(G, "shape_error.json")))
    net = ou.Unet()
    with pytest.raises(RuntimeError) as e:
        net(torch.zeros(1, 3, *rec["shape"]))
    assert str(e.value) == rec["message"]


def test_unet_parameter_count_and_keys():
    net = ou.Unet()
    n = sum(p.numel() for p in net.parameters())
    assert n == 24436659  # SURVEY/BASELINE: 21 284 672 enc + 3 151 552 dec + 435 head
    keys = set(net.state_dict().keys())
    for k in ("encoder.conv1.weight", "encoder.bn1.running_mean", "encoder.layer2.0.downsample.0.weight",
              "encoder.layer4.2.bn2.weight", "decoder.blocks.0.conv1.0.weight", "decoder.blocks.4.conv2.1.running_var",
              "segmentation_head.0.weight", "segmentation_head.0.bias"):
        assert k in keys, k
    y = net.eval()(torch.rand(1, 3, 64, 96))
    assert y.shape == (1, 3, 64, 96) and float(y.detach().min()) > 0 and float(y.detach().max()) < 1


# ---------------------------------------------------------------------------------- detector transform
def test_transform_matches_reference():
    z = npz("transform.npz")
    t = od.FixedSizeTransform(300).eval()
    idx = torch.arange(512 * 640, dtype=torch.float32).view(1, 512, 640).repeat(3, 1, 1)
    boxes = z["boxes_in"]
    il, tg = t([idx, idx.flip(-1)], [{"boxes": boxes, "labels": torch.ones(3, dtype=torch.int64)},
                                     {"boxes": boxes[:1], "labels": torch.ones(1, dtype=torch.int64)}])
    assert torch.equal(il.tensors[0, 0].to(torch.int32), z["src_index"])
    assert torch.equal(il.tensors[1, 0].to(torch.int32), z["src_index_flipped"])
    assert [list(s) for s in il.image_sizes] == z["image_sizes"].tolist()
    assert torch.equal(tg[0]["boxes"], z["boxes_out0"]) and torch.equal(tg[1]["boxes"], z["boxes_out1"])
    post = t.postprocess([{"boxes": tg[0]["boxes"].clone()}, {"boxes": tg[1]["boxes"].clone()}], il.image_sizes, [(512, 640)] * 2)
    assert torch.equal(post[0]["boxes"], z["post0"]) and torch.equal(post[1]["boxes"], z["post1"])
    # SURVEY 0.6 known answer
    assert torch.allclose(tg[0]["boxes"][0], torch.tensor([4.6875, 11.71875, 51.5625, 128.90625]))
    # exactly 90 000 source pixels are selected (App. A.14b)
    assert z["src_index"].unique().numel() == 90000
    # other fixed size + float64 boxes stay float64
    t2 = od.FixedSizeTransform(24).eval()
    il2, tg2 = t2([z["small_img"]], [{"boxes": z["small_boxes_in"], "labels": torch.ones(1, dtype=torch.int64)}])
    assert torch.equal(il2.tensors, z["small_out"])
    assert str(tg2[0]["boxes"].dtype) == str(z["small_boxes_out_dtype"]) == "torch.float64"
    assert torch.equal(tg2[0]["boxes"], z["small_boxes_out"])


def test_config_defaults_recorded():
    rec = json.load(open(os.path.join(G, "config_defaults.json")))
    assert rec["loss_weights"]["det_regression"] == 0.1 and rec["loss_weights"]["pixel_rgb"] == 0.0
    assert rec["optimizer_name"] == "adam" and rec["n_gpus"] == 1 and rec["decoder_head"] == "sigmoid"
    assert rec["args"]["seed"] == 123 and rec["args"]["precision"] == 32


# ---------------------------------------------------------------------------------- orchestration
def test_oracle_orchestration_equals_reference_glue():
    """The fixture was produced by the REFERENCE's eval_forward_fasterrcnn.py driving the oracle detector object; the
    oracle's own orchestration must reproduce it exactly (same weights by seed, same injected sampler permutations)."""
    mg = _mg()
    z = npz("glue_fasterrcnn.npz")
    model, images, targets, perm_log, _ = mg.make_detector_case()
    assert torch.equal(images, z["images"])
    losses, dets = od.eval_forward_fasterrcnn(model, images, targets, train_det=False)
    assert len(perm_log) == int(z["n_perm"]) == 8  # 2 images x (RPN pos/neg + RoI pos/neg)
    for k in ("loss_classifier", "loss_box_reg", "loss_objectness", "loss_rpn_box_reg"):
        assert torch.allclose(losses[k], z["loss." + k], rtol=1e-6, atol=1e-7), k
    for i, d in enumerate(dets):
        assert torch.equal(d["labels"], z["det%d.labels" % i])
        assert torch.allclose(d["boxes"], z["det%d.boxes" % i], rtol=1e-6, atol=1e-5)
        assert torch.allclose(d["scores"], z["det%d.scores" % i], rtol=1e-6, atol=1e-7)
        assert d["boxes"].shape[0] <= 100 and d["labels"].dtype == torch.int64


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference tree only exists in the build container")
def test_reference_glue_live():
    mg = _mg()
    ref = mg.load_reference()
    model, images, targets, _, _ = mg.make_detector_case(seed=11)
    l_ref, d_ref = ref.glue.eval_forward_fasterrcnn(model, images, targets, train_det=False)
    model2, images2, targets2, _, _ = mg.make_detector_case(seed=11)
    l_or, d_or = od.eval_forward_fasterrcnn(model2, images2, targets2, train_det=False)
    for k in l_ref:
        assert torch.allclose(l_ref[k], l_or[k], rtol=1e-6, atol=1e-7), k
    for a, b in zip(d_ref, d_or):
        assert torch.equal(a["labels"], b["labels"]) and torch.allclose(a["boxes"], b["boxes"], atol=1e-5)


# ---------------------------------------------------------------------------------- RetinaNet (config 4)
def test_retinanet_losses_equal_reference():
    from oracle import retinanet as orn
    z = npz("glue_retinanet.npz")
    x, t = z["focal.x"], z["focal.t"]
    assert torch.allclose(orn.sigmoid_focal_loss(x, t), z["focal.none"], rtol=1e-6, atol=1e-7)
    assert torch.allclose(orn.sigmoid_focal_loss(x, t, reduction="sum"), z["focal.sum"], rtol=1e-6)
    assert torch.allclose(orn.sigmoid_focal_loss(x, t, alpha=-1, gamma=0, reduction="mean"), z["focal.mean_a-1_g0"], rtol=1e-6)
    bc = od.BoxCoder((1.0,) * 4)
    tr = bc.encode_single(z["boxloss.gts"], z["boxloss.anchors"])
    assert torch.allclose(torch.nn.functional.smooth_l1_loss(z["boxloss.breg"], tr, reduction="sum", beta=1.0), z["boxloss.smooth_l1"], rtol=1e-6)
    assert torch.allclose(torch.nn.functional.l1_loss(z["boxloss.breg"], tr, reduction="sum"), z["boxloss.l1"], rtol=1e-6)


def test_retinanet_orchestration_equals_reference_glue():
    """Fixture = the REFERENCE's eval_forward_retinanet.py over the oracle RetinaNet (one image with boxes, one without)."""
    from oracle import retinanet as orn
    mg = _mg()
    z = npz("glue_retinanet.npz")
    model, images, targets = mg.make_retinanet_case()
    assert torch.equal(images, z["images"])
    losses, dets = orn.eval_forward_retinanet(model, images, targets, train_det=False)
    assert set(losses) == {"classification", "bbox_regression"}
    for k in losses:
        assert torch.allclose(losses[k], z["loss." + k], rtol=1e-6, atol=1e-7), k
    for i, d in enumerate(dets):
        assert torch.equal(d["labels"], z["det%d.labels" % i])
        assert torch.allclose(d["boxes"], z["det%d.boxes" % i], rtol=1e-6, atol=1e-5)
        assert torch.allclose(d["scores"], z["det%d.scores" % i], rtol=1e-6, atol=1e-7)
        assert d["boxes"].shape[0] <= 300


def test_retinanet_structure_known_answers():
    from oracle import retinanet as orn
    m = orn.RetinaNet(num_classes=2, size=300)
    ag = m.anchor_generator
    assert ag.sizes == ((32, 40, 50), (64, 80, 101), (128, 161, 203), (256, 322, 406), (512, 645, 812))
    assert ag.num_anchors_per_location() == [9] * 5
    il = od.ImageList(torch.zeros(1, 3, 300, 300), [(300, 300)])
    feats = [torch.zeros(1, 256, s, s) for s in (38, 19, 10, 5, 3)]
    assert ag(il, feats)[0].shape == (9 * (38 * 38 + 19 * 19 + 100 + 25 + 9), 4)
    cl = m.head.classification_head.cls_logits
    assert cl.weight.shape == (18, 256, 3, 3) and torch.allclose(cl.bias, torch.full((18,), -math.log(99.0)))
    keys = set(m.state_dict().keys())
    for k in ("backbone.fpn.extra_blocks.p6.weight", "backbone.fpn.extra_blocks.p7.bias", "backbone.fpn.inner_blocks.2.weight",
              "head.classification_head.conv.6.weight", "head.regression_head.bbox_reg.bias", "backbone.body.layer4.2.bn3.running_var"):
        assert k in keys, k
    assert not any(k.startswith("backbone.fpn.inner_blocks.3") for k in keys)
    f = m.backbone(torch.rand(1, 3, 300, 300))
    assert [tuple(v.shape[-2:]) for v in f.values()] == [(38, 38), (19, 19), (10, 10), (5, 5), (3, 3)] and list(f) == ["0", "1", "2", "p6", "p7"]


# ---------------------------------------------------------------------------------- FCOS (row f4)
def test_fcos_orchestration_equals_reference_glue():
    """Fixture = the REFERENCE's eval_forward_fcos.py (src/utils/eval_forward_fcos.py:11-83) over the oracle FCOS: one image with
    three boxes, one without any, one with two nested boxes."""
    from oracle import fcos as ofc
    mg = _mg()
    z = npz("glue_fcos.npz")
    model, images, targets = mg.make_fcos_case()
    assert torch.equal(images, z["images"])
    losses, dets = ofc.eval_forward_fcos(model, images, targets, train_det=False)
    assert set(losses) == {"classification", "bbox_regression", "bbox_ctrness"}
    for k in losses:
        assert torch.allclose(losses[k], z["loss." + k], rtol=1e-6, atol=1e-7), k
    for i, d in enumerate(dets):
        assert torch.equal(d["labels"], z["det%d.labels" % i])
        assert torch.allclose(d["boxes"], z["det%d.boxes" % i], rtol=1e-6, atol=1e-5)
        assert torch.allclose(d["scores"], z["det%d.scores" % i], rtol=1e-6, atol=1e-7)
        assert d["boxes"].shape[0] <= 100


def test_fcos_structure_and_known_answers():
    """Hand-worked answers for the torchvision-side pieces of FCOS (parity unpinned against torchvision itself: absent)."""
    from oracle import fcos as ofc
    m = ofc.FCOS(num_classes=2, size=300)
    assert m.anchor_generator.num_anchors_per_location() == [1] * 5
    cl = m.head.classification_head.cls_logits
    assert cl.weight.shape == (2, 256, 3, 3) and torch.allclose(cl.bias, torch.full((2,), -math.log(99.0)))
    keys = set(m.state_dict().keys())
    for k in ("head.classification_head.conv.1.weight", "head.classification_head.conv.10.bias", "head.regression_head.conv.9.weight",
              "head.regression_head.bbox_ctrness.bias", "backbone.fpn.extra_blocks.p6.weight"):
        assert k in keys, k
    # BoxLinearCoder: ltrb from the anchor centre in anchor sizes, and its inverse
    bc = ofc.BoxLinearCoder(True)
    anchor = torch.tensor([[12.0, 20.0, 28.0, 36.0]])                       # centre (20, 28), size 16
    box = torch.tensor([[4.0, 20.0, 52.0, 44.0]])
    enc = bc.encode_single(anchor, box)
    assert torch.equal(enc, torch.tensor([[1.0, 0.5, 2.0, 1.0]]))
    assert torch.equal(bc.decode_single(enc, anchor), box)
    # generalized IoU: identical boxes -> 0; disjoint unit squares two apart -> 1 - (0 - (3 - 2) / 3) = 4/3
    a = torch.tensor([[0.0, 0.0, 1.0, 1.0]])
    assert float(ofc.generalized_box_iou_loss(a, a)) < 1e-6
    assert abs(float(ofc.generalized_box_iou_loss(a, torch.tensor([[2.0, 0.0, 3.0, 1.0]]))) - 4.0 / 3.0) < 1e-6
    # half-overlapping unit squares: iou 1/3, hull 1.5 -> 1 - (1/3 - 0) = 2/3
    assert abs(float(ofc.generalized_box_iou_loss(a, torch.tensor([[0.5, 0.0, 1.5, 1.0]]))) - 2.0 / 3.0) < 1e-6
    # target assignment on a 2-level toy: level 0 = stride 8 (size 8, range (0, 64)), level 1 = stride 16 (size 16, range (64, inf))
    l0 = torch.tensor([[x - 4.0, y - 4.0, x + 4.0, y + 4.0] for y in (4.0, 12.0, 20.0, 28.0) for x in (4.0, 12.0, 20.0, 28.0)])
    l1 = torch.tensor([[x - 8.0, y - 8.0, x + 8.0, y + 8.0] for y in (8.0, 24.0) for x in (8.0, 24.0)])
    anchors = torch.cat([l0, l1])
    t = {"boxes": torch.tensor([[6.0, 6.0, 26.0, 26.0], [10.0, 10.0, 22.0, 22.0]]), "labels": torch.tensor([1, 1])}
    got = m.match(anchors, t, [16, 4])
    # level-0 centres (12, 12), (20, 12), (12, 20), (20, 20) lie in both boxes and within 1.5 * 8 of both centres (16, 16): the
    # smaller box (index 1) wins; nothing else is strictly inside a box and close enough; level 1 needs a side distance > 64
    want = torch.full((20,), -1, dtype=torch.int64)
    want[[5, 6, 9, 10]] = 1
    assert torch.equal(got, want)
    assert torch.equal(m.match(anchors, {"boxes": torch.zeros(0, 4), "labels": torch.zeros(0, dtype=torch.int64)}, [16, 4]), torch.full((20,), -1))


# ---------------------------------------------------------------------------------- known answers (torchvision side)
def test_anchor_generator_known_answers():
    ag = od.AnchorGenerator()
    base = ag.base_anchors((32,), (0.5, 1.0, 2.0))
    assert torch.equal(base, torch.tensor([[-23., -11., 23., 11.], [-16., -16., 16., 16.], [-11., -23., 11., 23.]]))
    il = od.ImageList(torch.zeros(2, 3, 300, 300), [(300, 300)] * 2)
    feats = [torch.zeros(2, 256, s, s) for s in (75, 38, 19, 10, 5)]
    a = ag(il, feats)
    assert len(a) == 2 and a[0].shape == (22665, 4)  # 3*(75^2+38^2+19^2+10^2+5^2)
    # strides are integer floor divisions: 4,7,15,30,60
    assert torch.equal(a[0][3], torch.tensor([4. - 23, -11., 4. + 23, 11.]))  # second location, first anchor of level 0
    lvl1 = a[0][3 * 75 * 75:]
    assert torch.equal(lvl1[3], torch.tensor([7. - 45, -23., 7. + 45, 23.]))


def test_box_coder_roundtrip_and_known_answer():
    bc = od.BoxCoder((10.0, 10.0, 5.0, 5.0))
    prop = torch.tensor([[10.0, 10.0, 50.0, 90.0]])
    gt = torch.tensor([[20.0, 30.0, 60.0, 70.0]])
    code = bc.encode_single(gt, prop)
    want = torch.tensor([[10 * (40 - 30) / 40, 10 * (50 - 50) / 80, 5 * math.log(40 / 40), 5 * math.log(40 / 80)]])
    assert torch.allclose(code, want, atol=1e-6)
    assert torch.allclose(bc.decode_single(code, prop), gt, atol=1e-4)
    big = torch.tensor([[0.0, 0.0, 100.0, 100.0]])
    d = bc.decode_single(big, prop)
    assert torch.isfinite(d).all() and float(d[0, 2] - d[0, 0]) <= 40 * 1000 / 16 + 1e-3  # clamp at log(1000/16)


def test_matcher_known_answers():
    iou = torch.tensor([[0.9, 0.1, 0.45, 0.2, 0.0], [0.2, 0.35, 0.45, 0.6, 0.0]])
    m = od.Matcher(0.7, 0.3, allow_low_quality_matches=True)(iou.clone())
    # col0 -> gt0 (>=0.7); col1 0.35 between -> -2 ; col2 0.45 between; col3 0.6 between but is gt1's best -> restored to 1; col4 below -> -1
    assert m.tolist() == [0, -2, -2, 1, -1]
    m2 = od.Matcher(0.5, 0.5, allow_low_quality_matches=False)(iou.clone())
    assert m2.tolist() == [0, -1, -1, 1, -1]


def test_sampler_counts_and_injection():
    perms = []
    s = od.BalancedPositiveNegativeSampler(8, 0.25, randperm_fn=lambda n: (perms.append(n), torch.arange(n))[1])
    lab = torch.tensor([1, 0, 0, 1, 1, -1, 0, 0, 0, 0, 0, 0])
    pos, neg = s([lab])
    assert perms == [3, 8] and int(pos[0].sum()) == 2 and int(neg[0].sum()) == 6
    assert pos[0].tolist()[:5] == [1, 0, 0, 1, 0]


def test_nms_known_answers():
    b = torch.tensor([[0., 0., 10., 10.], [1., 1., 11., 11.], [20., 20., 30., 30.], [0., 0., 10., 10.]])
    s = torch.tensor([0.9, 0.8, 0.7, 0.6])
    assert od.nms(b, s, 0.5).tolist() == [0, 2]
    assert od.nms(b, s, 0.7).tolist() == [0, 1, 2]      # IoU(0,1)=81/119=0.68
    assert od.batched_nms(b, s, torch.tensor([0, 1, 0, 1]), 0.5).tolist() == [0, 1, 2]  # 3 falls to 1 (same class, IoU .68); 0 never suppresses 1
    assert od.batched_nms(b, s, torch.tensor([0, 1, 2, 3]), 0.5).tolist() == [0, 1, 2, 3]  # different classes never suppress
    assert od.batched_nms(b, s, torch.tensor([0, 0, 0, 0]), 0.5).tolist() == [0, 2]
    assert od.nms(torch.zeros(0, 4), torch.zeros(0), 0.5).numel() == 0


def test_roi_align_vectorised_equals_scalar():
    torch.manual_seed(0)
    feat = torch.randn(2, 4, 19, 19)
    rois = torch.tensor([[0, 10.0, 20.0, 110.0, 220.0], [1, 0.0, 0.0, 299.0, 299.0], [1, 150.3, 40.2, 160.9, 47.7],
                         [0, 280.0, 280.0, 330.0, 310.0], [0, -20.0, -5.0, 30.0, 60.0]])
    a = od.roi_align_autograd(feat, rois, 7, 1 / 16, 2)
    b = ok.roi_align_nchw(feat, rois, 7, 7, 1 / 16, 2)
    assert torch.allclose(a, b, atol=1e-5)
    # constant feature map -> constant output wherever samples are inside
    c = od.roi_align_autograd(torch.ones(1, 1, 10, 10), torch.tensor([[0, 16.0, 16.0, 80.0, 80.0]]), 7, 1 / 16, 2)
    assert torch.allclose(c, torch.ones_like(c))


def test_level_mapper_and_scales():
    p = od.MultiScaleRoIAlign()
    feats = [torch.zeros(1, 1, s, s) for s in (75, 38, 19, 10)]
    assert [p.infer_scale(f, (300, 300)) for f in feats] == [0.25, 0.125, 0.0625, 0.03125]
    boxes = [torch.tensor([[0., 0., 32., 32.], [0., 0., 112., 112.], [0., 0., 224., 224.], [0., 0., 300., 300.], [0., 0., 111.9, 111.9]])]
    assert p.level_map(boxes, 2, 5).tolist() == [0, 1, 2, 2, 0]


def test_fastrcnn_and_rpn_loss_shapes():
    logits = torch.tensor([[2.0, 0.0], [0.0, 3.0], [1.0, 1.0]])
    reg = torch.zeros(3, 8)
    cls, box = od.fastrcnn_loss(logits, reg, [torch.tensor([0, 1, 0])], [torch.tensor([[0.0] * 4, [0.5, 0.0, 0.0, 0.0], [0.0] * 4])])
    want_cls = torch.nn.functional.cross_entropy(logits, torch.tensor([0, 1, 0]))
    assert torch.isclose(cls, want_cls)
    # smooth-l1 beta=1/9 on |0.5| = 0.5 - 0.5/9 ; / numel(labels)=3
    assert torch.isclose(box, torch.tensor((0.5 - 0.5 / 9) / 3))


def test_pins_from_own_decisions_reproduce_the_plain_forward_and_gradient():
    """oracle.detection.Pins (the test aid that shares the product's ReLU / max-pool decisions with the oracle): fed the oracle's
    OWN decisions it must change nothing -- same features, same input gradient -- for all three trunks / heads, and a flipped
    mask must change the result (the pins are really consumed)."""
    import torch.nn.functional as F
    from oracle import detection as od, retinanet as orn, fcos as ofc

    class Rec(od.Pins):
        def relu(self, tag, x):
            y = F.relu(x)
            self.masks[tag] = (y > 0).float()
            return y

        def maxpool3x3s2(self, x):
            y, idx = F.max_pool2d(x, 3, 2, 1, return_indices=True)
            W = x.shape[-1]
            ho = torch.arange(y.shape[2]).view(1, 1, -1, 1)
            wo = torch.arange(y.shape[3]).view(1, 1, 1, -1)
            self.pool = (idx // W - (2 * ho - 1)) * 3 + (idx % W - (2 * wo - 1))
            return y

    torch.manual_seed(0)
    x = torch.rand(2, 3, 64, 96)
    for model, run in ((od.FasterRCNN(2, 300), lambda m, t: list(m.rpn.head(list(m.backbone(t).values()))[0])),
                       (orn.RetinaNet(2, 300), lambda m, t: [m.head(list(m.backbone(t).values()))["cls_logits"]]),
                       (ofc.FCOS(2, 300), lambda m, t: [m.head(list(m.backbone(t).values()))["bbox_regression"]])):
        model.eval()
        rec = Rec()
        model.set_pins(rec)
        xa = x.clone().requires_grad_(True)
        ya = run(model, xa)
        sum(t.sum() for t in ya).backward()
        assert rec.pool is not None and int(rec.pool.min()) >= 0 and int(rec.pool.max()) <= 8
        pins = od.Pins(dict(rec.masks), rec.pool)
        model.set_pins(pins)
        xb = x.clone().requires_grad_(True)
        yb = run(model, xb)
        sum(t.sum() for t in yb).backward()
        assert pins.used == set(pins.masks) and len(pins.masks) >= 1 + 48 + 5
        for a, b in zip(ya, yb):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-6)
        assert float((xa.grad - xb.grad).abs().max()) <= 1e-5 * float(xa.grad.abs().max())       # summation order of unfold / gather vs max_pool2d's backward
        flipped = dict(rec.masks)
        flipped[("b", 1, 0, 2)] = 1.0 - flipped[("b", 1, 0, 2)]
        model.set_pins(od.Pins(flipped, rec.pool))
        yc = run(model, x)
        assert not torch.allclose(ya[0], yc[0], rtol=1e-3, atol=1e-3)
        model.set_pins(None)
        yd = run(model, x)
        assert torch.allclose(ya[0], yd[0], rtol=1e-6, atol=1e-6)


def test_roi_align_against_aten_grid_sample():
    """An INDEPENDENT statement of RoIAlign(aligned=False, sampling_ratio=2) on top of ATen's own bilinear sampler: torchvision's
    `bilinear_interpolate` puts pixel i at coordinate i and clamps to the border for samples in [-1, H], which is exactly
    `F.grid_sample(mode='bilinear', padding_mode='border', align_corners=True)` at the normalised coordinate 2 y / (H - 1) - 1;
    samples outside [-1, H] contribute zero.  The oracle's two RoIAlign forms (scalar loops, vectorised autograd form) must agree
    with it -- this pins the sample-point geometry (bin size, the (i + .5) / 2 offsets, the max(., 1) of the RoI extent, the /4)."""
    import torch.nn.functional as F
    torch.manual_seed(3)
    N, C, H, W, P, sr, scale = 2, 5, 19, 23, 7, 2, 1 / 16
    feat = torch.randn(N, C, H, W)
    g = torch.Generator().manual_seed(4)
    x1 = torch.rand(40, generator=g) * 330 - 20
    y1 = torch.rand(40, generator=g) * 280 - 20
    w = torch.rand(40, generator=g) * 200 + 1
    h = torch.rand(40, generator=g) * 160 + 1
    idx = torch.randint(0, N, (40,), generator=g).float()
    rois = torch.stack([idx, x1, y1, x1 + w, y1 + h], dim=1)
    rois = torch.cat([rois, torch.tensor([[0, 5.0, 5.0, 5.5, 5.2], [1, 100.0, 90.0, 100.0, 90.0]])])     # extent below one feature pixel
    out = torch.zeros(rois.shape[0], C, P, P)
    for r in range(rois.shape[0]):
        n = int(rois[r, 0])
        rsw, rsh, rew, reh = [float(rois[r, k]) * scale for k in (1, 2, 3, 4)]
        rw, rh = max(rew - rsw, 1.0), max(reh - rsh, 1.0)
        bw, bh = rw / P, rh / P
        ys = torch.tensor([rsh + ph * bh + (iy + 0.5) * bh / sr for ph in range(P) for iy in range(sr)])
        xs = torch.tensor([rsw + pw * bw + (ix + 0.5) * bw / sr for pw in range(P) for ix in range(sr)])
        gy, gx = torch.meshgrid(ys, xs, indexing="ij")
        inside = ((gy >= -1.0) & (gy <= H) & (gx >= -1.0) & (gx <= W)).float()
        grid = torch.stack([2 * gx / (W - 1) - 1, 2 * gy / (H - 1) - 1], dim=-1)[None]
        s = F.grid_sample(feat[n:n + 1], grid, mode="bilinear", padding_mode="border", align_corners=True)[0] * inside
        out[r] = s.reshape(C, P, sr, P, sr).mean(dim=(2, 4))
    a = od.roi_align_autograd(feat, rois, P, scale, sr)
    b = ok.roi_align_nchw(feat, rois, P, P, scale, sr)
    assert torch.allclose(a, out, atol=2e-5), float((a - out).abs().max())
    assert torch.allclose(b, out, atol=2e-5), float((b - out).abs().max())


def test_nms_box_coder_matcher_properties():
    """Properties that any correct statement of the torchvision pieces must have, checked on random inputs (independent of the
    oracle's own arithmetic): NMS -- the kept set is independent (no two kept boxes overlap above the threshold), maximal (every
    removed box overlaps a kept box of no lower score above the threshold) and greedy (the best-scoring box is kept); BoxCoder --
    decode(encode(gt, anchors), anchors) == gt; Matcher -- a matched GT is the arg-max IoU of its anchor, BELOW / BETWEEN codes sit
    in their IoU bands, and with low-quality matching every GT with any overlap keeps its best anchor(s)."""
    g = torch.Generator().manual_seed(11)
    for trial in range(5):
        n = 200
        xy = torch.rand(n, 2, generator=g) * 100
        wh = torch.rand(n, 2, generator=g) * 40 + 2
        boxes = torch.cat([xy, xy + wh], dim=1)
        scores = torch.rand(n, generator=g)
        thr = 0.3 + 0.1 * trial
        keep = od.nms(boxes, scores, thr)
        kept = torch.zeros(n, dtype=torch.bool)
        kept[keep] = True
        iou = ok.box_iou(boxes, boxes)
        kk = iou[keep][:, keep]
        assert float((kk - torch.eye(len(keep))).max()) <= thr + 1e-6            # independent
        assert int(keep[0]) == int(scores.argmax())                               # greedy start, descending order
        assert bool((scores[keep][:-1] >= scores[keep][1:]).all())
        for i in (~kept).nonzero().flatten().tolist():                            # maximal
            sup = (iou[i, keep] > thr) & (scores[keep] >= scores[i])
            assert bool(sup.any()), i
    coder = od.BoxCoder((10.0, 10.0, 5.0, 5.0))
    a_xy = torch.rand(50, 2, generator=g) * 200
    anchors = torch.cat([a_xy, a_xy + torch.rand(50, 2, generator=g) * 80 + 4], dim=1)
    g_xy = torch.rand(50, 2, generator=g) * 200
    gt = torch.cat([g_xy, g_xy + torch.rand(50, 2, generator=g) * 80 + 4], dim=1)
    back = coder.decode_single(coder.encode_single(gt, anchors), anchors)
    assert torch.allclose(back, gt, atol=1e-3)
    m = od.Matcher(0.7, 0.3, allow_low_quality_matches=True)
    iou = ok.box_iou(gt[:6], anchors)
    res = m(iou)
    vals, arg = iou.max(dim=0)
    plain = od.Matcher(0.7, 0.3, allow_low_quality_matches=False)(iou)
    assert bool((plain[vals >= 0.7] == arg[vals >= 0.7]).all())
    assert bool((plain[vals < 0.3] == od.Matcher.BELOW_LOW_THRESHOLD).all())
    assert bool((plain[(vals >= 0.3) & (vals < 0.7)] == od.Matcher.BETWEEN_THRESHOLDS).all())
    best = iou.max(dim=1).values
    for gi in range(6):
        if best[gi] > 0:
            cols = (iou[gi] == best[gi]).nonzero().flatten()
            assert bool((res[cols] >= 0).all())                # low-quality rule: the best anchor(s) of every GT stay matched


@pytest.fixture
def no_torchvision_stub():
    """The live reference-glue tests above leave stand-in `torchvision*` modules (no __spec__) in sys.modules; transformers probes for
    torchvision with importlib.util.find_spec, which raises on those.  Hidden for the duration of a test, restored afterwards."""
    import sys
    hidden = {k: sys.modules.pop(k) for k in list(sys.modules)
              if (k == "torchvision" or k.startswith("torchvision.")) and getattr(sys.modules[k], "__spec__", None) is None}
    try:
        yield
    finally:
        sys.modules.update(hidden)


def test_box_iou_focal_loss_giou_against_the_detr_utilities_of_transformers(no_torchvision_stub):
    """Three torchvision ops of the detector half restated in oracle/ (box_iou of the matchers, sigmoid_focal_loss of RetinaNet / FCOS,
    generalized_box_iou_loss of FCOS) against implementations this repository did not write: the DETR loss utilities shipped in the
    installed `transformers` wheel (transformers/loss/loss_for_object_detection.py: box_iou, generalized_box_iou, sigmoid_focal_loss,
    descended from torchvision's).  torchvision itself cannot be installed here; this is the nearest externally held pin."""
    T = pytest.importorskip("transformers.loss.loss_for_object_detection")
    from oracle import fcos as ofc
    from oracle import retinanet as orn
    g = torch.Generator().manual_seed(11)

    def boxes(n, size=400.0):
        xy = torch.rand(n, 2, generator=g) * size
        wh = torch.rand(n, 2, generator=g) * size * 0.4 + 1.0
        return torch.cat([xy, xy + wh], dim=1)

    a, b = boxes(37), boxes(211)
    b[5] = a[3]                                          # an exact duplicate: IoU 1
    b[6] = torch.tensor([1000.0, 1000.0, 1010.0, 1010.0])  # disjoint from everything: IoU 0
    iou_t, _union = T.box_iou(a, b)
    assert torch.allclose(ok.box_iou(a, b), iou_t, rtol=0, atol=1e-6)
    assert float(ok.box_iou(a, b)[3, 5]) == 1.0 and float(ok.box_iou(a, b)[:, 6].max()) == 0.0
    # GIoU loss = 1 - GIoU of matched pairs (torchvision adds eps = 1e-7 to both denominators)
    p, q = boxes(300), boxes(300)
    want = 1.0 - torch.diag(T.generalized_box_iou(p, q))
    got = ofc.generalized_box_iou_loss(p, q, reduction="none")
    assert torch.allclose(got, want, rtol=0, atol=2e-6)
    assert abs(float(ofc.generalized_box_iou_loss(p, q, reduction="sum")) - float(want.sum())) < 1e-3
    # focal loss, element-wise: transformers reduces as mean over dim 1, summed, / num_boxes -> recover the sum
    x = torch.randn(1, 5000, generator=g) * 3
    t = (torch.rand(1, 5000, generator=g) < 0.1).float()
    for alpha, gamma in ((0.25, 2), (0.5, 2), (-1.0, 2), (0.25, 1)):
        want_sum = float(T.sigmoid_focal_loss(x, t, 1, alpha=alpha, gamma=gamma)) * x.shape[1]
        got_sum = float(orn.sigmoid_focal_loss(x, t, alpha=alpha, gamma=gamma, reduction="sum"))
        assert abs(got_sum - want_sum) <= 1e-4 * abs(want_sum) + 1e-4, (alpha, gamma, got_sum, want_sum)


def _hf_resnet(layer_type, hidden):
    R = pytest.importorskip("transformers.models.resnet.modeling_resnet")
    from transformers import ResNetConfig
    cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=list(hidden), depths=[3, 4, 6, 3], layer_type=layer_type,
                       hidden_act="relu", downsample_in_first_stage=False, downsample_in_bottleneck=False)
    torch.manual_seed(3)
    m = R.ResNetModel(cfg)
    g = torch.Generator().manual_seed(4)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):          # non-trivial affine + running statistics
            mod.weight.data = torch.rand(mod.num_features, generator=g) + 0.5
            mod.bias.data = torch.randn(mod.num_features, generator=g) * 0.1
            mod.running_mean.data = torch.randn(mod.num_features, generator=g) * 0.1
            mod.running_var.data = torch.rand(mod.num_features, generator=g) + 0.5
    return m


def _copy_bn(dst, src):
    for k in ("weight", "bias", "running_mean", "running_var"):
        getattr(dst, k).data.copy_(getattr(src, k).data)


def _load_from_hf(oracle_net, hf, nconv):
    """HF ResNetModel -> the oracle's torchvision-shaped module tree (conv1/bn1, layerN[j].convK/bnK/downsample)."""
    oracle_net.conv1.weight.data.copy_(hf.embedder.embedder.convolution.weight.data)
    _copy_bn(oracle_net.bn1, hf.embedder.embedder.normalization)
    for i, stage in enumerate(hf.encoder.stages):
        for j, lay in enumerate(stage.layers):
            blk = getattr(oracle_net, "layer%d" % (i + 1))[j]
            for k in range(nconv):
                getattr(blk, "conv%d" % (k + 1)).weight.data.copy_(lay.layer[k].convolution.weight.data)
                _copy_bn(getattr(blk, "bn%d" % (k + 1)), lay.layer[k].normalization)
            has_sc = hasattr(lay.shortcut, "convolution")
            assert has_sc == (blk.downsample is not None), (i, j)
            if has_sc:
                assert lay.shortcut.convolution.stride == blk.downsample[0].stride
                blk.downsample[0].weight.data.copy_(lay.shortcut.convolution.weight.data)
                _copy_bn(blk.downsample[1], lay.shortcut.normalization)


@pytest.mark.parametrize("block", ["basic", "bottleneck"])
def test_unet_encoder_against_the_resnet_of_transformers(block, no_torchvision_stub):
    """The oracle's ResNet-34 / ResNet-50 encoder (oracle/unet.py; the reference takes it from torchvision through
    src/segmentation_models/encoders/resnet.py:36-60) against an implementation held outside this repository: the ResNet of the installed
    `transformers` wheel (torchvision's v1.5 architecture: stride on the 3x3 of a bottleneck, no stride in stage 1), same weights, in
    eval mode (running statistics) and in training mode (batch statistics, as the hallucination net runs): the five feature maps and
    the input gradient must agree to fp32 round-off."""
    hf = _hf_resnet(block, (64, 128, 256, 512) if block == "basic" else (256, 512, 1024, 2048))
    enc = ou.ResNet34Encoder(block=block)
    _load_from_hf(enc, hf, 2 if block == "basic" else 3)
    x = torch.randn(2, 3, 64, 96, generator=torch.Generator().manual_seed(5))
    for train in (False, True):
        hf.train(train)
        enc.train(train)
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        stem = hf.embedder.embedder(xa)
        hs = hf.encoder(hf.embedder.pooler(stem), output_hidden_states=True).hidden_states
        feats = enc(xb)
        assert len(feats) == 6 and torch.equal(feats[0], xb)
        pairs = [(feats[1], stem)] + list(zip(feats[2:], hs[1:]))
        for lvl, (a, b) in enumerate(pairs):
            assert a.shape == b.shape, (lvl, a.shape, b.shape)
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * float(b.detach().abs().max())), (block, train, lvl, float((a - b).detach().abs().max()))
        w = torch.randn(feats[-1].shape, generator=torch.Generator().manual_seed(6))
        ga, = torch.autograd.grad((hs[-1] * w).sum(), xa)
        gb, = torch.autograd.grad((feats[-1] * w).sum(), xb)
        assert torch.allclose(ga, gb, rtol=1e-3, atol=1e-4 * float(ga.abs().max())), (block, train)


def test_detector_trunk_against_the_resnet_of_transformers(no_torchvision_stub):
    """The oracle's frozen ResNet-50 trunk of the three detectors (oracle/detection.py ResNet50Body + FrozenBatchNorm2d; the reference
    builds it with torchvision.models.detection.*_resnet50_fpn, src/models/detector.py:20-60) against transformers' ResNet-50 in eval
    mode with the same weights and statistics: the four stage outputs (strides 4 / 8 / 16 / 32) agree to fp32 round-off."""
    D = pytest.importorskip("transformers.models.detr.modeling_detr")
    g = torch.Generator().manual_seed(1)
    fa, fb = D.DetrFrozenBatchNorm2d(16), od.FrozenBatchNorm2d(16)      # torchvision's FrozenBatchNorm2d as DETR carries it (eps 1e-5 inside the rsqrt)
    for k in ("weight", "bias", "running_mean", "running_var"):
        v = torch.rand(16, generator=g) + 0.3
        getattr(fa, k).data.copy_(v)
        getattr(fb, k).data.copy_(v)
    xf = torch.randn(2, 16, 5, 7, generator=g)
    assert torch.allclose(fa(xf), fb(xf), rtol=0, atol=1e-6)
    hf = _hf_resnet("bottleneck", (256, 512, 1024, 2048)).eval()
    body = od.ResNet50Body()
    _load_from_hf(body, hf, 3)
    x = torch.randn(2, 3, 64, 96, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        hs = hf(x, output_hidden_states=True).hidden_states
        out = body(x, lambda t: t)
    assert list(out.keys()) == ["0", "1", "2", "3"]
    for i in range(4):
        a, b = out[str(i)], hs[i + 1]
        assert a.shape == b.shape == (2, 256 << i, 16 >> i, 24 >> i)
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max())), (i, float((a - b).abs().max()))
