"""BASELINE configs[1] at FULL size (8 x 512x640, 24 x 300x300 detector inputs): the CPU oracle cannot run these sizes in
seconds, so the hot path is checked through size-independent properties -- exact scaling by a power of two, batch
independence, graph replay == eager, sortedness / idempotence of the selection kernels, sampler invariants -- and, for the
hallucination network, directly against the oracle's module tree evaluated on the GPU (ATen fp32 operators), where the full size
takes seconds."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rnd(*shape, scale=1.0, seed=0, dev="cuda"):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half().to(dev)


@pytest.mark.parametrize("shape", [(8, 128, 160, 64, 64), (8, 512, 640, 16, 16), (8, 256, 320, 32, 32), (8, 32, 40, 256, 256),
                                   (24, 75, 75, 256, 256)])
def test_conv_full_size_scaling_and_batch_independence(dev, shape):
    """conv(0.5 x) == 0.5 conv(x) bit for bit (power-of-two scaling commutes with every fp16 / fp32 rounding as long as
    nothing under- or overflows), image n of a batched launch == the same image launched alone (the tile -> block mapping
    must not leak across images: bit for bit with the tile model pinned to the batched launch's batch, hd_conv_nominal_batch; to fp16
    rounding under the shipped per-launch rule, which may pick another tile for one image), BN partial sums == sums over the stored output."""
    from hallucidet_amd import ops
    N, H, W, Cin, Cout = shape
    x = _rnd(N, H, W, Cin, seed=1, dev=dev)
    x = torch.where(x.abs() < 1e-2, torch.full_like(x, 0.5), x)      # no fp16 subnormals after halving
    w = _rnd(Cout, 9 * Cin, scale=1.0 / math.sqrt(9 * Cin), seed=2, dev=dev)
    w = torch.where(w.abs() < 1e-3, torch.full_like(w, 0.01), w)
    y, stats = ops.conv2d(x, w, 3, 3, pad=1, want_stats=True)
    y_half = ops.conv2d(x * 0.5, w, 3, 3, pad=1)
    tiny = y.float().abs() < 1e-3                      # halves of subnormal-range outputs may round differently
    assert torch.equal(torch.where(tiny, torch.zeros_like(y), y * 0.5), torch.where(tiny, torch.zeros_like(y), y_half))
    from hallucidet_amd import _abi
    lib = _abi.load()
    for n in (0, N - 1):
        alone = ops.conv2d(x[n:n + 1].contiguous(), w, 3, 3, pad=1)[0]
        assert float((alone.float() - y[n].float()).abs().max()) <= 4e-3 * float(y[n].float().abs().max())
        lib.hd_conv_nominal_batch(N)
        try:
            assert torch.equal(ops.conv2d(x[n:n + 1].contiguous(), w, 3, 3, pad=1)[0], y[n])
        finally:
            lib.hd_conv_nominal_batch(0)
    s = stats.double().sum(0)
    yf = y.double().reshape(-1, Cout)
    assert torch.allclose(s[0], yf.sum(0), rtol=1e-5, atol=1e-2) and torch.allclose(s[1], (yf * yf).sum(0), rtol=1e-5, atol=1e-2)


def test_full_size_step_graph_replay_equals_eager_and_is_reproducible(dev):
    """One full-size training step (the bench workload): two modules built from the same seed give the same loss and the
    same flat gradient bit for bit, with the U-Net replayed from hipGraphs in one and launched eagerly in the other."""
    from hallucidet_amd import synthetic
    outs = []
    for use_graphs in (True, False):
        lit = synthetic.make_module(seed=321)
        lit.encoder_decoder.runner.enable_graphs(use_graphs)
        batch = synthetic.make_batch(8, device=dev)
        torch.manual_seed(5)
        lit.encoder_decoder.train()
        loss = lit.training_step(batch, 0)
        lit.scaler.scale(loss).backward()
        g = lit.encoder_decoder.runner.flat_grads
        assert torch.isfinite(loss) and bool(torch.isfinite(g).all())
        outs.append((float(loss.detach()), g.clone()))
        del lit
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][1].abs().sum()) > 0


def test_selection_kernels_at_full_proposal_counts(dev):
    """Pre-NMS top-k over [24, 21765] objectness (5 levels), batched NMS over 24 x 3375 candidates, sampler over
    [24, 21765] labels: sortedness, cut value, idempotence and counting invariants."""
    from hallucidet_amd import ops
    import hallucidet_amd.models.detection as D
    torch.manual_seed(4)
    segs = [16875, 4332, 1083, 300, 75]
    B, k = 24, 1000
    obj = torch.randn(B, sum(segs), device=dev)
    top = ops.topk_rows_segments(obj, segs, k)
    off = ooff = 0
    for n in segs:
        kk = min(k, n)
        idx = top[:, ooff:ooff + kk]
        assert int(idx.min()) >= off and int(idx.max()) < off + n
        sc = torch.gather(obj, 1, idx)
        assert bool((sc[:, :-1] >= sc[:, 1:]).all())                                   # descending
        assert bool((torch.sort(idx, dim=1)[0][:, 1:] != torch.sort(idx, dim=1)[0][:, :-1]).all())   # distinct
        if kk < n:                                                                     # nothing left out beats the cut
            rest = obj[:, off:off + n].clone().scatter_(1, idx - off, float("-inf"))
            assert bool((rest.max(dim=1).values <= sc[:, -1]).all())
        off += n
        ooff += kk
    # batched NMS: survivors are in descending score order, and NMS of the survivors alone keeps all of them
    n = top.shape[1]
    xy = torch.rand(B, n, 2, device=dev) * 260
    boxes = torch.cat([xy, xy + 4 + torch.rand(B, n, 2, device=dev) * 60], dim=2)
    scores = torch.rand(B, n, device=dev)
    lv = torch.zeros((B, n), dtype=torch.int64, device=dev)      # one category: the second pass then repeats the first pass's arithmetic exactly
    valid = torch.rand(B, n, device=dev) > 0.1
    pick, cnt = D._batched_nms_pick(boxes, scores, lv, valid, 0.7, 1000)
    c = cnt.tolist()
    assert all(0 < ci <= 1000 for ci in c)
    ps = torch.gather(scores, 1, pick)
    for b in (0, B - 1):
        assert bool((ps[b, :c[b] - 1] >= ps[b, 1:c[b]]).all())
        assert bool(torch.gather(valid, 1, pick)[b, :c[b]].all())
    kb = torch.gather(boxes, 1, pick[:, :, None].expand(-1, -1, 4))
    kl = torch.gather(lv, 1, pick)
    kvalid = torch.arange(pick.shape[1], device=dev)[None, :] < cnt[:, None]
    pick2, cnt2 = D._batched_nms_pick(kb, ps, kl, kvalid, 0.7, 1000)
    assert torch.equal(cnt2, cnt)
    assert all(torch.equal(pick2[b, :c[b]], torch.arange(c[b], device=dev)) for b in range(B))
    # sampler: members only, counts as torchvision defines them
    labels = torch.randint(-1, 2, (B, 21765), device=dev, dtype=torch.int64)
    labels[:, ::3] = torch.clamp(labels[:, ::3], max=0)
    keys = torch.randint(0, 1 << 30, labels.shape, dtype=torch.int32, device=dev)
    pos, neg, counts = ops.sample_pos_neg(labels, keys, 256, 128)
    assert bool((labels[pos] >= 1).all()) and bool((labels[neg] == 0).all())
    assert torch.equal(pos.sum(1), counts[:, 0]) and torch.equal(neg.sum(1), counts[:, 1])
    P, Nn = (labels >= 1).sum(1), (labels == 0).sum(1)
    assert torch.equal(counts[:, 0], P.clamp(max=128)) and torch.equal(counts[:, 1], torch.minimum(Nn, 256 - counts[:, 0]))


def test_full_size_retinanet16_step_graph_replay_equals_eager(dev):
    """BASELINE configs[3] at its stated size (train_hallucidet, RetinaNet, batch 16 per GPU, 512x640): graph replay == eager ==
    same seed rerun (loss and the whole flat gradient bit for bit), and the selection invariants of its post-processing at
    17 451 anchors per image: <= 300 detections per image in descending score order, every box inside its image."""
    from hallucidet_amd import synthetic
    outs = []
    for use_graphs in (True, False):
        lit = synthetic.make_module(seed=322, detector_name="retinanet")
        lit.encoder_decoder.runner.enable_graphs(use_graphs)
        batch = synthetic.make_batch(16, device=dev)
        torch.manual_seed(5)
        lit.encoder_decoder.train()
        loss = lit.training_step(batch, 0)
        lit.scaler.scale(loss).backward()
        g = lit.encoder_decoder.runner.flat_grads
        assert torch.isfinite(loss) and bool(torch.isfinite(g).all())
        dets = [{k: v.clone() for k, v in d.items()} for d in lit._last_detections["hall"]]
        outs.append((float(loss.detach()), g.clone(), dets))
        del lit
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1]) and float(outs[0][1].abs().sum()) > 0
    dets = outs[0][2]
    assert len(dets) == 16
    for d0, d1 in zip(dets, outs[1][2]):
        assert torch.equal(d0["boxes"], d1["boxes"]) and torch.equal(d0["labels"], d1["labels"])
        n = d0["boxes"].shape[0]
        assert n <= 300 and d0["labels"].dtype == torch.int64
        if n > 1:
            assert bool((d0["scores"][:-1] >= d0["scores"][1:]).all())
        if n:
            assert float(d0["boxes"][:, 0::2].max()) <= 640 + 1e-3 and float(d0["boxes"][:, 1::2].max()) <= 512 + 1e-3 and float(d0["boxes"].min()) >= 0.0


def test_full_size_detector16_training_step_is_reproducible(dev):
    """BASELINE configs[4] at its stated size (train_detector.py, Faster R-CNN, RGB, batch 16 per GPU): train-mode RPN keeps up
    to 2000 proposals per image from 2000 pre-NMS candidates per level; two runs from the same seed give the same loss and the
    same parameter-gradient arena bit for bit; the step moves the trainable parameters and nothing else."""
    from hallucidet_amd import synthetic
    from hallucidet_amd.models.detector import Detector
    from hallucidet_amd.train_detector import DetectorLit
    rgb, trgb, _, _ = synthetic.make_batch(16, device=dev)
    outs = []
    for _ in range(2):
        torch.manual_seed(9)
        det = Detector(name="fasterrcnn", pretrained=False, n_classes=2, size=300).detector.to(dev)
        il, _ = det.transform(rgb[:2], None)
        det.backbone.calibrate_(il.tensors)
        lit = DetectorLit(batch_size=16, detector=det, pretrained=False, device=str(dev)).prepare()
        frozen = {k: v.clone() for k, v in det.state_dict().items() if k.startswith("backbone.body.conv1") or k.startswith("backbone.body.layer1")}
        before = lit.arena.flat_params.clone()
        torch.manual_seed(10)
        loss = lit.fit_step((rgb, trgb))
        assert torch.isfinite(loss) and float(lit.optimizer.found_inf) == 0.0
        g = lit.arena.flat_grads
        assert bool(torch.isfinite(g).all()) and float(g.abs().sum()) > 0
        assert not torch.equal(lit.arena.flat_params, before)
        for k, v in frozen.items():
            assert torch.equal(det.state_dict()[k], v), k       # conv1 / layer1 stay fixed (trainable_layers = 3)
        dets = lit._last_detections
        assert len(dets) == 16 and all(d["boxes"].shape[0] <= 100 for d in dets)
        outs.append((float(loss), g.clone()))
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])


def test_config0_eval_batch_one_full_size(dev, pinned_tiles):
    """BASELINE configs[0] (eval_hallucidet.py:135-182: Faster R-CNN, LLVIP geometry, batch = 1) through the HIP path at full size:
    one 512x640 IR / RGB pair per test_step, eval-mode U-Net (running statistics) and detector.  Size-independent properties: the
    hallucinated image and the detector's FPN features of the image evaluated ALONE equal, bit for bit, those of the same image
    as member 1 of a batch of two (batch invariance of every kernel on the path); the three detection lists obey the reference's
    contract (<= 100 per image, boxes inside 640x512, scores descending); the mAP hook returns the three streams."""
    from hallucidet_amd import synthetic
    lit = synthetic.make_module(seed=77)
    with torch.no_grad():
        lit.detector.roi_heads.box_predictor.cls_score.weight.mul_(30.0)      # spread the random-init class scores
    lit.detector.invalidate_packs()
    lit.eval()
    rgb, trgb, ir, tir = synthetic.make_batch(2, device=dev, seed=78)
    one = (rgb[1:2].contiguous(), trgb[1:2], ir[1:2].contiguous(), tir[1:2])
    with torch.no_grad():
        out2 = lit.forward_step(rgb, trgb, ir, tir, 0, step="test")
        h2 = out2["output"]["imgs_hallucinated"].clone()
        il2, _ = lit.detector.transform(h2, None)
        f2 = {k: v.clone() for k, v in lit.detector.backbone(il2.tensors).items()}
        out1 = lit.forward_step(*one, 0, step="test")
        h1 = out1["output"]["imgs_hallucinated"]
        il1, _ = lit.detector.transform(h1, None)
        f1 = lit.detector.backbone(il1.tensors)
    assert h1.shape == (1, 3, 512, 640) and torch.equal(h1[0], h2[1])
    for k in f1:
        assert torch.equal(f1[k][0], f2[k][1]), k
    loss, dets = lit.test_step(one, 0)
    assert torch.isfinite(loss) and set(dets) == {"hall", "rgb", "ir"}
    for k in dets:
        assert len(dets[k]) == 1
        d = dets[k][0]
        assert d["boxes"].shape[0] <= 100 and d["labels"].dtype == torch.int64
        if d["boxes"].numel():
            assert float(d["boxes"][:, 0::2].max()) <= 640.0 + 1e-3 and float(d["boxes"][:, 1::2].max()) <= 512.0 + 1e-3 and float(d["boxes"].min()) >= 0.0
            assert bool((d["scores"][:-1] >= d["scores"][1:]).all())
    m = lit.on_test_epoch_end()
    assert set(m) == {"map_rgb", "map_hall", "map_ir"}


def test_full_size_unet_forward_backward_against_the_oracle_run_on_the_gpu(dev):
    """configs[1]'s hallucination network at its FULL size (8 x 3 x 512 x 640, training-mode BatchNorm, loss scale 1024) against the
    oracle's module tree (oracle/unet.py: plain nn.Conv2d / BatchNorm2d / max_pool2d, i.e. ATen's fp32 operators -- the operators the
    reference itself runs, src/segmentation_models/base/model.py:24-37) evaluated ON THE GPU, where it finishes in seconds: same
    weights, the product's fp16 rounding schedule and ReLU decisions (so that what is compared is the arithmetic, not the sign of a
    value next to zero).  Output, every parameter gradient, the BatchNorm running statistics."""
    from oracle import unet as ou
    from hallucidet_amd.models.encoder_decoder import EncoderDecoder
    torch.manual_seed(11)
    net = EncoderDecoder(name="resnet34", encoder_weights=None, in_channels=3, output_channels=3).encoder_decoder
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.Conv2d):
                m.weight.copy_(m.weight.half().float())
    ref = ou.Unet(classes=3)
    ref.load_state_dict(net.state_dict())
    net, ref = net.to(dev).train(), ref.to(dev).train()
    N, H, W, S = 8, 512, 640, 1024.0
    g = torch.Generator().manual_seed(12)
    x = torch.rand(N, 3, H, W, generator=g).to(dev)
    gout = (torch.randn(N, 3, H, W, generator=g) * 1e-2).to(dev)
    net.runner.grad_scale = S
    out = net(x)
    from _pins import unet_decisions
    masks, uvalues = unet_decisions(net.runner, device=x.device)
    (out * (gout * S)).sum().backward()
    torch.cuda.synchronize()
    torch.backends.cudnn.allow_tf32 = False
    uctx = ou.Ctx(ou.fp16_round, masks, uvalues)
    wq = ref(x, q=uctx)
    from _pins import assert_borrowed_decisions_are_noise
    assert_borrowed_decisions_are_noise(uctx, "U-Net")
    (wq * gout).sum().backward()
    torch.cuda.synchronize()
    e = (out.detach() - wq.detach()).abs()
    print("full-size U-Net output: mean |err| %.2e, max %.2e" % (float(e.mean()), float(e.max())))
    assert e.mean() < 4e-3 and e.max() < 5e-2, (float(e.mean()), float(e.max()))          # measured 2.3e-3 / 2.6e-2 (a sigmoid output in (0, 1))
    worst, worst_cos, name_w = 0.0, 1.0, ""
    for (n, p), (_, pw) in zip(net.named_parameters(), ref.named_parameters()):
        gg, w = p.grad.float(), pw.grad
        rel = float((gg - w).norm() / (w.norm() + 1e-12))
        cos = float(torch.nn.functional.cosine_similarity(gg.flatten(), w.flatten(), dim=0))
        if rel > worst:
            worst, name_w = rel, n
        worst_cos = min(worst_cos, cos)
    print("full-size U-Net parameter gradients: worst rel-L2 %.4f (%s), worst cosine %.5f" % (worst, name_w, worst_cos))
    assert worst < 0.04 and worst_cos > 0.999, (worst, name_w, worst_cos)        # measured 0.023 / 0.99973
    sd_g, sd_w = net.state_dict(), ref.state_dict()
    for k in sd_w:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert torch.allclose(sd_g[k], sd_w[k], rtol=3e-2, atol=3e-3), k


def test_full_size_detector_trunk_and_fpn_against_the_oracle_run_on_the_gpu(dev):
    """The frozen ResNet-50 + FPN trunk of configs[1]'s three detector passes at FULL size (24 images of 512x640 -> the transform's
    24 x 300 x 300) against the oracle's module tree (oracle/detection.py BackboneWithFPN: nn.Conv2d + FrozenBatchNorm2d + max_pool2d +
    nearest top-down, ATen fp32 operators) evaluated on the GPU with the same folded fp16 weights and the product's rounding schedule:
    the five pyramid levels."""
    from oracle import detection as od
    from oracle import unet as ou
    from test_detector_gpu import fold_oracle_
    from hallucidet_amd.models.detector import Detector
    torch.manual_seed(11)
    det = Detector(name="fasterrcnn", pretrained=False, n_classes=2, size=300).detector
    with torch.no_grad():
        for mod in det.modules():
            if isinstance(mod, (torch.nn.Conv2d, torch.nn.Linear)) and mod.bias is not None:
                mod.weight.copy_(mod.weight.half().float())
    det = det.to(dev).eval()
    images = torch.rand(24, 3, 512, 640, generator=torch.Generator().manual_seed(12)).to(dev)
    il, _ = det.transform(images, None)
    assert il.tensors.shape[:3] == (24, 300, 300)
    det.backbone.calibrate_(il.tensors)
    oracle = od.FasterRCNN(num_classes=2, size=300)
    oracle.load_state_dict({k: v.cpu() for k, v in det.state_dict().items()})
    fold_oracle_(oracle)
    oracle.eval()
    oracle.set_quant(ou.fp16_round)
    ob = oracle.backbone.to(dev)
    with torch.no_grad():
        f = det.backbone(il.tensors)
        of = ob(il.tensors[..., :3].permute(0, 3, 1, 2).float().contiguous())
    torch.cuda.synchronize()
    assert list(f.keys()) == ["0", "1", "2", "3", "pool"] == list(of.keys())
    for k in f:
        a, b = f[k].permute(0, 3, 1, 2).float(), of[k]
        assert a.shape == b.shape, k
        e = (a - b).abs()
        print("level %s: mean |err| / mean |ref| = %.2e, max |err| / max |ref| = %.2e" % (k, float(e.mean() / b.abs().mean()), float(e.max() / b.abs().max())))
        assert e.mean() < 1.5e-2 * b.abs().mean() + 1e-4 and e.max() < 0.08 * b.abs().max() + 1e-2, (k, float(e.mean()), float(e.max()), float(b.abs().mean()))
