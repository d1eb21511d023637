"""End-to-end parity of ONE training step (BASELINE north-star path: U-Net fwd -> 3 detector passes -> weighted loss ->
backward into the U-Net -> clip 0.5 -> Adam) against oracle/step.py (the CPU restatement of train_hallucidet.py:161-240
plus Lightning's optimisation loop), for both detectors (config 1/2: Faster R-CNN, config 4: RetinaNet).

The forward quantities (11-key loss dict) are compared directly.  The update is compared through Adam's first step, which
is -lr * sign(g) wherever |g| >> eps: the sign agreement of the parameter deltas measures the whole backward chain
(detector dgrad, resize bwd, U-Net dgrad/wgrad/BN bwd, loss-scale removal, clipping, fused Adam) in one number.  fp16
activations flip ReLU masks relative to the fp32 oracle (DESIGN.md "fp16 noise"), which bounds that agreement below 1."""
import pytest
import torch

from oracle import detection as od
from oracle import retinanet as orn
from oracle import unet as ou
from oracle.step import OracleTrainer
from test_detector_gpu import fold_oracle_

pytestmark = pytest.mark.gpu


def _pair(dev, detector_name, seed):
    from hallucidet_amd import synthetic
    lit = synthetic.make_module(seed=seed, device=str(dev), precision=16, detector_name=detector_name)
    det = lit.detector
    with torch.no_grad():
        for mod in det.modules():
            if isinstance(mod, (torch.nn.Conv2d, torch.nn.Linear)) and mod.bias is not None:
                mod.weight.copy_(mod.weight.half().float())
        if detector_name == "retinanet":
            det.head.classification_head.cls_logits.bias.fill_(-2.0)
    det.invalidate_packs()
    ounet = ou.Unet(classes=3)
    ounet.load_state_dict({k: v.cpu() for k, v in lit.encoder_decoder.state_dict().items()})
    odet = orn.RetinaNet(num_classes=2, size=300) if detector_name == "retinanet" else od.FasterRCNN(num_classes=2, size=300)
    odet.load_state_dict({k: v.cpu() for k, v in det.state_dict().items()})
    fold_oracle_(odet)
    odet.set_quant(ou.fp16_round)
    tr = OracleTrainer(unet=ounet, detector=odet, lr=lit.lr, clip=0.5)
    tr.unet_q = ou.fp16_round
    with torch.no_grad():       # product stores conv weights in fp16 for the GEMMs
        for m in ounet.modules():
            if isinstance(m, torch.nn.Conv2d):
                m.weight.copy_(m.weight.half().float())
    # yardstick: the SAME oracle weights evaluated without any activation rounding (pure fp32)
    import copy
    tr32 = OracleTrainer(unet=copy.deepcopy(ounet), detector=copy.deepcopy(odet), lr=lit.lr, clip=0.5)
    tr32.det.set_quant(lambda t: t)
    if detector_name == "fasterrcnn":
        for t_, sd in ((tr, 1), (tr32, 1)):       # same sampler draws on both oracle sides
            fn = Draws(sd)
            t_.det.rpn.fg_bg_sampler.randperm_fn = fn
            t_.det.roi_heads.fg_bg_sampler.randperm_fn = fn
        # ... and on the product side: the reference's per-pass call order (RPN sampler, then RoI sampler, image by image;
        # `fused_passes` / the batched three-pass evaluation draw in a different order) with the same seeded permutations
        lit.batch_detector_passes = False
        lit.detector.fused_passes = False
        fn = Draws(1)
        lit.detector.rpn.fg_bg_sampler.randperm_fn = fn
        lit.detector.roi_heads.fg_bg_sampler.randperm_fn = fn
    return lit, tr, tr32


class Draws:
    """torch.randperm from a private seeded generator; logs the population sizes so that a test can tell whether two sides
    drew for the same populations (then the sampled index sets are identical)."""

    def __init__(self, seed):
        self.seed = seed
        self.reset()

    def reset(self):
        self.g = torch.Generator().manual_seed(self.seed)
        self.sizes = []

    def __call__(self, n):
        self.sizes.append(int(n))
        return torch.randperm(n, generator=self.g)


def _to_cpu(batch):
    rgb, trgb, ir, tir = batch
    c = lambda ts: [{k: v.cpu() for k, v in t.items()} for t in ts]
    return rgb.cpu(), c(trgb), ir.cpu(), c(tir)


@pytest.mark.parametrize("detector_name", ["retinanet", "fasterrcnn"])
def test_training_step_matches_oracle(dev, detector_name):
    from hallucidet_amd import synthetic
    import os
    # Faster R-CNN: a seed at which the first pass's RoI populations of product and oracle coincide (fp16 noise moves a proposal or
    # two across the 0.5-IoU line at most seeds: 1 of 7 tried coincide), so that the 3 % bound below is the one that is exercised
    seed = int(os.environ.get("HD_STEP_TEST_SEED", "17" if detector_name == "fasterrcnn" else "5"))
    lit, tr, tr32 = _pair(dev, detector_name, seed=seed)
    batch = synthetic.make_batch(2, 128, 160, seed=seed + 1, device=str(dev))
    names = [n for n, _ in tr.unet.named_parameters()]
    p_before = {n: p.detach().clone() for n, p in tr.unet.named_parameters()}
    tr32.unet.train()
    tr32.train_step(_to_cpu(batch))              # one step from the same start, for the yardstick
    g_before = {k: v.detach().cpu().clone() for k, v in lit.encoder_decoder.state_dict().items()}

    # ---- forward quantities
    lit.encoder_decoder.train()
    tr.unet.train()
    out = lit.forward_step(*batch, 0, step="train")
    assert set(out["loss"]) == {"total", "pixel_rgb", "perceptual_rgb", "pixel_ir", "perceptual_ir", "det_regression",
                                "det_classification", "det_objectness", "det_rpn_box_reg", "det_bbox_ctrness", "det_total"}
    total, olosses, _ = tr.forward_step(*_to_cpu(batch))
    keymap = ({"det_classification": "classification", "det_regression": "bbox_regression"} if detector_name == "retinanet" else
              {"det_classification": "loss_classifier", "det_regression": "loss_box_reg", "det_objectness": "loss_objectness",
               "det_rpn_box_reg": "loss_rpn_box_reg"})
    # RetinaNet has no sampler: every loss is a deterministic function of the features -> 3 %.  Faster R-CNN: both sides draw
    # the SAME seeded permutations in the same call order; the RPN sampler's populations depend on anchors and targets only, so
    # its subsets are identical -> 3 %.  The RoI sampler's populations depend on the proposals, which fp16 noise may reorder at
    # the NMS / top-k margins: where the first pass's populations coincide the subsets are identical and the RoI losses are
    # held to 3 % too, otherwise (different random subsets of ~512 of ~1000 RoIs) to 30 %.
    tol = {k: 0.03 for k in keymap}
    if detector_name == "fasterrcnn":
        ps, os_ = lit.detector.rpn.fg_bg_sampler.randperm_fn.sizes, tr.det.rpn.fg_bg_sampler.randperm_fn.sizes
        n_img = 2
        assert ps[:2 * n_img] == os_[:2 * n_img], "RPN sampler populations must coincide (anchors and targets are identical)"
        same_roi = ps[2 * n_img:4 * n_img] == os_[2 * n_img:4 * n_img]
        print("sampler populations (first pass) product %s oracle %s -> RoI subsets %s" % (ps[:4 * n_img], os_[:4 * n_img],
                                                                                         "identical" if same_roi else "differ"))
        if not same_roi:
            tol["det_classification"] = tol["det_regression"] = 0.3
    for pk, ok_ in keymap.items():
        a, b = float(out["loss"][pk]), 0.1 * float(olosses[ok_])
        assert abs(a - b) < tol[pk] * abs(b) + 2e-3, (pk, a, b, tol[pk])
    if detector_name == "retinanet":
        assert out["loss"]["det_objectness"] == 0.0 and out["loss"]["det_rpn_box_reg"] == 0.0 and out["loss"]["det_bbox_ctrness"] == 0.0
        assert abs(float(out["loss"]["total"]) - float(total)) < 0.03 * abs(float(total)) + 2e-3
    hall = out["output"]["imgs_hallucinated"]
    assert hall.shape == (2, 3, 128, 160) and float(hall.min()) >= 0.0 and float(hall.max()) <= 1.0

    # ---- one optimisation step on both sides (fresh forward: BN running statistics advance once more on both)
    if detector_name == "fasterrcnn":                # re-seed so that both optimisation steps draw what tr32's single step drew
        lit.detector.rpn.fg_bg_sampler.randperm_fn.reset()
        tr.det.rpn.fg_bg_sampler.randperm_fn.reset()
    loss = lit.fit_step(batch)
    tr.train_step(_to_cpu(batch))
    assert torch.isfinite(loss)
    after = {k: v.detach().cpu() for k, v in lit.encoder_decoder.state_dict().items()}
    agree = tot = yard = 0
    moved = 0
    p32 = dict(tr32.unet.named_parameters())
    for n in names:
        d_ref = dict(tr.unet.named_parameters())[n].detach() - p_before[n]
        d_got = after[n].float() - g_before[n].float()
        d_32 = p32[n].detach() - p_before[n]
        assert float(d_got.abs().max()) <= lit.lr * 1.01 + 1e-9, n         # Adam's first step is bounded by lr
        sel = d_ref.abs() > 0.5 * lit.lr                                     # |g| >> eps on the oracle side
        if n.endswith(".bias") and ".bn" not in n and "segmentation_head" not in n:
            continue
        agree += int((torch.sign(d_ref[sel]) == torch.sign(d_got[sel])).sum())
        yard += int((torch.sign(d_ref[sel]) == torch.sign(d_32[sel])).sum())
        tot += int(sel.sum())
        moved += int((d_got != 0).any())
    frac, yfrac = agree / max(tot, 1), yard / max(tot, 1)
    print("%s: Adam-step sign agreement product~oracle(fp16 schedule) %.3f ; yardstick oracle(fp16)~oracle(fp32) %.3f ; "
          "%d coordinates, %d/%d tensors moved" % (detector_name, frac, yfrac, tot, moved, len(names)))
    assert tot > 1e6 and moved > 0.9 * len(names)
    # the product must be about as close to the fp16-schedule oracle as that oracle is to exact fp32 arithmetic
    assert frac > yfrac - 0.06 and frac > 0.6, (frac, yfrac)
    # BN running statistics advanced identically (two train-mode forwards on both sides)
    for k in ("encoder.bn1.running_mean", "decoder.blocks.4.conv2.1.running_var"):
        ref = tr.unet.state_dict()[k]
        assert torch.allclose(after[k].float(), ref, rtol=5e-2, atol=5e-3), k


def test_validation_and_test_hooks_accumulate_map(dev):
    """validation_step / test_step feed the three detection streams into COCO-style mAP accumulators and the epoch-end
    hooks return {'map_rgb','map_hall','map_ir'} -> {map, map_50, map_75} (train_hallucidet.py:213-215, 328-362, 399-427)."""
    from hallucidet_amd import synthetic
    lit = synthetic.make_module(seed=5, device=str(dev), precision=16)
    batch = synthetic.make_batch(2, 128, 160, seed=6, device=str(dev))
    loss, dets = lit.validation_step(batch, 0)
    assert torch.isfinite(loss) and set(dets) == {"hall", "rgb", "ir"} and len(dets["hall"]) == 2
    lit.validation_step(batch, 1)
    out = lit.on_validation_epoch_end()
    assert set(out) == {"map_rgb", "map_hall", "map_ir"}
    for v in out.values():
        assert set(v) == {"map", "map_50", "map_75"} and all(-1.0 <= float(t) <= 1.0 for t in v.values())
    lit.test_step(batch, 0)
    out_t = lit.on_test_epoch_end()
    assert set(out_t) == set(out)
    # accumulators were reset by the epoch-end hook
    assert all(float(t) == -1.0 for t in lit.on_validation_epoch_end()["map_hall"].values())


def test_skip_unused_train_passes_is_opt_in_and_keeps_the_losses(dev):
    """Opt-in flag: the training step without the RGB / IR passes whose results the reference discards (RetinaNet: no sampler,
    so the hallucinated pass's losses are bit-identical with and without them)."""
    from hallucidet_amd import synthetic
    lit = synthetic.make_module(seed=5, device=str(dev), precision=16, detector_name="retinanet")
    assert lit.skip_unused_train_passes is False
    batch = synthetic.make_batch(2, 128, 160, seed=6, device=str(dev))
    lit.encoder_decoder.eval()                       # fixed BN statistics: two forward_step calls see the same network
    with torch.no_grad():
        a = lit.forward_step(*batch, 0, step="train")
        lit.skip_unused_train_passes = True
        b = lit.forward_step(*batch, 0, step="train")
        c = lit.forward_step(*batch, 0, step="val")
    for k in ("det_classification", "det_regression", "total"):
        assert float(a["loss"][k]) == float(b["loss"][k]) == float(c["loss"][k]), k
    assert lit._last_detections["rgb"] is not None and len(lit._last_detections["rgb"]) == 2      # validation still runs all three


def test_overlapped_allreduce_buckets_rccl_world1(dev):
    """The data-parallel exchange on the GPU (RCCL, world size 1 via HD_FORCE_DIST): with the bucket hooks the U-Net backward is
    replayed as five graph segments and each finished arena slice is all-reduced while the next segment runs.  Must give the
    same step, bit for bit, as the un-overlapped exchange, report the arena from its end, and cover it exactly once."""
    import os
    import socket
    import torch.distributed as dist
    from hallucidet_amd import synthetic
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["HD_FORCE_DIST"] = "1"
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device(dev))
    try:
        batch = synthetic.make_batch(2, 128, 160, seed=6, device=str(dev))
        results = []
        for overlap in (False, True):
            lit = synthetic.make_module(seed=5, device=str(dev), precision=16)
            lit.overlap_allreduce = overlap
            torch.manual_seed(0)
            for _ in range(3):                      # capture step + two replays
                loss = lit.fit_step(batch)
            r = lit.encoder_decoder.runner
            results.append((float(loss), r.flat_grads.clone(), r.flat_params.clone(), list(lit.averager.issued), r.bucket_ranges(), r.flat_grads.numel()))
        (l0, g0, p0, issued0, _, n), (l1, g1, p1, issued1, ranges, _) = results
        assert l0 == l1 and torch.equal(g0, g1) and torch.equal(p0, p1)
        assert issued1[:5] == ranges and ranges[0][1] == n and ranges[-1][0] == 0
        assert all(a[0] == b[1] for a, b in zip(ranges, ranges[1:])), "buckets must tile the arena from its end"
        assert len(issued0) == 4 and sorted(issued0)[0][0] == 0 and sorted(issued0)[-1][1] == n      # no hooks: four equal slices after backward
    finally:
        dist.destroy_process_group()
        os.environ.pop("HD_FORCE_DIST", None)
