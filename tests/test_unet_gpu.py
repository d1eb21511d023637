"""GPU parity of the hallucination network (forward, backward, BatchNorm bookkeeping) against the CPU oracle.

Two comparisons:
  * against the oracle run with the product's fp16 rounding schedule (oracle.unet.fp16_round): tight -- what is left is
    fp32 accumulation order inside convs / BN sums;
  * against the plain fp32 oracle (the reference semantics): loose -- documents the cost of fp16 storage.
Tolerances are stated per assertion."""
import pytest
import torch

from oracle import unet as ou

pytestmark = pytest.mark.gpu


def _pair(dev, seed=0, name="resnet34"):
    from hallucidet_amd.models.encoder_decoder import EncoderDecoder
    torch.manual_seed(seed)
    net = EncoderDecoder(name=name, encoder_weights=None, in_channels=3, output_channels=3).encoder_decoder
    # conv weights representable in fp16: the fp32 master -> fp16 GEMM-layout repack is then exact on both sides
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.Conv2d):
                m.weight.copy_(m.weight.half().float())
    ref = ou.Unet(classes=3, layers=(2, 2, 2, 2) if name == "resnet18" else (3, 4, 6, 3), block="bottleneck" if name == "resnet50" else "basic")
    ref.load_state_dict(net.state_dict())   # identical key layout
    return net.to(dev), ref


def test_state_dict_layout_matches_oracle(dev):
    net, ref = _pair(dev)
    assert list(net.state_dict().keys()) == list(ref.state_dict().keys())
    assert sum(p.numel() for p in net.parameters()) == 24436659


def test_shape_check(dev):
    net, _ = _pair(dev)
    with pytest.raises(RuntimeError, match="Wrong input shape height=500, width=640"):
        net(torch.zeros(1, 3, 500, 640, device=dev))
    with pytest.raises(RuntimeError, match="no CPU path"):
        net.eval()(torch.zeros(1, 3, 64, 64))


def test_forward_eval_mode(dev):
    net, ref = _pair(dev, 1)
    net.eval(); ref.eval()
    x = torch.rand(2, 3, 64, 96)
    with torch.no_grad():
        got = net(x.to(dev)).cpu()
        want_q = ref(x, q=ou.fp16_round)
        want = ref(x)
    assert got.shape == (2, 3, 64, 96) and got.dtype == torch.float32
    # same rounding schedule: what is left is fp32 summation order, which flips an fp16 rounding now and then and is
    # carried through ~45 conv layers of a randomly initialised (un-normalised in eval mode) network
    e = (got - want_q).abs()
    assert e.mean() < 1e-3 and e.max() < 2e-2, (float(e.mean()), float(e.max()))
    e32 = (got - want).abs()      # fp16 storage vs the fp32 reference semantics; output is a sigmoid in (0,1)
    assert e32.mean() < 2e-3 and e32.max() < 3e-2, (float(e32.mean()), float(e32.max()))


def _train_pair_run(dev, seed, N, H, W, S):
    """Runs HIP forward+backward, then the oracle with the SAME ReLU decisions (masks taken from the HIP activations):
    a ulp-level forward difference next to zero otherwise flips a ReLU and, through ~45 layers, decorrelates the
    gradients of this randomly initialised network by ~30 % (measured: the fp16-rounded oracle differs from the fp32
    oracle by 0.37 rel-L2 on encoder gradients), which would hide real routing bugs."""
    net, ref = _pair(dev, seed)
    net.train(); ref.train()
    x = torch.rand(N, 3, H, W)
    gout = torch.randn(N, 3, H, W) * 1e-2
    net.runner.grad_scale = S
    out = net(x.to(dev))
    from _pins import unet_decisions
    masks, uvalues = unet_decisions(net.runner, device="cpu")
    (out * (gout.to(dev) * S)).sum().backward()
    torch.cuda.synchronize()
    uctx = ou.Ctx(ou.fp16_round, masks, uvalues)
    wq = ref(x, q=uctx)
    from _pins import assert_borrowed_decisions_are_noise
    assert_borrowed_decisions_are_noise(uctx, "U-Net")
    (wq * gout).sum().backward()
    return net, ref, out.detach().cpu(), wq.detach()


def test_forward_backward_train_mode(dev):
    net, ref, out, wq = _train_pair_run(dev, 2, 2, 64, 96, 1024.0)
    e = (out - wq).abs()
    assert e.mean() < 4e-3 and e.max() < 4e-2, (float(e.mean()), float(e.max()))
    # running statistics updated once, with momentum 0.1
    sd_g, sd_w = net.state_dict(), ref.state_dict()
    for k in sd_w:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert torch.allclose(sd_g[k].cpu(), sd_w[k], rtol=3e-2, atol=3e-3), k
        if k.endswith("num_batches_tracked"):
            assert int(sd_g[k]) == int(sd_w[k]) == 1
    # parameter gradients (fp16 gradient storage, loss scale 1024 removed in-kernel): rel-L2 per tensor
    worst = 0.0
    for (n, p), (_, pw) in zip(net.named_parameters(), ref.named_parameters()):
        g, w = p.grad.cpu(), pw.grad
        rel = float((g - w).norm() / (w.norm() + 1e-12))
        worst = max(worst, rel)
        assert rel < 0.03, "%s rel-L2 %.4f" % (n, rel)         # measured worst: 0.017
        cos = float(torch.nn.functional.cosine_similarity(g.flatten(), w.flatten(), dim=0))
        assert cos > 0.998, "%s cosine %.5f" % (n, cos)
    print("worst per-tensor rel-L2 gradient error: %.4f" % worst)


def test_loss_scale_is_removed_from_parameter_gradients(dev):
    a = _train_pair_run(dev, 4, 1, 64, 64, 1.0)[0]
    b = _train_pair_run(dev, 4, 1, 64, 64, 4096.0)[0]
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        rel = float((p.grad - q.grad).norm() / (q.grad.norm() + 1e-12))
        assert rel < 0.05, (n, rel)


def test_train_step_updates_parameters(dev):
    from hallucidet_amd.optim import FusedAdam
    net, ref, _, _ = _train_pair_run(dev, 3, 2, 64, 64, 256.0)
    sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
    opt = FusedAdam(net, lr=1e-4, clip_value=0.5)
    opt.step()
    opt_w = torch.optim.Adam(ref.parameters(), lr=1e-4)
    torch.nn.utils.clip_grad_value_(ref.parameters(), 0.5)
    opt_w.step()
    # the first Adam step moves every weight by ~lr*g/(|g|+eps): compare the update itself
    agree, total = 0, 0
    for (n, p), (_, pw) in zip(net.named_parameters(), ref.named_parameters()):
        dg = p.detach().cpu() - sd0[n]
        dw = pw.detach() - sd0[n]
        big = pw.grad.abs() > 1e-6
        agree += int(((dg[big] - dw[big]).abs() < 2e-5).sum())
        total += int(big.sum())
    assert total > 1_000_000 and agree / total > 0.99, (agree, total)


def test_graph_replay_equals_eager(dev):
    """hipGraph replay of the forward/backward schedule gives bit-identical outputs, gradients and BN buffers."""
    from hallucidet_amd.models.encoder_decoder import EncoderDecoder
    outs = []
    for graphs in (False, True):
        torch.manual_seed(7)
        net = EncoderDecoder(name="resnet34").encoder_decoder.to(dev).train()
        net.runner.enable_graphs(graphs)
        net.runner.grad_scale = 128.0
        x = torch.rand(2, 3, 64, 64, device=dev)
        g = torch.randn(2, 3, 64, 64, device=dev)
        for it in range(3):                     # 3 steps: capture + replays, BN running stats evolve
            out = net(x + 0.01 * it)
            out.backward(g * 128.0)
        torch.cuda.synchronize()
        outs.append((out.detach().clone(), net.runner.flat_grads.clone(), {k: v.clone() for k, v in net.state_dict().items()}))
    (o0, g0, s0), (o1, g1, s1) = outs
    assert torch.equal(o0, o1)
    assert torch.equal(g0, g1)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
    assert int(s1["encoder.bn1.num_batches_tracked"]) == 3


def test_resnet18_backbone_forward_and_gradients(dev):
    """`--decoder-backbone resnet18` (config.py:147; encoders/resnet.py:127-135): same runner, BasicBlock [2,2,2,2]."""
    net, ref = _pair(dev, 4, name="resnet18")
    assert sum(p.numel() for p in net.parameters()) == sum(p.numel() for p in ref.parameters()) == 14328499
    net.eval(); ref.eval()
    x = torch.rand(2, 3, 64, 96)
    with torch.no_grad():
        e = (net(x.to(dev)).cpu() - ref(x, q=ou.fp16_round)).abs()
    assert e.mean() < 1e-3 and e.max() < 2e-2, (float(e.mean()), float(e.max()))
    net.train(); ref.train()
    gout = torch.randn(2, 3, 64, 96) * 1e-2
    net.runner.grad_scale = 256.0
    out = net(x.to(dev))
    from _pins import unet_decisions
    masks, uvalues = unet_decisions(net.runner, device="cpu")
    (out * (gout.to(dev) * 256.0)).sum().backward()
    uctx = ou.Ctx(ou.fp16_round, masks, uvalues)
    wq = ref(x, q=uctx)
    from _pins import assert_borrowed_decisions_are_noise
    assert_borrowed_decisions_are_noise(uctx, "U-Net")
    (wq * gout).sum().backward()
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        a, b = p.grad.detach().cpu().flatten().double(), q.grad.flatten().double()
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
        assert cos > 0.995, (n, cos)


def test_resnet50_backbone_forward_and_gradients(dev):
    """`--decoder-backbone resnet50` (config.py:147; encoders/resnet.py:136-144): Bottleneck [3,4,6,3] through the same
    runner -- 1x1 convs with train-mode BN, the stride on the 3x3, 2048+1024-channel first decoder block."""
    net, ref = _pair(dev, 6, name="resnet50")
    assert list(net.state_dict().keys()) == list(ref.state_dict().keys())
    assert sum(p.numel() for p in net.parameters()) == sum(p.numel() for p in ref.parameters())
    net.eval(); ref.eval()
    x = torch.rand(2, 3, 64, 96)
    with torch.no_grad():
        e = (net(x.to(dev)).cpu() - ref(x, q=ou.fp16_round)).abs()
    assert e.mean() < 2e-3 and e.max() < 4e-2, (float(e.mean()), float(e.max()))
    net.train(); ref.train()
    gout = torch.randn(2, 3, 64, 96) * 1e-2
    net.runner.grad_scale = 256.0
    out = net(x.to(dev))
    from _pins import unet_decisions
    masks, uvalues = unet_decisions(net.runner, device="cpu")
    (out * (gout.to(dev) * 256.0)).sum().backward()
    uctx = ou.Ctx(ou.fp16_round, masks, uvalues)
    wq = ref(x, q=uctx)
    from _pins import assert_borrowed_decisions_are_noise
    assert_borrowed_decisions_are_noise(uctx, "U-Net")
    (wq * gout).sum().backward()
    worst = 1.0
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        a, b = p.grad.detach().cpu().flatten().double(), q.grad.flatten().double()
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
        worst = min(worst, cos)
        # 0.995 holds for resnet18/34; the gradient of the first layers crosses ~70 conv+BN units here and collects more
        # fp16 rounding (measured worst: 0.990 at encoder.layer1.0.conv1)
        assert cos > 0.98, (n, cos)


def test_graph_recapture_survives_allocations_from_another_thread(dev):
    """The input pipeline's helper thread (DevicePrefetcher) pins host memory and allocates device memory while the main thread may
    be inside a hipGraph capture -- the backward graph is re-captured whenever the loss scale changes.  Captures use
    capture_error_mode='thread_local', so foreign-thread allocations must neither invalidate the capture nor change its result."""
    import threading
    from hallucidet_amd.models.encoder_decoder import EncoderDecoder
    results = []
    for hammer in (False, True):
        torch.manual_seed(9)
        net = EncoderDecoder(name="resnet34").encoder_decoder.to(dev).train()
        net.runner.enable_graphs(True)
        x = torch.rand(2, 3, 64, 96, device=dev)
        g = torch.randn(2, 3, 64, 96, device=dev) * 1e-2
        stop = threading.Event()
        count = [0]

        def stage():
            side = torch.cuda.Stream(device=dev)
            k = 0
            while not stop.is_set():
                k += 1
                h = torch.empty(3, 64 + 8 * (k % 7), 96, dtype=torch.uint8).pin_memory()      # new shapes: fresh hipHostMalloc
                with torch.cuda.stream(side):
                    d = h.to(dev, non_blocking=True).float()                                   # fresh hipMalloc on a side stream
                side.synchronize()
                del d, h
                count[0] += 1
        th = threading.Thread(target=stage, daemon=True)
        if hammer:
            th.start()
        try:
            for scale in (128.0, 256.0, 64.0, 512.0):          # every change of the scale re-captures the backward graph
                net.runner.grad_scale = scale
                out = net(x)
                out.backward(g * scale)
            torch.cuda.synchronize()
        finally:
            stop.set()
            if hammer:
                th.join(timeout=30)
        if hammer:
            assert count[0] > 0, "the staging thread never ran"
        results.append((out.detach().clone(), net.runner.flat_grads.clone()))
    assert bool(torch.isfinite(results[0][1]).all()) and float(results[0][1].abs().sum()) > 0
    assert torch.equal(results[0][0], results[1][0]) and torch.equal(results[0][1], results[1][1])


def test_consumer_side_batchnorm_step_is_bit_identical_to_materialised_activations(dev):
    """The runner's consumer-side BatchNorm (four decoder units hand their RAW conv output + BatchNorm coefficients to the next
    small-channel convolution; hd_bn_apply is not launched for them) against the same network with every activation materialised:
    output, every parameter gradient and every BatchNorm buffer bit for bit, over two training steps, eager and graph-replayed."""
    from hallucidet_amd.models.encoder_decoder import EncoderDecoder
    for graphs in (False, True):
        outs = []
        for fuse in (True, False):
            torch.manual_seed(11)
            net = EncoderDecoder(name="resnet34").encoder_decoder.to(dev).train()
            net.runner.fuse_bn = fuse
            net.runner.enable_graphs(graphs)
            net.runner.grad_scale = 256.0
            assert net.runner._raw_units == {"decoder.blocks.3.conv1", "decoder.blocks.3.conv2", "decoder.blocks.4.conv1", "decoder.blocks.4.conv2"}
            x = torch.rand(2, 3, 64, 96, device=dev)
            g = torch.randn(2, 3, 64, 96, device=dev) * 1e-2
            for it in range(2):
                out = net(x + 0.01 * it)
                raw = [k for k, v in net.runner.saved["rec"].items() if not torch.is_tensor(v["z"])]
                assert (len(raw) == 4) == fuse and (len(raw) == 0) == (not fuse)
                out.backward(g * 256.0)
            torch.cuda.synchronize()
            outs.append((out.detach().clone(), net.runner.flat_grads.clone(), {k: v.clone() for k, v in net.state_dict().items()}))
            net.eval()
            with torch.no_grad():
                outs[-1] += (net(x).clone(),)             # eval mode (running statistics) takes the same route
        (o0, g0, s0, e0), (o1, g1, s1, e1) = outs
        assert torch.equal(o0, o1) and torch.equal(g0, g1) and torch.equal(e0, e1)
        assert bool(torch.isfinite(g0).all()) and float(g0.abs().sum()) > 0
        for k in s0:
            assert torch.equal(s0[k], s1[k]), k


def test_one_to_three_channel_repeat_is_a_view_and_bit_identical(dev):
    """SURVEY K1 (src/utils/utils.py:52-53, train_hallucidet.py:171): `expand_one_channel_to_output_channels` hands the U-Net and the
    detector transform a stride-0 channel view of the single-channel IR batch instead of a materialised `repeat`; the layout kernel
    reads the plane three times.  Outputs, gradients and the detector's ImageList must equal the materialised form bit for bit,
    eager and graph-replayed."""
    from hallucidet_amd.models.encoder_decoder import EncoderDecoder
    from hallucidet_amd.models.detector import Detector
    from hallucidet_amd.utils.utils import Utils
    ir = torch.rand(2, 1, 64, 96, device=dev)
    view = Utils.expand_one_channel_to_output_channels(ir, 3)
    full = ir.repeat(1, 3, 1, 1)
    assert view.shape == full.shape and view.stride(1) == 0 and view.data_ptr() == ir.data_ptr() and torch.equal(view, full)
    assert torch.equal(Utils.expand_one_channel_to_output_channels(full, 1), full)           # multi-channel input: plain repeat
    for graphs in (False, True):
        outs = []
        for x in (view, full):
            torch.manual_seed(13)
            net = EncoderDecoder(name="resnet34").encoder_decoder.to(dev).train()
            net.runner.enable_graphs(graphs)
            net.runner.grad_scale = 64.0
            g = torch.randn(2, 3, 64, 96, device=dev) * 1e-2
            for it in range(2):
                out = net(x)
                out.backward(g * 64.0)
            torch.cuda.synchronize()
            outs.append((out.detach().clone(), net.runner.flat_grads.clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    det = Detector(name="fasterrcnn", pretrained=False, n_classes=2, size=300).detector.to(dev).eval()
    a, _ = det.transform.forward_batches([full, view], None)
    assert torch.equal(a.tensors[:2], a.tensors[2:]) and a.tensors.shape == (4, 300, 300, 8)


def test_backward_sums_from_epilogues_equal_the_separate_reductions(dev):
    """Round 4: the BatchNorm backward sums of 25 units leave with the data-gradient kernels (runner.fuse_bwd_sums).  Same parameter
    gradients as with the separate hd_bn_bwd_reduce passes up to fp32 summation order."""
    net, _ = _pair(dev, seed=5)
    net.train()
    x = torch.rand(4, 3, 256, 320, device=dev)
    g = torch.randn(4, 3, 256, 320, generator=torch.Generator().manual_seed(6)).to(dev) * 1e-3
    r = net.runner
    r.enable_graphs(False)
    grads, named, bufs = [], [], {k: v.clone() for k, v in net.state_dict().items()}
    for on in (False, True):
        net.load_state_dict(bufs)                    # same BatchNorm running statistics for both runs
        r.fuse_bwd_sums = on
        r.grad_scale = 256.0
        for p_ in net.parameters():
            p_.grad = None
        out = net(x)
        out.backward(g * 256.0)
        grads.append(r.flat_grads.clone())
        named.append({n_: p_.grad.detach().clone().double() for n_, p_ in net.named_parameters()})
    a, b = grads[0].double(), grads[1].double()
    assert float(a.abs().max()) > 0 and torch.isfinite(b).all()
    rel = float((a - b).norm() / a.norm())
    print("whole gradient arena, fused vs separate BatchNorm backward sums: rel-L2 %.2e" % rel)
    # Not 1e-6: with fp16 storage ANY re-ordering of an fp32 sum grows to the fp16 ulp within a few layers of the backward chain -- a
    # 1e-7 change of a BatchNorm sum flips the fp16 rounding of ~1e-4 of the next gradient tensor's elements (1e-5 in L2), the next
    # layer flips ~1e-2 of its elements ... the fixed point is ~1e-3.  Measured along the chain: 1.4e-7 at the first fused unit
    # (decoder block 1), 1e-5, 7e-5, 4e-4 one, two, three units later, 1.0 - 2.5e-3 from layer3 down to the stem.
    assert rel <= 5e-3
    # the first unit of the backward chain whose sums come from an epilogue: decoder block 2's conv1 (its gradient is written by the
    # 64 -> 64 kernel); everything before it in the chain is bit-identical, the unit itself differs by fp32 summation order only
    for n_ in ("decoder.blocks.4.conv1.1.bias", "decoder.blocks.3.conv2.1.bias", "decoder.blocks.2.conv2.1.bias"):
        assert torch.equal(named[0][n_], named[1][n_]), n_
    x0, x1 = named[0]["decoder.blocks.2.conv1.1.bias"], named[1]["decoder.blocks.2.conv1.1.bias"]
    d1 = float((x0 - x1).norm() / x0.norm())
    print("first fused unit (decoder block 2, conv1): dbeta rel-L2 %.2e" % d1)
    assert d1 <= 1e-5


def test_deferred_multi_layer_weight_gradients_equal_the_per_layer_launches(dev):
    """Round 5: the 8-wave weight gradients of a backward segment run as ONE multi-layer grid at its end (hd_wgrad_multi; one pixel split
    for the deep stages, written as the OIHW gradient by the kernel) instead of one grid per layer behind that layer's data gradient.
    The data-gradient chain is untouched: every BatchNorm gradient and every non-deferred weight gradient is bit-identical; a deferred
    layer's weight gradient differs by the grouping of its fp32 pixel sum only.  With and without an exchange hook (per-segment launches
    vs two groups) the gradients are bit-identical (splits are planned before the pass)."""
    import hallucidet_amd.segmentation_models.unet as unet_mod
    net, _ = _pair(dev, seed=7)
    net.train()
    x = torch.rand(4, 3, 256, 320, device=dev)
    g = torch.randn(4, 3, 256, 320, generator=torch.Generator().manual_seed(8)).to(dev) * 1e-3
    r = net.runner
    r.enable_graphs(False)
    bufs = {k: v.clone() for k, v in net.state_dict().items()}
    was = unet_mod._WGRAD_DEFER
    named, hooks_seen = {}, []
    try:
        for mode in ("per_layer", "deferred", "deferred_hook"):
            net.load_state_dict(bufs)
            unet_mod._WGRAD_DEFER = mode != "per_layer"
            r.bucket_hook = (lambda lo, hi: hooks_seen.append((lo, hi))) if mode == "deferred_hook" else None
            r.grad_scale = 256.0
            for p_ in net.parameters():
                p_.grad = None
            net(x).backward(g * 256.0)
            torch.cuda.synchronize()
            named[mode] = {n_: p_.grad.detach().clone() for n_, p_ in net.named_parameters()}
    finally:
        unet_mod._WGRAD_DEFER = was
        r.bucket_hook = None
    assert hooks_seen == r.bucket_ranges() and len(hooks_seen) in (2, 5)
    deferred = {n_ for n_ in named["deferred"] if n_.endswith(".weight") and named["deferred"][n_].dim() == 4 and
                not torch.equal(named["deferred"][n_], named["per_layer"][n_])}
    print("%d conv weight gradients differ between the schedules" % len(deferred))
    assert 20 <= len(deferred) <= 40                      # resnet34: 30 encoder + 8 decoder 3x3 layers with >= 64 channels on both sides
    worst = 0.0
    for n_, a in named["per_layer"].items():
        b = named["deferred"][n_]
        assert torch.equal(named["deferred_hook"][n_], b), n_                 # launch grouping does not change a bit
        if n_ in deferred:
            rel = float((a.double() - b.double()).norm() / a.double().norm())
            worst = max(worst, rel)
            assert rel <= 2e-6, (n_, rel)
        else:
            assert torch.equal(a, b), n_
    print("deferred vs per-layer weight gradients: worst rel-L2 %.2e" % worst)
