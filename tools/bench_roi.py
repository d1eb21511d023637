"""A/B of the multi-level RoIAlign forward (one block per RoI, hd_roi_align_ml with HD_ROI_ROWS=1) against the one-thread-per-output
form (HD_ROI_ROWS=0) on the RoI population of one training step: 24 images at 300x300, four pyramid levels of 256 channels, 512 RoIs
per image; the two launches of a step (16 images without gradient, 8 with).  The kernel choice is read once per process, so each leg
runs in a child; outputs must be bit-identical.
    python tools/bench_roi.py"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import math
    import torch
    from hallucidet_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    n_img = 24
    sizes = [75, 38, 19, 10]
    feats = [(torch.randn(n_img, s, s, 256, generator=g) * 0.5).half().to(dev) for s in sizes]
    scales = [2.0 ** -round(math.log2(300.0 / s)) for s in sizes]
    per = 512
    # boxes: log-uniform sizes 8..300 px, some partly outside, a few degenerate
    wh = torch.exp(torch.rand(n_img * per, 2, generator=g) * math.log(300.0 / 8.0)) * 8.0
    c = torch.rand(n_img * per, 2, generator=g) * 300.0
    b = torch.cat([c - wh / 2, c + wh / 2], 1).clamp(0, 300)
    b[::97, 2:] = b[::97, :2]
    img = torch.arange(n_img).repeat_interleave(per).float()[:, None]
    rois = torch.cat([img, b], 1).to(dev)
    area = ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])).clamp(min=1e-6)
    lvl = torch.floor(4 + torch.log2(torch.sqrt(area) / 224.0) + 1e-6).clamp(2, 5).to(torch.int32) - 2
    lvl = lvl.to(dev)
    outs = []
    for (lo, hi, name) in ((0, 16 * per, "16 images (8192 RoIs)"), (16 * per, 24 * per, "8 images (4096 RoIs)")):
        r, l = rois[lo:hi].contiguous(), lvl[lo:hi].contiguous()
        for _ in range(3):
            o = ops.roi_align_ml(feats, scales, r, l, 7, 7, 2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            o = ops.roi_align_ml(feats, scales, r, l, 7, 7, 2)
        e1.record(); torch.cuda.synchronize()
        print("   %-24s %7.1f us" % (name, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
        outs.append(o.cpu())
    # backward (gather form) of the 8 images with gradient
    r, l = rois[:8 * per].contiguous(), lvl[:8 * per].contiguous()
    dout = (torch.randn(8 * per, 7, 7, 256, generator=g) * 0.1).half().to(dev)
    shapes = [(8, s_, s_, 256) for s_ in sizes]
    for _ in range(3):
        dfs = ops.roi_align_ml_bwd_gather(dout, r, l, shapes, scales, 2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        dfs = ops.roi_align_ml_bwd_gather(dout, r, l, shapes, scales, 2)
    e1.record(); torch.cuda.synchronize()
    print("   %-24s %7.1f us" % ("backward, 8 images", e0.elapsed_time(e1) / 20 * 1e3), flush=True)
    outs.append(torch.cat([d.flatten() for d in dfs]).cpu())
    torch.save(outs, sys.argv[2])
    raise SystemExit(0)

import torch
res = {}
for mode in ("0", "1"):
    print("HD_ROI_ROWS=%s" % mode, flush=True)
    path = "/tmp/bench_roi_%s.pt" % mode
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", path], env=dict(os.environ, HD_ROI_ROWS=mode), check=True)
    res[mode] = torch.load(path)
for a, b in zip(res["0"], res["1"]):
    d = (a.float() - b.float()).abs()
    ulp = torch.maximum(a.float().abs(), b.float().abs()) * 2.0 ** -10 + 2.0 ** -24
    print("bit-identical: %s  max |diff| %.3g  largest diff in fp16 ulps of the value %.2f  elements that differ %.2e  non-zero share %.3f" %
          (torch.equal(a.view(torch.int16), b.view(torch.int16)), float(d.max()), float((d / ulp).max()), float((d > 0).float().mean()), float((a != 0).float().mean())))
