"""What would the U-Net's 3x3 weight gradients cost if a stage's L layers ran as ONE 8-wave grid at the end of the stage (each block keeps
its 64 x 64 x 9 accumulators over L x more pixel tiles, 1 / L of the fp32 slab bytes), instead of one grid per layer fused behind that
layer's data gradient?  Emulated with today's kernels: the stage-wide grid has the per-block work of a single layer's problem at batch
8 * L with the same number of blocks, so `wgrad(N = 8 L, nsplit = s)` is timed against L x `wgrad(N = 8, nsplit)`; the data gradient alone
and the fused data + weight gradient grid are timed beside them.
    python tools/probe_wgrad_defer.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hallucidet_amd import ops

dev = torch.device("cuda:0")
STAGES = [  # name, H, W, C, layers of the stage (3x3 / s1 convs C -> C)
    ("layer1 128x160x64", 128, 160, 64, 6),
    ("layer2 64x80x128", 64, 80, 128, 7),
    ("layer3 32x40x256", 32, 40, 256, 11),
    ("layer4 16x20x512", 16, 20, 512, 5),
]


def timed(fn, reps=4):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


gen = torch.Generator(device="cuda").manual_seed(0)
tot_now = tot_def = 0.0
for name, H, W, C, L in STAGES:
    x = (torch.randn(8, H, W, C, device=dev, generator=gen) * 0.5).half()
    dy = (torch.randn(8, H, W, C, device=dev, generator=gen) * 0.5).half()
    wd = (torch.randn(C, 9 * C, device=dev, generator=gen) / (9 * C) ** 0.5).half()
    dwt = torch.zeros(C, C, 3, 3, device=dev)
    xs = (torch.randn(8 * L, H, W, C, device=dev, generator=gen) * 0.5).half()
    dys = (torch.randn(8 * L, H, W, C, device=dev, generator=gen) * 0.5).half()

    slab = ops.wgrad(x, dy, 3, 3, pad=1)
    ns = slab.shape[0]
    t_w = timed(lambda: ops.wgrad(x, dy, 3, 3, pad=1))
    t_r = timed(lambda: ops.wgrad_reduce(slab, dwt, 3, 3, C))
    t_d = timed(lambda: ops.conv2d(dy, wd, 3, 3, pad=1))
    try:
        t_f = timed(lambda: ops.wgrad_dgrad(x, dy, 3, 3, wd, pad=1))
    except Exception as ex:          # the fused launch is not available for this shape
        t_f = float("nan")
    line = []
    best = (1e9, None)
    for s in (ns, max(1, ns // 2), max(1, ns // 4), ns * 2):
        # stage-wide grid: same blocks as one layer's launch (s splits x tiles), L x the pixel tiles per block
        slab_s = ops.wgrad(xs, dys, 3, 3, pad=1, nsplit=s)
        t_ws = timed(lambda: ops.wgrad(xs, dys, 3, 3, pad=1, nsplit=s))
        t_rs = timed(lambda: ops.wgrad_reduce(slab_s, dwt, 3, 3, C))          # one tensor's worth of slabs: x L tensors below
        per_layer = t_ws / L + t_rs * (s / ns) if False else t_ws / L + t_rs / L
        line.append("s=%d: grid %.1f us (%.1f / layer) + reduce %.1f" % (s, t_ws, t_ws / L, t_rs))
        if t_ws / L + t_rs / L < best[0]:
            best = (t_ws / L + t_rs / L, s)
    now = (t_f if t_f == t_f else t_d + t_w) + t_r
    deferred = t_d + best[0]
    tot_now += L * now
    tot_def += L * deferred
    print("%-20s x%2d | now: fused dgrad+wgrad %.1f (dgrad alone %.1f, wgrad alone %.1f, nsplit %d) + reduce %.1f = %.1f us/layer | stage-wide: %s | "
          "dgrad + best stage-wide share = %.1f us/layer" % (name, L, t_f, t_d, t_w, ns, t_r, now, "; ".join(line), deferred), flush=True)
print("sum over the four stages: now %.0f us, dgrad alone + stage-wide weight gradients %.0f us" % (tot_now, tot_def))
