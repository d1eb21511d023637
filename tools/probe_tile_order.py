"""The detector's small-grid long-K layers (10x10 / 19x19 stages at N = 8 / 24) and the box head's GEMMs one by one: warm
graph-replayed duration, the error against ATen's fp32 convolution, batch invariance -- for the process's HD_CONV_TGROUP setting
(conv_params.h: hd_conv_tile_order; the knob is read once per process, so A/B = two processes on one box).
    HD_CONV_TGROUP=0 python tools/probe_tile_order.py ; HD_CONV_TGROUP=1 python tools/probe_tile_order.py
Round 5 also ran an intra-block K split through this script (4 groups of 4 waves per 64x64 tile, one LDS reduction): N = 8 layers
-18 ... -26 % once the weights were L2-resident, every > 256-tile launch +40 ... +70 % (128 KB of LDS = one block per CU), and the
choice cannot depend on the batch (bit-invariance): net ~ -0.06 ms per step for two more kernel variants -- not kept."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from hallucidet_amd import ops

dev = torch.device("cuda:0")
SHAPES = [
    # N, H, W, Cin, Cout, K, res, mask
    (8, 10, 10, 512, 512, 3, False, True),
    (24, 10, 10, 512, 512, 3, False, False),
    (8, 10, 10, 2048, 512, 1, False, True),
    (24, 10, 10, 2048, 512, 1, False, False),
    (8, 19, 19, 256, 256, 3, False, True),
    (24, 19, 19, 256, 256, 3, False, False),
    (8, 19, 19, 1024, 256, 1, False, True),
    (24, 19, 19, 1024, 256, 1, False, False),
    (8, 10, 10, 256, 256, 3, False, False),
    (24, 10, 10, 256, 256, 3, False, False),
    (24, 5, 5, 256, 256, 3, False, False),
    (8, 19, 19, 256, 1024, 1, True, True),      # short K
    (8, 38, 38, 128, 128, 3, False, True),      # 362 tiles
    (24, 10, 10, 512, 2048, 1, True, False),    # layer4 conv3 (weights 2 MB)
    (8, 10, 10, 2048, 512, 1, False, True),
    (24, 19, 19, 1024, 2048, 1, False, False),  # layer4 downsample at stride 1 geometry (weights 4 MB)
    (4096, 1, 1, 1024, 12544, 1, False, False), # fc6 data gradient (weights 25.7 MB)
    (8192, 7, 7, 256, 1024, 7, False, False),   # fc6 forward
    (4096, 7, 7, 256, 1024, 7, False, False),
    (8192, 1, 1, 1024, 1024, 1, False, False),  # fc7
]


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


tot = 0.0
gen = torch.Generator(device="cuda").manual_seed(0)
for (N, H, W, Cin, Cout, K, use_res, use_mask) in SHAPES:
    x = (torch.randn(N, H, W, Cin, device=dev, generator=gen) * 0.5).half()
    w = (torch.randn(Cout, K * K * Cin, device=dev, generator=gen) / (K * K * Cin) ** 0.5).half()
    res = (torch.randn(N, H, W, Cout, device=dev, generator=gen) * 0.5).half() if use_res else None
    mask = (torch.randn(N, H, W, Cout, device=dev, generator=gen) > 0).half() if use_mask else None
    assert not ((use_res or use_mask) and K == 7)
    Ho = H if (H > 7 or K < 7) else 1
    y = torch.empty(N, Ho, Ho if Ho == 1 else W, Cout, device=dev, dtype=torch.float16)
    pad = K // 2 if H > 7 or K < 7 else 0
    ops.conv2d(x, w, K, K, pad=pad, res=res, mask=mask, out=y)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().reshape(Cout, K, K, Cin).permute(0, 3, 1, 2), padding=pad).permute(0, 2, 3, 1)
    if res is not None:
        ref = ref + res.float()
    if mask is not None:
        ref = ref * (mask.float() > 0)
    err = float((y.float() - ref).abs().max())
    one = torch.empty(1, y.shape[1], y.shape[2], Cout, device=dev, dtype=torch.float16)
    ops.conv2d(x[3:4].contiguous(), w, K, K, pad=pad, res=None if res is None else res[3:4].contiguous(),
               mask=None if mask is None else mask[3:4].contiguous(), out=one)
    inv = bool(torch.equal(one, y[3:4]))
    us = timed(lambda: ops.conv2d(x, w, K, K, pad=pad, res=res, mask=mask, out=y))
    tot += us
    fl = 2.0 * N * H * W * Cout * K * K * Cin
    print("x%dx%dx%dx%d k%d -> %d%s%s  %7.1f us  %6.1f TFLOP/s  max err %.2e  batch-invariant %s" % (
        N, H, W, Cin, K, Cout, " res" if use_res else "", " mask" if use_mask else "", us, fl / us / 1e6, err, inv), flush=True)
print("HD_CONV_TGROUP=%s: sum %.1f us" % (os.environ.get("HD_CONV_TGROUP", "(default)"), tot))
