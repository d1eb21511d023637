"""Data gradient + weight gradient of the deep U-Net layers: one fused grid (ops.wgrad_dgrad / hd_conv2d_wgrad) against the two
launches, graph-replayed, at configs[1] sizes.  HD_FUSE_DGRAD_WGRAD=0 in the environment turns the fused call into two launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hallucidet_amd import ops


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
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


for name, N, H, W, C in (("layer2 128 @64x80", 8, 64, 80, 128), ("layer3 256 @32x40", 8, 32, 40, 256), ("layer4 512 @16x20", 8, 16, 20, 512)):
    x = torch.randn(N, H, W, C, device="cuda").half()
    dy = torch.randn(N, H, W, C, device="cuda").half()
    wd = (torch.randn(C, 9 * C, device="cuda") * 0.02).half()
    tw = timed(lambda: ops.wgrad(x, dy, 3, 3, pad=1))
    td = timed(lambda: ops.conv2d(dy, wd, 3, 3, pad=1, cout=C))
    tf = timed(lambda: ops.wgrad_dgrad(x, dy, 3, 3, wd, pad=1, dgrad=dict(pad=1, cout=C)))
    print("%-20s wgrad %6.1f us + dgrad %6.1f us = %6.1f | one call %6.1f us" % (name, tw, td, tw + td, tf))
