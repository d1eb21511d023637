"""Probe for DESIGN 6.8-2: would a K split help the detector's small-M convolutions?  The full problem as one launch against the same
FLOPs as TWO (FOUR) half-K (quarter-K) problems in one multi-problem grid (hd_conv2d_multi): the halves run on twice as many CUs with
the same tile -- what an intra-block or two-block K split would buy BEFORE its reduction cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import ops

dev = "cuda"
def t_of(fn, reps=10):
    """graph-replayed (the python call of a multi-problem launch is ~20 us of host time, more than the kernels probed here)"""
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


for (N, H, W, Cin, Cout, K) in [(8, 10, 10, 512, 512, 3), (8, 19, 19, 256, 256, 3), (8, 19, 19, 1024, 256, 1), (8, 10, 10, 2048, 512, 1), (24, 10, 10, 512, 512, 3)]:
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, H, W, Cin, generator=g).half().to(dev)
    w = (torch.randn(Cout, K * K * Cin, generator=g) * 0.02).half().to(dev)
    full = t_of(lambda: ops.conv2d(x, w, K, K, pad=K // 2))
    res = [full]
    for parts in (2, 4):
        c = Cin // parts
        xs = [x[..., i * c:(i + 1) * c].contiguous() for i in range(parts)]
        ws = [(torch.randn(Cout, K * K * c, generator=g) * 0.02).half().to(dev) for _ in range(parts)]
        calls = [(xs[i], ws[i], K, K, dict(pad=K // 2)) for i in range(parts)]
        res.append(t_of(lambda: ops.conv2d_multi(calls)))
    tiles = ((N * H * W + 63) // 64) * (Cout // 64)
    print("N %2d %2dx%2d Cin %4d -> %4d k%d (%3d tiles of 64x64): one launch %5.1f us | 2 half-K problems in one grid %5.1f | 4 quarter-K %5.1f" % (
        N, H, W, Cin, Cout, K, tiles, res[0], res[1], res[2]), flush=True)
