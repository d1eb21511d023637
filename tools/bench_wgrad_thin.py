"""Times the thin-output weight-gradient launches of the U-Net decoder at configs[1] sizes (batch 8, 512x640), graph-replayed, with a
sweep over the block count.  Usage: python tools/bench_wgrad_thin.py [nsplit ...]; HD_HIP_LIB selects another build for A/Bs."""
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


def main():
    dev = "cuda"
    N = 8
    splits = [int(a) or None for a in sys.argv[1:]] or [None]   # 0: the heuristic of ops.wgrad
    cases = [("dec3.conv1 64up+64 -> 32 @256x320", (N, 128, 160, 64), (N, 256, 320, 64), 32, True),
             ("dec3.conv2 32 -> 32 @256x320", (N, 256, 320, 32), None, 32, False),
             ("dec4.conv1 32up -> 16 @512x640", (N, 256, 320, 32), None, 16, True),
             ("dec4.conv2 16 -> 16 @512x640", (N, 512, 640, 16), None, 16, False)]
    for name, xs, x2s, cout, up in cases:
        x = torch.randn(xs, device=dev).half()
        x2 = torch.randn(x2s, device=dev).half() if x2s else None
        H, W = (xs[1] * 2, xs[2] * 2) if up else (xs[1], xs[2])
        dy = torch.randn(N, H, W, cout, device=dev).half()
        cin = xs[3] + (x2s[3] if x2s else 0)
        dw = torch.empty(cout, cin, 3, 3, device=dev)
        row = []
        for ns in splits:
            t = timed(lambda: ops.wgrad(x, dy, 3, 3, x2=x2, pad=1, up1=up, nsplit=ns))
            slab = ops.wgrad(x, dy, 3, 3, x2=x2, pad=1, up1=up, nsplit=ns)
            t2 = timed(lambda: ops.wgrad_reduce(slab, dw, 3, 3, cin, scale=1.0))
            row.append("ns=%s: %.1f+%.1f" % (slab.shape[0], t, t2))
        hbm = (x.numel() + (x2.numel() if x2 is not None else 0) + dy.numel()) * 2 / 8e12 * 1e6
        print("%-36s hbm %.1f us | %s" % (name, hbm, " | ".join(row)), flush=True)


if __name__ == "__main__":
    main()
