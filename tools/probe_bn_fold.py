"""BatchNorm of one Conv2dReLU unit with and without the fold (hd_bn_finalize_apply / hd_bn_bwd_apply's in-block coefficients) at the
(rows, C, pixels) combinations of a training step: device time of finalize + apply (forward) and coefficients + apply (backward) as
they run in the step -- dependent launches inside one hipGraph, 20 repetitions.
    python tools/probe_bn_fold.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hallucidet_amd import _abi, ops

dev = "cuda"
lib = _abi.load()


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


# (unit class, rows of the conv epilogue, C, pixels)
CASES = [("layer4 16x20x512", 20, 512, 8 * 16 * 20), ("layer3 32x40x256", 80, 256, 8 * 32 * 40), ("layer2 64x80x128 (256-px tiles)", 160, 128, 8 * 64 * 80),
         ("layer2 64x80x128 (128-px tiles)", 320, 128, 8 * 64 * 80), ("layer1 128x160x64 (c64: 256 blocks)", 256, 64, 8 * 128 * 160),
         ("stem 256x320x64", 256, 64, 8 * 256 * 320), ("bwd reduce rows 512, C 64", 512, 64, 8 * 128 * 160), ("bwd reduce rows 512, C 128", 512, 128, 8 * 64 * 80)]
for name, rows, C, npix in CASES:
    y = torch.randn(npix, C, device=dev).half()
    dz = torch.randn(npix, C, device=dev).half()
    part = torch.rand(rows, 2 * C, device=dev) + 0.5
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    res = {}
    for lim in (0, 1 << 30):
        lib.hd_bn_fold_limit(lim)

        def fwd():
            return ops.bn_finalize_apply(part, float(npix), gamma, beta, rm, rv, 0.1, 1e-5, y, relu=True)
        mean, invstd = fwd()[:2]
        tf = timed(fwd)
        tb = timed(lambda: ops.bn_backward(dz, None, y, mean, invstd, gamma, beta, part=part))
        res[lim] = (tf, tb)
    lib.hd_bn_fold_limit(-1)
    print("%-40s rows*2C %6d floats | fwd finalize+apply %5.1f -> fold %5.1f us | bwd coef+apply %5.1f -> fold %5.1f us | rule folds: %s" % (
        name, rows * 2 * C, res[0][0], res[1 << 30][0], res[0][1], res[1 << 30][1], bool(lib.hd_bn_fold_ok(rows, C))), flush=True)
