"""Device-time of the BatchNorm backward launches per U-Net shape (each launch 20x inside one hipGraph)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import ops, _abi
from hallucidet_amd._abi import check, ptr
dev = "cuda"
lib = _abi.load()


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


SHAPES = [(8 * 16 * 20, 512), (8 * 32 * 40, 256), (8 * 64 * 80, 128), (8 * 128 * 160, 64), (8 * 256 * 320, 32), (8 * 512 * 640, 16), (8 * 256 * 320, 64)]
for npix, C in SHAPES:
    y = torch.randn(npix, C, device=dev).half()
    dz = torch.randn(npix, C, device=dev).half()
    mean = torch.zeros(C, device=dev); invstd = torch.ones(C, device=dev); gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    line = "npix %7d C %3d (%5.1f MB x2):" % (npix, C, npix * C * 2 / 1e6)
    for rows in sorted({max(1, min(512, npix // 64)), max(1, min(2048, npix // 16)), max(1, min(4096, npix // 8)), 256, 1024}):
        part = torch.empty(rows, 2 * C, device=dev)
        t = timed(lambda: check(lib.hd_bn_bwd_reduce(ptr(dz), None, ptr(y), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), ptr(part), rows, npix, C, 1, _abi.current_stream()), "r"))
        line += "  rows %4d: %5.1f us" % (rows, t)
    z = torch.empty_like(y)
    sc = torch.ones(C, device=dev); sh = torch.zeros(C, device=dev)
    t = timed(lambda: ops.bn_apply(y, sc, sh, relu=True))
    line += " | bn_apply %5.1f us" % t
    t = timed(lambda: ops.bn_backward(dz, None, y, mean, invstd, gamma, beta))
    line += " | bn_backward (3 launches) %5.1f us" % t
    print(line)
