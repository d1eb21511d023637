"""Device time of the BatchNorm finalize / backward-coefficient launches against the floor of a dependent one-block launch
(hd_bn_eval_scale_shift on 64 channels), each 40x back to back inside one hipGraph: how much of their ~5 us is the launch itself."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import ops, _abi
from hallucidet_amd._abi import check, ptr

dev = "cuda"
lib = _abi.load()


def timed(fn, reps=40):
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
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


g64, b64, rm64, rv64 = torch.ones(64, device=dev), torch.zeros(64, device=dev), torch.zeros(64, device=dev), torch.ones(64, device=dev)
print("floor: hd_bn_eval_scale_shift on 64 channels (one block)  %.2f us" % timed(lambda: ops.bn_eval_scale_shift(g64, b64, rm64, rv64, 1e-5)))
for rows, C in [(32, 512), (64, 256), (128, 128), (256, 64), (1024, 64), (1280, 64), (2048, 32), (4096, 16)]:
    part = torch.randn(rows, 2 * C, device=dev).abs()
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    t_fin = timed(lambda: ops.bn_finalize(part, 1e5, gamma, beta, rm, rv, 0.1, 1e-5))
    print("rows %5d C %4d: finalize %.2f us" % (rows, C, t_fin))
