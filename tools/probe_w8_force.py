import os, sys
sys.path.insert(0, "/root/repo")
import torch
from hallucidet_amd import ops, _abi
lib = _abi.load()
dev = "cuda"
SHAPES = [(8, 19, 19, 256, 256), (24, 19, 19, 256, 256), (8, 10, 10, 512, 512), (24, 10, 10, 512, 512), (8, 10, 10, 256, 256), (24, 10, 10, 256, 256),
          (24, 5, 5, 256, 256), (8, 38, 38, 128, 128), (24, 38, 38, 128, 128), (24, 38, 38, 256, 256), (8, 38, 38, 256, 256), (24, 75, 75, 64, 64)]
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
gen = torch.Generator(device="cuda").manual_seed(0)
for N, H, W, Cin, Cout in SHAPES:
    x = (torch.randn(N, H, W, Cin, device=dev, generator=gen) * 0.5).half()
    w = (torch.randn(Cout, 9 * Cin, device=dev, generator=gen) / (9 * Cin) ** 0.5).half()
    mask = (torch.randn(N, H, W, Cout, device=dev, generator=gen) > 0).half()
    y = torch.empty(N, H, W, Cout, device=dev, dtype=torch.float16)
    line = []
    ref = None
    for cfg in (-1, 10, 11, 12, 13):
        lib.hd_conv_tune_w8(cfg, 0)
        us = timed(lambda: ops.conv2d(x, w, 3, 3, pad=1, mask=mask, out=y))
        if ref is None: ref = y.clone()
        line.append("%s %6.1f%s" % ("auto" if cfg < 0 else "c%d" % (cfg - 10), us, "" if torch.allclose(y.float(), ref.float(), atol=2e-2, rtol=2e-2) else "!"))
    lib.hd_conv_tune_w8(-1, 0)
    print("x%dx%dx%dx%d -> %d | %s" % (N, H, W, Cin, Cout, " | ".join(line)), flush=True)
