"""COLD-cache timing of conv launches: in a training step every layer meets its weights (and most of its input) in HBM, not in
L2 / the Infinity Cache, which a back-to-back re-run of one launch hides (the in-step duration of the 512-channel 16x20 layers is
33 us against 25 us re-run warm).  Each timed launch is preceded, inside one hipGraph, by a 512 MiB fill that evicts both caches;
the fill's own time is measured with an identical graph without the launch and subtracted.
    python tools/bench_cold.py            # shipped heuristic vs the 8-wave im2col family with split-K, per shape"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import ops, _abi
lib = _abi.load()
dev = "cuda"
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
REPS = 8


def graph_time(fn):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS):
            flush.fill_(1)
            if fn is not None:
                fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


SHAPES = [  # name, N, H, W, Cin, Cout, k, stride
    ("unet layer4 3x3 512", 8, 16, 20, 512, 512, 3, 1),
    ("det layer3 3x3 256 x24", 24, 19, 19, 256, 256, 3, 1),
    ("det layer3 3x3 256 x8", 8, 19, 19, 256, 256, 3, 1),
    ("det layer4 3x3 512 x24", 24, 10, 10, 512, 512, 3, 1),
    ("det layer4 3x3 512 x8", 8, 10, 10, 512, 512, 3, 1),
    ("det layer4 1x1 2048->512 x24", 24, 10, 10, 2048, 512, 1, 1),
    ("det layer4 1x1 2048->512 x8", 8, 10, 10, 2048, 512, 1, 1),
    ("det layer3 1x1 1024->256 x24", 24, 19, 19, 1024, 256, 1, 1),
    ("det layer3 1x1 256->1024 x24", 24, 19, 19, 256, 1024, 1, 1),
    ("unet layer3 3x3 256", 8, 32, 40, 256, 256, 3, 1),
]
TILES = [(256, 128), (128, 128), (256, 64), (128, 64), (128, 256), (64, 128), (64, 256)]
base = graph_time(None)
print("flush alone: %.1f us" % base)
for name, N, H, W, Cin, Cout, k, s in SHAPES:
    x = torch.randn(N, H, W, Cin, device=dev).half()
    w = (torch.randn(Cout, k * k * Cin, device=dev) * 0.05).half()
    call = lambda: ops.conv2d(x, w, k, k, pad=k // 2, stride=s)
    lib.hd_conv_tune_w8(-1, 0)
    ref = call().float()
    t_cold = graph_time(call) - base
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS):
            call()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    t_warm = e0.elapsed_time(e1) / REPS * 1e3
    fl = 2.0 * N * H * W * Cout * k * k * Cin
    line = "%-30s %5.1f GF  heuristic: cold %5.1f us  warm %5.1f us |" % (name, fl / 1e9, t_cold, t_warm)
    best = (t_cold, "heuristic")
    for cfg in (1, 3, 5, 0):
        for ns in (1, 2, 3, 4, 6, 8):
            lib.hd_conv_tune_w8(cfg, ns)
            try:
                out = call().float()
            except Exception as e:
                continue
            err = float((out - ref).abs().max() / (ref.abs().max() + 1e-6))
            if err > 2e-2:
                line += " [cfg %d x%d WRONG %.2g]" % (cfg, ns, err)
                continue
            t = graph_time(call) - base
            if t < best[0]:
                best = (t, "w8 %dx%d split %d" % (TILES[cfg][0], TILES[cfg][1], ns))
            line += " %dx%d/%d: %5.1f" % (TILES[cfg][0], TILES[cfg][1], ns, t)
    lib.hd_conv_tune_w8(-1, 0)
    print(line)
    print("        best: %.1f us (%s)" % best)
