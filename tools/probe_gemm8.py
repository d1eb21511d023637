"""The plain-GEMM problems of a step (box head fc6 / fc7, fc6's data gradient, the wide 1x1 convolutions) on the 4-wave implicit-GEMM
family (mode 0) and on the large-tile 8-wave GEMM (gemm_w8.hip: 256 x 128 tiles on 8 waves, 128 x 128 on 4): warm graph-replayed duration,
bit-equality of the two paths, error against ATen's fp32 matmul.
    python tools/probe_gemm8.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hallucidet_amd import _abi, ops

dev = torch.device("cuda:0")
lib = _abi.load()
SHAPES = [
    # name, M rows (N, H, W), Cin, Cout, K window, bias+relu, mask
    ("fc6 fwd 8192", (8192, 7, 7), 256, 1024, 7, True, False),
    ("fc6 fwd 4096", (4096, 7, 7), 256, 1024, 7, True, False),
    ("fc6 dgrad", (4096, 1, 1), 1024, 12544, 1, False, False),
    ("fc7 fwd 8192", (8192, 1, 1), 1024, 1024, 1, True, False),
    ("fc7 dgrad 4096 mask", (4096, 1, 1), 1024, 1024, 1, False, True),
    ("l4 down 24x19x19", (24, 19, 19), 1024, 2048, 1, False, False),
    ("l3 conv3 24x19x19", (24, 19, 19), 256, 1024, 1, False, False),
    ("l2 conv1 24x38x38", (24, 38, 38), 512, 128, 1, False, False),
    ("l3 conv1 24x38x38", (24, 38, 38), 512, 256, 1, True, False),
    ("fpn lat 24x38x38", (24, 38, 38), 512, 256, 1, False, False),
    ("ragged 5000x1000", (5000, 1, 1), 576, 1000, 1, True, True),
    # bottleneck 1x1 layers of the detector trunk at N = 24 (conv3: + residual + ReLU; conv1: + ReLU) and their N = 8 data gradients (mask)
    ("l1 conv3 24x75x75 res", (24, 75, 75), 64, 256, 1, "res", False),
    ("l2 conv3 24x38x38 res", (24, 38, 38), 128, 512, 1, "res", False),
    ("l3 conv3 24x19x19 res", (24, 19, 19), 256, 1024, 1, "res", False),
    ("l4 conv3 24x10x10 res", (24, 10, 10), 512, 2048, 1, "res", False),
    ("l3 conv1 24x19x19", (24, 19, 19), 1024, 256, 1, True, False),
    ("l4 conv1 24x10x10", (24, 10, 10), 2048, 512, 1, True, False),
    ("l2 conv1 24x75x75", (24, 75, 75), 256, 128, 1, True, False),
    ("l1 conv1 24x75x75", (24, 75, 75), 256, 64, 1, True, False),
    ("l3 dgrad conv3 8x19x19", (8, 19, 19), 1024, 256, 1, False, True),
    ("l3 dgrad conv1 8x19x19 res", (8, 19, 19), 256, 1024, 1, "resmask", True),
    ("l2 dgrad conv1 8x38x38 res", (8, 38, 38), 128, 512, 1, "resmask", True),
    ("fpn lat 24x75x75", (24, 75, 75), 256, 256, 1, False, False),
]


def timed(fn, reps=6):
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
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


gen = torch.Generator(device="cuda").manual_seed(0)
for name, (N, H, W), Cin, Cout, K, bias_relu, use_mask in SHAPES:
    x = (torch.randn(N, H, W, Cin, device=dev, generator=gen) * 0.5).half()
    w = (torch.randn(Cout, K * K * Cin, device=dev, generator=gen) / (K * K * Cin) ** 0.5).half()
    Ho, Wo = (1, 1) if K > 1 else (H, W)
    use_res = isinstance(bias_relu, str)
    bias = torch.randn(Cout, device=dev, generator=gen) if bias_relu in (True, "res") else None
    res = (torch.randn(N, Ho, Wo, Cout, device=dev, generator=gen) * 0.5).half() if use_res else None
    mask = (torch.randn(N, Ho, Wo, Cout, device=dev, generator=gen) > 0).half() if use_mask else None
    kw = dict(bias=bias, mask=mask, res=res, act=1 if bias_relu in (True, "res") else 0)
    outs, line = {}, []
    for mode in (0, 128, 1128):
        lib.hd_gemm_w8_mode(mode)
        y = torch.empty(N, Ho, Wo, Cout, device=dev, dtype=torch.float16)
        us = timed(lambda: ops.conv2d(x, w, K, K, out=y, **kw))
        outs[mode] = y.clone()
        fl = 2.0 * N * Ho * Wo * Cout * K * K * Cin
        line.append("%s %6.1f" % ("igemm" if mode == 0 else "g%d" % mode, us))
    lib.hd_gemm_w8_mode(-1)
    ya = torch.empty(N, Ho, Wo, Cout, device=dev, dtype=torch.float16)
    us_auto = timed(lambda: ops.conv2d(x, w, K, K, out=ya, **kw))
    ref = x.reshape(N * Ho * Wo, -1).float() @ w.float().t()
    if res is not None:
        ref = ref + res.reshape(ref.shape).float()
    if bias is not None:
        ref = ref + bias
    if mask is not None:
        ref = ref * (mask.reshape(ref.shape).float() > 0)
    if kw["act"]:
        ref = torch.relu(ref)
    err = float((outs[1128].reshape(ref.shape).float() - ref).abs().max())
    print("%-26s M=%6d N=%5d K=%5d | %s | auto %6.1f us | bit-equal %s %s %s | max err %.2e" % (
        name, N * Ho * Wo, Cout, K * K * Cin, " | ".join(line), us_auto, bool(torch.equal(outs[0], outs[128])), "-",
        bool(torch.equal(outs[0], outs[1128])), err), flush=True)
