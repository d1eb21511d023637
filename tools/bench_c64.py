"""A/B of the register-resident-weights 64 -> 64 channel 3x3 kernel (conv3x3_c64.hip) against the implicit-GEMM family on the
shapes of one training step (HIP events on the launch stream; the igemm leg is forced through hd_conv_tune_override)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import ops, _abi

lib = _abi.load()
dev = torch.device("cuda:0")
SHAPES = [
    # name, N, H, W, stats, bias, res, mask, act
    ("unet layer1 fwd (stats)", 8, 128, 160, True, False, False, False, 0),
    ("unet layer1 dgrad", 8, 128, 160, False, False, False, False, 0),
    ("unet layer1 dgrad + res", 8, 128, 160, False, False, True, False, 0),
    ("det layer1 3x3 fwd N=24", 24, 75, 75, False, True, False, False, 1),
    ("det layer1 3x3 dgrad N=8", 8, 75, 75, False, False, False, True, 0),
]


def run(N, H, W, stats, bias, res, mask, act, it=30):
    g = torch.Generator(device="cpu").manual_seed(1)
    x = (torch.randn(N, H, W, 64, generator=g) * 0.5).half().to(dev)
    w = (torch.randn(64, 576, generator=g) / 24.0).half().to(dev)
    b = torch.randn(64, generator=g).to(dev) if bias else None
    r = (torch.randn(N, H, W, 64, generator=g) * 0.5).half().to(dev) if res else None
    m = (torch.rand(N, H, W, 64, generator=g) > 0.4).half().to(dev) if mask else None
    y = torch.empty(N, H, W, 64, device=dev, dtype=torch.float16)
    kw = dict(bias=b, res=r, mask=m, pad=1, act=act, out=y, want_stats=stats)
    for _ in range(3):
        o = ops.conv2d(x, w, 3, 3, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        o = ops.conv2d(x, w, 3, 3, **kw)
    e1.record(); torch.cuda.synchronize()
    out = o[0] if stats else o
    return e0.elapsed_time(e1) / it * 1e3, out.clone(), (o[1].sum(0).clone() if stats else None)


for (name, N, H, W, stats, bias, res, mask, act) in SHAPES:
    t_new, y_new, s_new = run(N, H, W, stats, bias, res, mask, act)
    lib.hd_conv_tune_override(128, 64, 64, 0)
    t_old, y_old, s_old = run(N, H, W, stats, bias, res, mask, act)
    lib.hd_conv_tune_override(-1, -1, -1, -1)
    fl = 2.0 * N * H * W * 64 * 576
    err = float((y_new.float() - y_old.float()).abs().max())
    serr = float((s_new - s_old).abs().max() / s_old.abs().max()) if stats else 0.0
    print("%-28s c64 %6.1f us (%5.0f TFLOP/s)   igemm %6.1f us   max|diff| %.3g  stats rel %.2g" % (name, t_new, fl / t_new / 1e6, t_old, err, serr), flush=True)
