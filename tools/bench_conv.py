"""Micro-benchmark of the implicit-GEMM conv kernel on the layer shapes of the hot path (HIP events, one stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import ops

dev = torch.device("cuda:0")
SHAPES = [
    # name, N, H, W, C1, C2, Cout, K, stride, pad, up1
    ("unet.stem 7x7s2", 8, 512, 640, 8, 0, 64, 7, 2, 3, False),
    ("unet.layer1 64->64", 8, 128, 160, 64, 0, 64, 3, 1, 1, False),
    ("unet.layer2 128->128", 8, 64, 80, 128, 0, 128, 3, 1, 1, False),
    ("unet.layer3 256->256", 8, 32, 40, 256, 0, 256, 3, 1, 1, False),
    ("unet.layer4 512->512", 8, 16, 20, 512, 0, 512, 3, 1, 1, False),
    ("unet.dec0.conv1 768->256", 8, 16, 20, 512, 256, 256, 3, 1, 1, True),
    ("unet.dec1.conv1 384->128", 8, 32, 40, 256, 128, 128, 3, 1, 1, True),
    ("unet.dec2.conv1 192->64", 8, 64, 80, 128, 64, 64, 3, 1, 1, True),
    ("unet.dec3.conv1 128->32", 8, 128, 160, 64, 64, 32, 3, 1, 1, True),
    ("unet.dec3.conv2 32->32", 8, 256, 320, 32, 0, 32, 3, 1, 1, False),
    ("unet.dec4.conv1 32->16", 8, 256, 320, 32, 0, 16, 3, 1, 1, True),
    ("unet.dec4.conv2 16->16", 8, 512, 640, 16, 0, 16, 3, 1, 1, False),
    ("det.layer1 1x1 256->64", 8, 75, 75, 256, 0, 64, 1, 1, 0, False),
    ("det.layer1 3x3 64->64", 8, 75, 75, 64, 0, 64, 3, 1, 1, False),
    ("det.layer1 1x1 64->256", 8, 75, 75, 64, 0, 256, 1, 1, 0, False),
    ("det.layer2 3x3 128", 8, 38, 38, 128, 0, 128, 3, 1, 1, False),
    ("det.layer3 3x3 256", 8, 19, 19, 256, 0, 256, 3, 1, 1, False),
    ("det.layer3 1x1 1024->256", 8, 19, 19, 1024, 0, 256, 1, 1, 0, False),
    ("det.layer4 3x3 512", 8, 10, 10, 512, 0, 512, 3, 1, 1, False),
    ("det.fpn 3x3 256 @75", 8, 75, 75, 256, 0, 256, 3, 1, 1, False),
    ("det.fc6 12544->1024", 4096, 7, 7, 256, 0, 1024, 7, 1, 0, False),
    ("det.fc7 1024->1024", 4096, 1, 1, 1024, 0, 1024, 1, 1, 0, False),
]
only = os.environ.get("ONLY")
tot_t = tot_f = 0.0
for (name, N, H, W, C1, C2, Cout, K, s, p, up) in SHAPES:
    if only and only not in name:
        continue
    x = (torch.randn(N, H, W, C1, device=dev) * 0.5).half()
    Hin, Win = (2 * H, 2 * W) if up else (H, W)
    x2 = (torch.randn(N, Hin, Win, C2, device=dev) * 0.5).half() if C2 else None
    w = (torch.randn(Cout, K * K * (C1 + C2), device=dev) * 0.05).half()
    Ho, Wo = ops.conv_out_size(Hin, K, s, p), ops.conv_out_size(Win, K, s, p)
    y = torch.empty(N, Ho, Wo, Cout, device=dev, dtype=torch.float16)
    for _ in range(3):
        ops.conv2d(x, w, K, K, x2=x2, stride=s, pad=p, up1=up, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    e0.record()
    for _ in range(it):
        ops.conv2d(x, w, K, K, x2=x2, stride=s, pad=p, up1=up, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    fl = 2.0 * N * Ho * Wo * Cout * K * K * (C1 + C2)
    byt = 2.0 * (x.numel() + (x2.numel() if C2 else 0) + y.numel() + w.numel())
    tot_t += ms; tot_f += fl
    print("%-28s M=%7d K=%5d N=%4d  %8.1f us  %7.1f TFLOP/s  %6.2f TB/s(min bytes)" % (name, N * Ho * Wo, K * K * (C1 + C2), Cout, ms * 1e3, fl / ms / 1e9, byt / ms / 1e9))
print("sum %.2f ms, %.1f TFLOP/s aggregate" % (tot_t, tot_f / tot_t / 1e9))
