"""Run ONE conv shape repeatedly with a forced kernel choice (profiling target for rocprofv3 --pmc / --kernel-trace).
    W8=cfg[,slices] (or W8=-2 for the 4-wave family)  SHAPE=N,H,W,C1,C2,Cout,K,stride,pad,up1  REPS=n  python tools/one_conv.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import ops, _abi

lib = _abi.load()
dev = torch.device("cuda:0")
N, H, W, C1, C2, Cout, K, s, p, up = [int(v) for v in os.environ.get("SHAPE", "24,75,75,256,0,256,3,1,1,0").split(",")]
w8 = [int(v) for v in os.environ.get("W8", "0").split(",")]
reps = int(os.environ.get("REPS", "20"))
x = (torch.randn(N, H, W, C1, device=dev) * 0.5).half()
Hin, Win = (2 * H, 2 * W) if up else (H, W)
x2 = (torch.randn(N, Hin, Win, C2, device=dev) * 0.5).half() if C2 else None
w = (torch.randn(Cout, K * K * (C1 + C2), device=dev) * 0.05).half()
Ho, Wo = ops.conv_out_size(Hin, K, s, p), ops.conv_out_size(Win, K, s, p)
y = torch.empty(N, Ho, Wo, Cout, device=dev, dtype=torch.float16)
lib.hd_conv_tune_w8(w8[0], w8[1] if len(w8) > 1 else 0)
for _ in range(3):
    ops.conv2d(x, w, K, K, x2=x2, stride=s, pad=p, up1=bool(up), out=y)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.conv2d(x, w, K, K, x2=x2, stride=s, pad=p, up1=bool(up), out=y)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * N * Ho * Wo * Cout * K * K * (C1 + C2)
print("W8=%s shape=%s  %.1f us  %.0f TFLOP/s" % (w8, (N, H, W, C1, C2, Cout, K, s, p, up), ms * 1e3, fl / ms / 1e9))
