"""How fast does the library GEMM (torch -> hipBLASLt / rocBLAS) run the box head's plain GEMMs, next to hd_conv2d?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import ops

def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

dev = "cuda"
for (M, K, N, name) in ((8192, 12544, 1024, "fc6 fwd 8192"), (4096, 12544, 1024, "fc6 fwd 4096"), (4096, 1024, 12544, "fc6 dgrad"),
                        (8192, 1024, 1024, "fc7 fwd"), (4096, 1024, 1024, "fc7 dgrad")):
    a = torch.randn(M, K, device=dev, dtype=torch.float16) * 0.1
    w = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.02
    b = torch.randn(N, device=dev, dtype=torch.float16)
    bf = b.float()
    fl = 2.0 * M * K * N
    t_lin = t(lambda: torch.nn.functional.linear(a, w, b))
    t_act = t(lambda: torch._addmm_activation(b, a, w.t(), use_gelu=False))
    x4 = a.view(M, 1, 1, K)
    t_hd = t(lambda: ops.conv2d(x4, w, 1, 1, bias=bf, act=1))
    y1 = torch._addmm_activation(b, a, w.t(), use_gelu=False)
    y2 = ops.conv2d(x4, w, 1, 1, bias=bf, act=1).view(M, N)
    err = float((y1.float() - y2.float()).abs().max()) / float(y2.float().abs().max())
    print("%-14s M=%5d K=%5d N=%5d | F.linear %6.1f us (%4.0f TF/s) | addmm+relu epilogue %6.1f us (%4.0f TF/s) | hd_conv2d %6.1f us (%4.0f TF/s) | rel diff %.1e" % (
        name, M, K, N, t_lin, fl / t_lin / 1e6, t_act, fl / t_act / 1e6, t_hd, fl / t_hd / 1e6, err))
