import sys, os
sys.path.insert(0, "/root/repo")
import torch
from hallucidet_amd import ops
dev = "cuda"
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for (N, H, W, C) in [(8, 512, 640, 16), (8, 256, 320, 32), (8, 128, 160, 64), (8, 64, 80, 128), (8, 32, 40, 256), (8, 16, 20, 512)]:
    y = torch.randn(N, H, W, C, device=dev).half(); dz = torch.randn_like(y)
    mean = torch.zeros(C, device=dev); invstd = torch.ones(C, device=dev); g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    npix = N * H * W
    out = []
    for rows in (128, 256, 512, 1024, 2048, 4096):
        if rows > npix // 16: continue
        def run():
            ops.bn_backward(dz, None, y, mean, invstd, g, b, relu=True, rows=rows)
        run()
        e0.record()
        for _ in range(10): run()
        e1.record(); e1.synchronize()
        out.append("%d:%.1f" % (rows, e0.elapsed_time(e1) / 10 * 1e3))
    print((N, H, W, C), "default rows", int(max(1, min(1024, npix // 64))), " us(reduce+rowsum+apply):", " ".join(out))
