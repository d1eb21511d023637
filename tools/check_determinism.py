import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
bad = 0
for (N, H, W, C, Cout, K, s, p) in [(2, 75, 75, 256, 256, 3, 1, 1), (2, 75, 75, 64, 256, 1, 1, 0), (2, 38, 38, 512, 128, 1, 1, 0), (2, 19, 19, 256, 256, 3, 1, 1),
                                   (2, 10, 10, 512, 512, 3, 1, 1), (2, 75, 75, 256, 64, 1, 1, 0), (2, 38, 38, 128, 128, 3, 2, 1), (2, 150, 150, 64, 64, 3, 1, 1)]:
    x = (torch.randn(6, H, W, C, device=dev) * 0.5).half()
    w = (torch.randn(Cout, K * K * C, device=dev) * 0.05).half()
    y6 = ops.conv2d(x, w, K, K, stride=s, pad=p)
    y6b = ops.conv2d(x, w, K, K, stride=s, pad=p)
    y2 = ops.conv2d(x[:2].contiguous(), w, K, K, stride=s, pad=p)
    torch.cuda.synchronize()
    same_run = torch.equal(y6, y6b)
    same_batch = torch.equal(y6[:2], y2)
    # round 6: the tile cost model sees the launch's own batch, so differently batched launches may run different tiles and then agree to
    # fp16 rounding (a few ulp of the output), not bit for bit; run-to-run identity at a fixed batch is the property that is kept
    diff = float((y6[:2].float() - y2.float()).abs().max())
    tol = 4e-3 * max(1.0, float(y2.float().abs().max()))
    print((N, H, W, C, Cout, K, s, p), "rerun identical:", same_run, " N=6 vs N=2 identical:", same_batch, "max diff %.3g (bound %.3g)" % (diff, tol))
    bad += (not same_run) + (diff > tol)
print("BAD" if bad else "OK")
