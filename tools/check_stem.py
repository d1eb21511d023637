import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import ops
from oracle import kernels as ok
dev = torch.device("cuda:0")
torch.manual_seed(0)
for N in (2, 6):
    x = torch.zeros(N, 300, 300, 8).half(); x[..., :3] = (torch.rand(N, 300, 300, 3)).half()
    w = (torch.randn(64, 49 * 8) * 0.05).half()
    want, _ = ok.conv2d_nhwc(x, w, 7, 7, stride=2, pad=3, act=1)
    for force in ("", "1", "0"):
        if force: os.environ["X"] = force
        got = ops.conv2d(x.to(dev), w.to(dev), 7, 7, stride=2, pad=3, act=1).float().cpu()
        e = (got - want).abs()
        print("N", N, "max err", float(e.max()), "mean", float(e.mean()), "bad frac", float((e > 0.05).float().mean()))
        break
