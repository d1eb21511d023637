"""Slabs / direct gradients of the 8-wave 3x3 weight-gradient kernel for fixed seeds -> a file (run once per library, HD_HIP_LIB=...), or
`compare a b`: bit-equality of two such files.  Used to check a re-decomposition of the kernel's waves against the previous build, and to
time both (warm, graph-replayed).
    HD_HIP_LIB=old.so python tools/probe_wg8_compare.py dump /tmp/a.pt; python tools/probe_wg8_compare.py dump /tmp/b.pt;
    python tools/probe_wg8_compare.py compare /tmp/a.pt /tmp/b.pt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

if sys.argv[1] == "compare":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    bad = 0
    for k in a:
        same = torch.equal(a[k], b[k])
        bad += not same
        if not same:
            print("DIFFERS", k, float((a[k] - b[k]).abs().max()))
    print("%d tensors compared, %d differ" % (len(a), bad))
    sys.exit(1 if bad else 0)

from hallucidet_amd import ops

dev = torch.device("cuda:0")
gen = torch.Generator(device="cuda").manual_seed(0)
r = lambda *sh: (torch.randn(*sh, device=dev, generator=gen) * 0.5).half()
SHAPES = [(8, 32, 40, 256, 0, 256, 16), (8, 16, 20, 512, 0, 512, 4), (8, 64, 80, 128, 0, 128, 64), (8, 128, 160, 64, 0, 64, 256), (2, 19, 21, 64, 0, 128, 3),
          (8, 32, 40, 512, 256, 256, 5), (8, 32, 40, 256, 0, 256, 1), (3, 30, 44, 128, 64, 64, 1), (8, 16, 20, 512, 0, 512, 1)]


def timed(fn, reps=4):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


out = {}
for i, (N, H, W, C1, C2, Cout, ns) in enumerate(SHAPES):
    x = r(N, H // 2 if C2 else H, W // 2 if C2 else W, C1)
    x2 = r(N, H, W, C2) if C2 else None
    dy = r(N, H, W, Cout)
    kw = dict(x2=x2, pad=1, up1=bool(C2), nsplit=ns)
    out["slab%d" % i] = ops.wgrad(x, dy, 3, 3, **kw).cpu()
    t = timed(lambda: ops.wgrad(x, dy, 3, 3, **kw))
    print("%-40s nsplit %3d: %7.1f us" % ((N, H, W, C1, C2, Cout), ns, t), flush=True)
torch.save(out, sys.argv[2])
