import sys, os, time, gc
sys.path.insert(0, "/root/repo")
import torch
from hallucidet_amd import synthetic
lit = synthetic.make_module()
batch = synthetic.make_batch(8, device="cuda")
for _ in range(5):
    lit.fit_step(batch)
torch.cuda.synchronize()
for mode in ("gc on", "gc off"):
    if mode == "gc off":
        gc.collect(); gc.disable()
    ts = []
    for w in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            lit.fit_step(batch)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 20 * 1e3)
    print(mode, " ".join("%.2f" % t for t in ts))
    # per-step host time of issuing (no sync)
    hs = []
    for _ in range(40):
        t0 = time.perf_counter(); lit.fit_step(batch); hs.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    hs.sort()
    print("   host issue ms: median %.2f p90 %.2f max %.2f" % (hs[20], hs[36], hs[-1]))
