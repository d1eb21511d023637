"""Per-step GPU time of the bench workload from HIP events at the step boundaries (no host sync inside the run)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import synthetic

if os.environ.get("THREADS"):
    torch.set_num_threads(int(os.environ["THREADS"]))
lit = synthetic.make_module()
batch = synthetic.make_batch(8, device="cuda")
for _ in range(10):
    lit.fit_step(batch)
torch.cuda.synchronize()
n = int(os.environ.get("STEPS", "100"))
if os.environ.get("NOGC"):
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
host = []
ev[0].record()
for i in range(n):
    t0 = time.perf_counter()
    lit.fit_step(batch)
    host.append(time.perf_counter() - t0)
    ev[i + 1].record()
torch.cuda.synchronize()
d = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
h = sorted(x * 1e3 for x in host)
print("gpu step ms: min %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f mean %.2f" % (d[0], d[n // 10], d[n // 2], d[n * 9 // 10], d[-1], sum(d) / n))
print("host issue ms: min %.2f p50 %.2f p90 %.2f max %.2f" % (h[0], h[n // 2], h[n * 9 // 10], h[-1]))
