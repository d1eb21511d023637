import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from hallucidet_amd import synthetic
torch.set_num_threads(8)
lit = synthetic.make_module()
batch = synthetic.make_batch(8, device="cuda")
mode = sys.argv[1]
if mode == "seg":
    import hallucidet_amd.train_hallucidet as T
    import hallucidet_amd.distributed as Dd
    Dd_is = Dd.is_dist
    # force the segmented backward with a no-op hook
    orig = lit.fit_step
    lit.averager.bucket_ready = lambda lo, hi: None
    Dd.is_dist = lambda: True
    lit.averager.start = lambda g: None
    lit.averager.finish = lambda g, defer_mean=False: 1.0
    lit.averager.begin = lambda g: None
for _ in range(8): lit.fit_step(batch)
torch.cuda.synchronize()
n = 100
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    lit.fit_step(batch); ev[i + 1].record()
torch.cuda.synchronize()
d = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
print(mode, "p50 %.2f mean %.2f" % (d[n // 2], sum(d) / n))
