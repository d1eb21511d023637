import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import synthetic
N = int(os.environ.get("N", 8)); steps = int(os.environ.get("STEPS", 5))
lit = synthetic.make_module()
batch = synthetic.make_batch(N, device="cuda")
for _ in range(2):
    l = lit.fit_step(batch)
torch.cuda.synchronize()
print("loss", float(l), "scale", lit.scaler.scale_value)
t0 = time.time()
for _ in range(steps):
    l = lit.fit_step(batch)
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print("train step: %.2f ms -> %.1f img/s ; %.1f TFLOP/s (428.5 GFLOP/img)" % (dt * 1e3, N / dt, 428.5e9 * N / dt / 1e12))
print("loss", float(l), "max mem GB", torch.cuda.max_memory_allocated() / 2**30)
