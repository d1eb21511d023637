import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import synthetic
N = int(os.environ.get("N", 8)); steps = int(os.environ.get("STEPS", 5))
if os.environ.get("HD_FORCE_DIST") == "1":
    # the N > 1 code path (RCCL process group, bucket hooks between the two backward graphs, exchange_and_step) at world size 1: what the
    # data-parallel schedule costs before a byte crosses xGMI (profiles/r06_forced_dist_steady_state.txt)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
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
