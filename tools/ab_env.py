"""In-process A/B of an environment knob that the host code reads at every step (e.g. HD_SIDE_STREAM): alternates the
values over several rounds inside ONE process, so clocks / allocator state / box are shared.
usage: python tools/ab_env.py HD_SIDE_STREAM 0 1 [rounds] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import synthetic

name, vals = sys.argv[1], sys.argv[2:4]
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
lit = synthetic.make_module()
batch = synthetic.make_batch(8, device="cuda")
for v in vals:
    os.environ[name] = v
    for _ in range(4):
        lit.fit_step(batch)
res = {v: [] for v in vals}
for r in range(rounds):
    for v in vals:
        os.environ[name] = v
        lit.fit_step(batch)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(steps):
            lit.fit_step(batch)
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t) / steps * 1e3)
for v in vals:
    print("%s=%s: %s  mean %.2f ms/step" % (name, v, " ".join("%.2f" % x for x in res[v]), sum(res[v]) / len(res[v])))
