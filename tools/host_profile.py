"""Where does the host spend its time in one step?  cProfile over 5 steps (GPU async)."""
import sys, os, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import synthetic
lit = synthetic.make_module()
batch = synthetic.make_batch(8, device="cuda")
for _ in range(3):
    lit.fit_step(batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    lit.fit_step(batch)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("cumulative")
ps.print_stats(45)
print(s.getvalue()[:9000])
