"""Lists the host-synchronising calls of one steady-state training step (torch.cuda.set_sync_debug_mode('warn')) with the
python frame that issued each: every one of them drains the launch queue and leaves the GPU idle until the host catches up."""
import os
import sys
import traceback
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hallucidet_amd import synthetic

lit = synthetic.make_module()
batch = synthetic.make_batch(int(os.environ.get("N", 8)), device="cuda")
for _ in range(3):
    lit.fit_step(batch)
torch.cuda.synchronize()
seen = []


def show(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "site-packages" not in f.filename and "dist-packages" not in f.filename]
    seen.append(str(message)[:60] + " | " + " <- ".join("%s:%d" % ("/".join(f.filename.split("/")[-2:]), f.lineno) for f in reversed(st[-7:-1])))


warnings.showwarning = show
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
lit.fit_step(batch)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print("%d synchronising calls in one step" % len(seen))
for s in seen:
    print("  ", s)
