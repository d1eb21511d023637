"""Which ATen ops does one training step still dispatch outside the C ABI, and from where?
    python tools/aten_census.py [detector]
Logs every op that launches GPU work during one `fit_step` (graphs warm, so the U-Net's captured launches do not appear), grouped
by (op, calling line inside hallucidet_amd)."""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from hallucidet_amd import synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = ("aten.view", "aten._unsafe_view", "aten.detach", "aten.t.", "aten.transpose", "aten.permute", "aten.expand", "aten.slice", "aten.select",
        "aten.unsqueeze", "aten.squeeze", "aten.alias", "aten.as_strided", "aten.reshape", "aten.empty", "aten.split", "aten.unbind", "aten.size",
        "aten.stride", "aten.is_", "aten.sym_", "aten.lift_fresh", "aten.new_empty", "aten.empty_like", "aten.unfold", "aten.narrow",
        "aten._local_scalar_dense", "aten.item", "aten.record_stream", "aten.is_pinned", "aten.set_", "aten.resize_")


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.n = collections.Counter()
        self.shapes = {}

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not name.startswith(SKIP):
            cuda = any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + list((kwargs or {}).values())) or (
                isinstance(out, torch.Tensor) and out.is_cuda)
            if cuda:
                where = "?"
                for fr in reversed(traceback.extract_stack(limit=30)):
                    if fr.filename.startswith(ROOT) and "tools/" not in fr.filename:
                        where = "%s:%d" % (os.path.relpath(fr.filename, ROOT), fr.lineno)
                        break
                key = (name, where)
                self.n[key] += 1
                if key not in self.shapes:
                    self.shapes[key] = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:3]
        return out


det = sys.argv[1] if len(sys.argv) > 1 else "fasterrcnn"
lit = synthetic.make_module(detector_name=det)
batch = synthetic.make_batch(8, device="cuda")
for _ in range(3):
    lit.fit_step(batch)
torch.cuda.synchronize()
c = Census()
with c:
    lit.fit_step(batch)
torch.cuda.synchronize()
tot = sum(c.n.values())
print("%d dispatched GPU ops in one step" % tot)
for (name, where), k in sorted(c.n.items(), key=lambda kv: -kv[1]):
    print("%4d  %-42s %-58s %s" % (k, name, where, c.shapes[(name, where)]))
