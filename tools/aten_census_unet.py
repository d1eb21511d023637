import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
sys.argv = ["x"]
import torch, collections, traceback
from torch.utils._python_dispatch import TorchDispatchMode
from hallucidet_amd import synthetic
ROOT = "/root/repo"
SKIP = ("aten.view", "aten._unsafe_view", "aten.detach", "aten.t.", "aten.transpose", "aten.permute", "aten.expand", "aten.slice", "aten.select",
        "aten.unsqueeze", "aten.squeeze", "aten.alias", "aten.as_strided", "aten.reshape", "aten.empty", "aten.split", "aten.unbind", "aten.size",
        "aten.stride", "aten.is_", "aten.sym_", "aten.lift_fresh", "aten.new_empty", "aten.empty_like", "aten.unfold", "aten.narrow",
        "aten._local_scalar_dense", "aten.item", "aten.record_stream", "aten.is_pinned", "aten.set_", "aten.resize_")
class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__(); self.n = collections.Counter(); self.shapes = {}
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not name.startswith(SKIP):
            where = "?"
            for fr in reversed(traceback.extract_stack(limit=30)):
                if fr.filename.startswith(ROOT) and "tools/" not in fr.filename and "/tmp" not in fr.filename:
                    where = "%s:%d" % (os.path.relpath(fr.filename, ROOT), fr.lineno); break
            key = (name, where); self.n[key] += 1
            if key not in self.shapes: self.shapes[key] = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:2]
        return out
lit = synthetic.make_module()
lit.encoder_decoder.runner.enable_graphs(False)
batch = synthetic.make_batch(8, device="cuda")
for _ in range(2): lit.fit_step(batch)
torch.cuda.synchronize()
c = Census()
with c: lit.fit_step(batch)
torch.cuda.synchronize()
rows = [(k, v) for k, v in c.n.items() if "segmentation_models" in k[0][1] or "ops.py" in k[0][1] or "optim" in k[0][1] or "train_hall" in k[0][1] or k[0][1] == "?"]
print(sum(c.n.values()), "ops total;", "U-Net / ops / optimizer side:")
for (name, where), k in sorted(c.n.items(), key=lambda kv: -kv[1]):
    if any(t in where for t in ("segmentation_models", "ops.py", "optim.py", "utils/utils.py", "distributed.py")) or where == "?":
        print("%4d  %-36s %-52s %s" % (k, name, where, c.shapes[(name, where)]))
