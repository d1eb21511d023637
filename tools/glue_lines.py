"""Which source lines of hallucidet_amd issue the small ATen ops of a training step: a TorchDispatchMode records, for every
ATen op, the innermost hallucidet_amd frame on the Python stack (ops issued by the autograd engine show up as 'backward')."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from hallucidet_amd import synthetic

SKIP = ("ops.py", "_abi.py")
per = collections.defaultdict(lambda: [0, collections.Counter()])


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        if not any(s in name for s in ("view", "reshape", "expand", "detach", "alias", "unsqueeze", "squeeze", "select", "slice", "t.default",
                                         "transpose", "permute", "unbind", "split", "as_strided", "empty", "_unsafe_view", "is_", "size", "stride")):
            fr = "backward / other"
            for f in reversed(traceback.extract_stack(limit=40)):
                if "hallucidet_amd" in f.filename and not f.filename.endswith(SKIP):
                    fr = "%s:%d %s" % (f.filename.split("hallucidet_amd/")[-1], f.lineno, f.name)
                    break
            per[fr][0] += 1
            per[fr][1][name] += 1
        return func(*args, **(kwargs or {}))


lit = synthetic.make_module()
batch = synthetic.make_batch(8, device="cuda")
for _ in range(3):
    lit.fit_step(batch)
torch.cuda.synchronize()
with Census():
    lit.fit_step(batch)
torch.cuda.synchronize()
rows = sorted(per.items(), key=lambda kv: -kv[1][0])
print("ATen ops that launch: %d" % sum(v[0] for v in per.values()))
for fr, (n, names) in rows[:int(os.environ.get("TOPN", "60"))]:
    print("%4d  %-66s %s" % (n, fr[:66], ", ".join("%s x%d" % (a.replace(".default", "").replace(".Tensor", ""), b) for a, b in names.most_common(5))))
