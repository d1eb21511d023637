"""Which python lines of the product issue ATen operators on device tensors in one training step (detector half issued eagerly,
U-Net graph-replayed): a TorchDispatchMode log with the first product frame of each call.  Candidates for fusion into the
hand-written kernels; the device time per operator family is in profiles/*kernel_stats.csv."""
import collections
import os
import sys
import traceback

os.environ.setdefault("HD_DET_GRAPH", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from hallucidet_amd import synthetic

lit = synthetic.make_module()
if os.environ.get("UNET_GRAPHS", "1") == "0":        # also log the U-Net's own operators (graph-replayed by default: invisible to dispatch)
    lit.use_graphs = False
    lit.encoder_decoder.runner.enable_graphs(False)
batch = synthetic.make_batch(int(os.environ.get("N", 8)), device="cuda")
CAPTURE = os.environ.get("ATEN_CAPTURE", "0") == "1"     # HD_DET_GRAPH=1 ATEN_CAPTURE=1: log from the first step on -- the step that captures the
if not CAPTURE:                                          # graphs logs what they hold, the steps after it log what is still issued eagerly
    for _ in range(3):
        lit.fit_step(batch)
    torch.cuda.synchronize()
SKIP = ("aten.view", "aten.reshape", "aten._unsafe_view", "aten.detach", "aten.slice", "aten.select", "aten.unsqueeze", "aten.squeeze", "aten.expand",
        "aten.permute", "aten.transpose", "aten.t.", "aten.alias", "aten.as_strided", "aten.split", "aten.unbind", "aten.empty", "aten.is_", "aten.sym_",
        "aten.stride", "aten.size", "aten.dim", "aten.numel", "aten.lift_fresh", "aten._local_scalar", "aten.unfold", "aten.narrow", "aten.chunk",
        "aten.view_as", "aten.result_type", "aten.new_empty", "aten.empty_like", "aten.record_stream", "aten.zeros_like" * 0 or "aten.__nothing__")
log = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            flat = [a for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)]
            flat += [b for a in args if isinstance(a, (list, tuple)) for b in a if isinstance(b, torch.Tensor)]
            if any(t.is_cuda for t in flat) or name.startswith(("aten.zeros", "aten.ones", "aten.full", "aten.arange", "aten.rand", "aten.tensor")):
                where = "?"
                for fr in reversed(traceback.extract_stack(limit=24)):
                    if "hallucidet_amd/" in fr.filename:
                        where = "%s:%d" % (fr.filename.split("hallucidet_amd/")[-1], fr.lineno)
                        break
                log[(where, name.replace("aten.", "").replace(".default", ""))] += 1
        return func(*args, **(kwargs or {}))


torch.autograd.set_multithreading_enabled(False)
for step in range(4 if CAPTURE else 1):
    log.clear()
    with Log():
        lit.fit_step(batch)
    torch.cuda.synchronize()
    if CAPTURE:
        print("step %d: %d device-side ATen calls" % (step, sum(log.values())))
        if step in (1, 2):
            continue
    per_line = collections.defaultdict(list)
    for (where, name), n in log.items():
        per_line[where].append((n, name))
    print("%d device-side ATen calls in one step, %d source lines" % (sum(log.values()), len(per_line)))
    for where, ops_ in sorted(per_line.items(), key=lambda kv: -sum(n for n, _ in kv[1])):
        print("%4d  %-58s %s" % (sum(n for n, _ in ops_), where, " ".join("%s x%d" % (nm, n) for n, nm in sorted(ops_, reverse=True))[:150]))
