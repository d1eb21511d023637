"""What the slab reductions of one training step consist of (hd_wgrad_reduce_multi: one launch per backward segment): per launch the
tensors, their split counts, the form `red_mode` picks and the slab bytes read.  python tools/wred_table.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hallucidet_amd import ops, synthetic

lit = synthetic.make_module()
batch = synthetic.make_batch(8, device="cuda")
lit.use_graphs = False
lit.encoder_decoder.runner.enable_graphs(False)
for _ in range(2):
    lit.fit_step(batch)
torch.cuda.synchronize()
log = []
orig = ops.WgradReduceBatch.flush


def flush(self):
    log.append([(tuple(s.shape), a) for s, _, a in self.items])
    return orig(self)


ops.WgradReduceBatch.flush = flush
lit.fit_step(batch)
torch.cuda.synchronize()
tot = 0
for i, items in enumerate(log):
    b = sum(s[0] * s[1] * s[2] * 4 for s, _ in items)
    tot += b
    print("launch %d: %d tensors, %.1f MB of slabs" % (i, len(items), b / 1e6))
    for s, a in items:
        nsplit, cout, K = s
        total4 = a[2] * K // 4
        mode = ("split" if (4096 <= total4 <= 65536 and nsplit >= 16) else "wave" if (total4 <= 16384 and nsplit >= 64) else "plain")
        print("     slab %4d x %4d x %5d  (%7.2f MB)  quads %7d  %s" % (nsplit, cout, K, nsplit * cout * K * 4 / 1e6, total4, mode))
print("total %.1f MB" % (tot / 1e6))
