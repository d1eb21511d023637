"""GPU time of the detector glue sections inside real training steps (synchronising wrappers; serialises the step)."""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import synthetic
import hallucidet_amd.models.detection as D
import hallucidet_amd.utils.eval_forward_fasterrcnn as G

acc = collections.defaultdict(float)
cnt = collections.defaultdict(int)


def wrap(mod, name):
    f = getattr(mod, name)

    def g(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc[name] += time.perf_counter() - t; cnt[name] += 1
        return r
    setattr(mod, name, g)


for n in ("pad_targets", "rpn_targets_sample_batched", "filter_proposals_padded", "rpn_loss_from_samples", "select_training_samples_batched",
          "roi_pool_rois", "fastrcnn_loss_flat", "postprocess_detections_flat"):
    wrap(D, n)
wrap(G, "concat_box_prediction_layers")
lit = synthetic.make_module()
batch = synthetic.make_batch(8, device="cuda")
for _ in range(3):
    lit.fit_step(batch)
acc.clear(); cnt.clear()
S = 10
for _ in range(S):
    lit.fit_step(batch)
torch.cuda.synchronize()
tot = 0
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("%-36s %7.3f ms/step (%d calls/step)" % (k, v / S * 1e3, cnt[k] // S)); tot += v
print("sum %.3f ms/step" % (tot / S * 1e3))
