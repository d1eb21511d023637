"""Device kernels per detector-glue section of one training step (torch.profiler, record_function around each section):
launch count and summed device time per section, and the most frequent kernel names inside it."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity, record_function
from hallucidet_amd import synthetic
import hallucidet_amd.models.detection as D
import hallucidet_amd.utils.eval_forward_fasterrcnn as G

SECTIONS = ["pad_targets", "rpn_targets_sample_batched", "filter_proposals_padded", "rpn_loss_from_samples", "select_training_samples_batched",
            "roi_pool_rois", "fastrcnn_loss_flat", "postprocess_detections_flat"]


def wrap(mod, name):
    f = getattr(mod, name)

    def g(*a, **k):
        with record_function("SEC:" + name):
            return f(*a, **k)
    setattr(mod, name, g)


for n in SECTIONS:
    wrap(D, n)
wrap(G, "concat_box_prediction_layers")
lit = synthetic.make_module()
batch = synthetic.make_batch(8, device="cuda")
for _ in range(3):
    lit.fit_step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    lit.fit_step(batch)
    torch.cuda.synchronize()
ev = prof.events()
secs = [e for e in ev if e.name.startswith("SEC:")]
kern = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]
# map kernels to sections through the launching CPU op's time range (correlation by CPU-side launch time)
cpu_launch = {}
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
        for k in e.kernels:
            cpu_launch[id(k)] = e
per = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
tot_n = tot_t = 0
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    t0 = e.time_range.start
    sec = "other"
    for s in secs:
        if s.time_range.start <= t0 <= s.time_range.end:
            sec = s.name[4:]
            break
    for k in e.kernels:
        per[sec][0] += 1
        per[sec][1] += k.duration
        per[sec][2][k.name[:int(os.environ.get('NAMELEN', '70'))]] += 1
seen = set()
for sec, (n, t, names) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("%-34s %4d kernels %8.1f us" % (sec, n, t))
    if sec != "other":
        for nm, c in names.most_common(int(os.environ.get("TOPN", "8"))):
            print("      %3d x %s" % (c, nm))
