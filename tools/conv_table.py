"""Every hd_conv2d launch of one training step with its shape, warm graph-replayed duration, algorithmic FLOPs / bytes and its
two-roof floor max(FLOPs / 2.5 PFLOP/s, bytes / 8 TB/s); grouped by identical shape, sorted by the time above the floor.
    python tools/conv_table.py [detector|unet|all]"""
import collections
import os
import sys

os.environ.setdefault("HD_DET_GRAPH", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hallucidet_amd import ops, synthetic

which = sys.argv[1] if len(sys.argv) > 1 else "detector"
lit = synthetic.make_module()
lit.encoder_decoder.runner.enable_graphs(False)
batch = synthetic.make_batch(8, device="cuda")
lit.fit_step(batch)
rec, where = [], ["detector"]
orig = ops.conv2d


def spy(x, w, KH, KW, **kw):
    out = orig(x, w, KH, KW, **kw)
    rec.append((x, w, KH, KW, {k_: v_ for k_, v_ in kw.items() if k_ != "_defer"}, out[0] if isinstance(out, tuple) else out, where[0]))
    return out


rr = lit.encoder_decoder.runner
for name in ("forward", "backward"):
    fn = getattr(rr, name)

    def tagged(*a, _fn=fn, **k):
        where[0] = "unet"
        try:
            return _fn(*a, **k)
        finally:
            where[0] = "detector"
    setattr(rr, name, tagged)
ops.conv2d = spy
import hallucidet_amd.models.detection as D
import hallucidet_amd.segmentation_models.unet as U
lit.fit_step(batch)
torch.cuda.synchronize()
ops.conv2d = orig


def timed(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


groups = collections.OrderedDict()
for x, w, KH, KW, kw, y, wh in rec:
    if which != "all" and wh != which:
        continue
    C2 = 0 if kw.get("x2") is None else kw["x2"].shape[3]
    if kw.get("out_nchw_f32"):
        n, co, ho, wo = y.shape
    else:
        n, ho, wo, co = y.shape
    pooled = kw.get("pool2") is not None and kw["pool2"].get("done")
    if pooled:      # out_pool2: y is the 2x2 sum-pooled upsampled half; the convolution still computes every output of the full map
        ho, wo, co = 2 * ho, 2 * wo, (kw.get("cout") or w.shape[0])
    dil = kw.get("in_dil", 1)
    flops = 2.0 * n * ho * wo * co * KH * KW * (x.shape[3] + C2) / (dil * dil)
    by = x.numel() * 2 + (0 if kw.get("x2") is None else kw["x2"].numel() * 2) + w.numel() * 2 + y.numel() * y.element_size()
    by += sum(t.numel() * 2 for t in (kw.get("res"), kw.get("mask")) if t is not None)
    key = (wh, tuple(x.shape), C2, KH, kw.get("stride", 1), dil, co, (ho, wo), kw.get("res") is not None, kw.get("mask") is not None,
           bool(kw.get("want_stats")), kw.get("in_scale") is not None)
    if key not in groups:
        t = timed(lambda: orig(x, w, KH, KW, **kw))
        groups[key] = [0, t, flops, by]
    groups[key][0] += 1
rows = []
for key, (cnt, t, fl, by) in groups.items():
    floor = max(fl / 2.5e15, by / 8e12) * 1e6
    rows.append((cnt * (t - floor), cnt, t, floor, fl, by, key))
rows.sort(reverse=True)
tot = sum(r[1] * r[2] for r in rows)
print("%d launches, %d shapes, %.1f us warm total, floor total %.1f us" % (sum(r[1] for r in rows), len(rows), tot, sum(r[1] * r[3] for r in rows)))
for above, cnt, t, floor, fl, by, key in rows[:60]:
    wh, xs, C2, k, st, dil, co, hw, res, mask, stats, insc = key
    print("%7.1f above | %2d x %6.1f us (floor %5.1f, %s) %5.2f GF %6.1f MB | %s x%s%s k%d s%d d%d -> %d @%dx%d%s%s%s%s" % (
        above, cnt, t, floor, "mfma" if fl / 2.5e15 > by / 8e12 else "hbm", fl / 1e9, by / 1e6, wh[0], "x".join(map(str, xs)), ("+%d" % C2) if C2 else "",
        k, st, dil, co, hw[0], hw[1], " res" if res else "", " mask" if mask else "", " stats" if stats else "", " inbn" if insc else ""))
