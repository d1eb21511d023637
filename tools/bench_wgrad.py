"""Record every hd_wgrad / hd_wgrad_reduce launch of one eager training step and time them back to back (HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import synthetic, ops

lit = synthetic.make_module()
batch = synthetic.make_batch(8, device="cuda")
lit.fit_step(batch)
rec, rrec = [], []
o_w, o_r = ops.wgrad, ops.wgrad_reduce


def spy_w(x, dy, KH, KW, **kw):
    out = o_w(x, dy, KH, KW, **kw)
    rec.append((x, dy, KH, KW, dict(kw), out.shape))
    return out


def spy_r(slab, dw, *a, **kw):
    rrec.append((slab, dw, a, dict(kw)))
    return o_r(slab, dw, *a, **kw)


ops.wgrad, ops.wgrad_reduce = spy_w, spy_r
import hallucidet_amd.segmentation_models.unet as U
r = lit.encoder_decoder.runner
r.enable_graphs(False)
lit.fit_step(batch)
torch.cuda.synchronize()
ops.wgrad, ops.wgrad_reduce = o_w, o_r
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tt = tf = 0.0
for (x, dy, KH, KW, kw, shp) in rec:
    o_w(x, dy, KH, KW, **kw)
    e0.record()
    for _ in range(5):
        o_w(x, dy, KH, KW, **kw)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    C2 = 0 if kw.get("x2") is None else kw["x2"].shape[3]
    M = dy.shape[0] * dy.shape[1] * dy.shape[2]
    fl = 2.0 * M * dy.shape[3] * KH * KW * (x.shape[3] + C2)
    tt += us; tf += fl
    print("wgrad %7.1f us %7.2f GF %6.1f TF/s  x=%s c2=%d dy=%s k=%d s=%d up=%d nsplit=%d" % (us, fl / 1e9, fl / us / 1e6, tuple(x.shape), C2, tuple(dy.shape), KH, kw.get("stride", 1), bool(kw.get("up1")), shp[0]))
print("wgrad total %.1f us, %.1f GF, %.1f TF/s" % (tt, tf / 1e9, tf / tt / 1e6))
tr = 0.0
for (slab, dw, a, kw) in rrec:
    o_r(slab, dw, *a, **kw)
    e0.record()
    for _ in range(5):
        o_r(slab, dw, *a, **kw)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    tr += us
    print("reduce %6.1f us slab=%s (%.1f MB)" % (us, tuple(slab.shape), slab.numel() * 4 / 1e6))
print("reduce total %.1f us" % tr)
