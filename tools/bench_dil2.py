"""Time the stride-2 data-gradient launches (in_dil = 2) of one training step; run with HD_CONV_PARITY=0 / 1 to A/B the
output-parity decomposition."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import synthetic, ops

lit = synthetic.make_module()
batch = synthetic.make_batch(8, device="cuda")
lit.fit_step(batch)
rec = []
orig = ops.conv2d


def spy(x, w, KH, KW, **kw):
    out = orig(x, w, KH, KW, **kw)
    if kw.get("in_dil", 1) == 2:
        rec.append((x, w, KH, KW, {k_: v_ for k_, v_ in kw.items() if k_ != "_defer"}))
    return out


ops.conv2d = spy
lit.encoder_decoder.runner.enable_graphs(False)
lit.fit_step(batch)
torch.cuda.synchronize()
ops.conv2d = orig
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot = 0.0
for x, w, KH, KW, kw in rec:
    kw = {k_: v_ for k_, v_ in kw.items() if k_ != "_defer"}; kw.pop("out", None)
    orig(x, w, KH, KW, **kw)
    e0.record()
    for _ in range(10):
        orig(x, w, KH, KW, **kw)
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) * 100
    tot += t
    print("%7.1f us  x=%s w=%s k=%d out=%s cout=%s res=%d mask=%d" % (t, tuple(x.shape), tuple(w.shape), KH, kw.get("out_hw"), kw.get("cout"),
                                                                   kw.get("res") is not None, kw.get("mask") is not None))
print("PARITY=%s total %.1f us over %d launches" % (os.environ.get("HD_CONV_PARITY", "1"), tot, len(rec)))
