"""Per-layer sweep of the weight-gradient split count (hd_wgrad + hd_wgrad_reduce timed together) on the launches of one step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import synthetic, ops

lit = synthetic.make_module()
batch = synthetic.make_batch(8, device="cuda")
lit.fit_step(batch)
rec = []
o_w = ops.wgrad


def spy_w(x, dy, KH, KW, **kw):
    out = o_w(x, dy, KH, KW, **kw)
    rec.append((x, dy, KH, KW, dict(kw), out.shape))
    return out


ops.wgrad = spy_w
lit.encoder_decoder.runner.enable_graphs(False)
lit.fit_step(batch)
torch.cuda.synchronize()
ops.wgrad = o_w
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot_h = tot_b = 0.0
for (x, dy, KH, KW, kw, shp) in rec:
    C2 = 0 if kw.get("x2") is None else kw["x2"].shape[3]
    Cin = x.shape[3] + C2
    Cout = dy.shape[3]
    M = dy.shape[0] * dy.shape[1] * dy.shape[2]
    K = KH * KW * Cin
    ns0 = shp[0]
    dw = torch.empty((Cout, Cin, KH, KW), dtype=torch.float32, device=x.device)
    res = {}
    for f in (0.125, 0.25, 0.5, 0.75, 1.0, 1.5, 2.0, 3.0):
        ns = max(1, min(int(round(ns0 * f)), M // 64))
        if ns in res or ns * Cout * K * 4 > (256 << 20):
            continue
        k2 = dict(kw); k2["nsplit"] = ns

        def run():
            slab = o_w(x, dy, KH, KW, **k2)
            ops.wgrad_reduce(slab, dw, KH, KW, Cin, Cout=Cout)
        run()
        e0.record()
        for _ in range(6):
            run()
        e1.record(); e1.synchronize()
        res[ns] = e0.elapsed_time(e1) / 6 * 1e3
    best = min(res, key=res.get)
    tot_h += res[ns0]; tot_b += res[best]
    print("M=%7d Cout=%4d K=%5d  heuristic ns=%3d %6.1f us | best ns=%3d %6.1f us | %s" % (M, Cout, K, ns0, res[ns0], best, res[best], " ".join("%d:%.0f" % (k, v) for k, v in sorted(res.items()))))
print("total heuristic %.2f ms, per-layer best %.2f ms" % (tot_h / 1e3, tot_b / 1e3))
