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
    rec.append((x, dy, KH, KW, {k_: v_ for k_, v_ in kw.items() if k_ != "_defer"}, out.shape))
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
    from hallucidet_amd import _abi
    lib = _abi.load()
    tm0 = 128 if Cout > 64 else (64 if Cout > 32 else 32)
    for tm in (32, 64, 128):
        if tm > max(Cout, 32) * 2:
            continue
        for f in (0.25, 0.5, 0.75, 1.0, 1.5, 2.0, 3.0):
            ns = max(1, min(int(round(ns0 * f * tm0 / tm)) if tm != tm0 else int(round(ns0 * f)), M // 64))
            if (tm, ns) in res or ns * Cout * K * 4 > (256 << 20):
                continue
            k2 = {k_: v_ for k_, v_ in kw.items() if k_ != "_defer"}; k2["nsplit"] = ns
            lib.hd_wgrad_tune_override(tm)

            def run():
                slab = o_w(x, dy, KH, KW, **k2)
                ops.wgrad_reduce(slab, dw, KH, KW, Cin, Cout=Cout)
            run()
            e0.record()
            for _ in range(6):
                run()
            e1.record(); e1.synchronize()
            res[(tm, ns)] = e0.elapsed_time(e1) / 6 * 1e3
    lib.hd_wgrad_tune_override(-1)
    best = min(res, key=res.get)
    tot_h += res[(tm0, ns0)]; tot_b += res[best]
    bytm = {t: min((v, k[1]) for k, v in res.items() if k[0] == t) for t in (32, 64, 128) if any(k[0] == t for k in res)}
    print("M=%7d Cout=%4d K=%5d  heuristic tm=%d ns=%3d %6.1f us | best %s %6.1f us | best per tm: %s" % (M, Cout, K, tm0, ns0, res[(tm0, ns0)], best, res[best], " ".join("%d:%.0f(ns %d)" % (t, v[0], v[1]) for t, v in bytm.items())))
print("total heuristic %.2f ms, per-layer best %.2f ms" % (tot_h / 1e3, tot_b / 1e3))
