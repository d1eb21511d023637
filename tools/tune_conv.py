"""Exhaustive per-shape search over hd_conv2d's igemm variants on the launches of one real training step: prints, per unique
launch signature, the heuristic's time and the best (bm, bn, bk, deep), and the total a perfect dispatcher would reach."""
import sys, os, itertools, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import synthetic, ops, _abi

lib = _abi.load()
lit = synthetic.make_module()
batch = synthetic.make_batch(8, device="cuda")
lit.fit_step(batch)
rec = []
orig = ops.conv2d


def spy(x, w, KH, KW, **kw):
    out = orig(x, w, KH, KW, **kw)
    rec.append((x, w, KH, KW, {k_: v_ for k_, v_ in kw.items() if k_ != "_defer"}))
    return out


ops.conv2d = spy
r = lit.encoder_decoder.runner
r.enable_graphs(False)
lit.fit_step(batch)
torch.cuda.synchronize()
ops.conv2d = orig


def sig(x, w, KH, KW, kw):
    return (tuple(x.shape), None if kw.get("x2") is None else tuple(kw["x2"].shape), tuple(w.shape), KH, kw.get("stride", 1), kw.get("pad", 0),
            kw.get("in_dil", 1), bool(kw.get("up1")), bool(kw.get("want_stats")), kw.get("res") is not None, kw.get("mask") is not None,
            bool(kw.get("out_nchw_f32")), kw.get("out_hw"), kw.get("cout"))


groups = collections.OrderedDict()
for x, w, KH, KW, kw in rec:
    groups.setdefault(sig(x, w, KH, KW, kw), []).append((x, w, KH, KW, kw))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timeit(x, w, KH, KW, kw, reps=8):
    orig(x, w, KH, KW, **kw)
    e0.record()
    for _ in range(reps):
        orig(x, w, KH, KW, **kw)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot_h = tot_b = 0.0
rows = []
dump = []
for s, lst in groups.items():
    x, w, KH, KW, kw = lst[0]
    lib.hd_conv_tune_override(-1, -1, -1, -1)
    th = timeit(x, w, KH, KW, kw)
    best = (th, "heuristic")
    allt = {}
    cout = w.shape[0] if kw.get("cout") is None else kw["cout"]
    for bm, bn, bk, deep in itertools.product((64, 128), (32, 64, 128), (32, 64), (0, 1)):
        if bn == 32 and bm == 64:
            continue
        if bn // 2 >= max(cout, 32) and bn > 32:      # tile twice as wide as the output: pointless
            continue
        lib.hd_conv_tune_override(bm, bn, bk, deep)
        try:
            t = timeit(x, w, KH, KW, kw)
        except Exception:
            continue
        allt["%d,%d,%d,%d" % (bm, bn, bk, deep)] = t
        if t < best[0]:
            best = (t, (bm, bn, bk, deep))
    lib.hd_conv_tune_override(-1, -1, -1, -1)
    n = len(lst)
    tot_h += th * n
    tot_b += best[0] * n
    rows.append((th * n - best[0] * n, n, th, best, s))
    dump.append(dict(sig=[list(v) if isinstance(v, tuple) else v for v in s], n=n, heuristic=th, all=allt))
rows.sort(key=lambda r_: -r_[0])
for gain, n, th, best, s in rows[:40]:
    print("gain %7.1f us  x%2d  heuristic %7.1f us  best %7.1f us %-20s x=%s x2=%s w=%s k=%d s=%d dil=%d up=%d" % (gain, n, th, best[0], best[1], s[0], s[1], s[2], s[3], s[4], s[6], s[7]))
print("total heuristic %.2f ms ; per-shape best %.2f ms (%.1f%% less) over %d launches / %d shapes" % (tot_h / 1e3, tot_b / 1e3, 100 * (1 - tot_b / tot_h), len(rec), len(groups)))

import json
os.makedirs("gpurun_out", exist_ok=True)
json.dump(dump, open("gpurun_out/tune_conv.json", "w"))
