"""Per-shape search over the igemm variants (hd_conv_tune_override) on the hd_conv2d launches of one training step, timed as conv_table.py
times them (8 launches inside one hipGraph, replayed): tools/tune_conv.py times eager calls, which is host-bound below ~20 us per launch and
misreads every small layer.  Prints, per launch signature, the dispatcher's time and the best forced igemm variant.
    python tools/tune_conv_graph.py"""
import collections
import itertools
import os
import sys

os.environ.setdefault("HD_DET_GRAPH", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hallucidet_amd import _abi, ops, synthetic

lib = _abi.load()
lit = synthetic.make_module()
lit.encoder_decoder.runner.enable_graphs(False)
batch = synthetic.make_batch(8, device="cuda")
lit.fit_step(batch)
rec = []
orig = ops.conv2d


def spy(x, w, KH, KW, **kw):
    out = orig(x, w, KH, KW, **kw)
    rec.append((x, w, KH, KW, {k_: v_ for k_, v_ in kw.items() if k_ != "_defer"}))
    return out


ops.conv2d = spy
lit.fit_step(batch)
torch.cuda.synchronize()
ops.conv2d = orig


def sig(x, w, KH, KW, kw):
    return (tuple(x.shape), None if kw.get("x2") is None else tuple(kw["x2"].shape), tuple(w.shape), KH, kw.get("stride", 1), kw.get("pad", 0),
            kw.get("in_dil", 1), bool(kw.get("up1")), bool(kw.get("want_stats")), kw.get("res") is not None, kw.get("mask") is not None,
            bool(kw.get("out_nchw_f32")), kw.get("out_hw"), kw.get("cout"))


def timed(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


groups = collections.OrderedDict()
for item in rec:
    groups.setdefault(sig(*item), []).append(item)
tot_h = tot_b = 0.0
rows = []
for s, lst in groups.items():
    x, w, KH, KW, kw = lst[0]
    if kw.get("in_scale") is not None or kw.get("bstat") is not None:
        continue
    lib.hd_conv_tune_override(-1, -1, -1, -1)
    th = timed(lambda: orig(x, w, KH, KW, **kw))
    best = (th, "dispatcher")
    cout = w.shape[0] if kw.get("cout") is None else kw["cout"]
    for bm, bn, bk, deep in itertools.product((64, 128), (32, 64, 128), (32, 64), (0, 1)):
        if (bn == 32 and bm == 64) or (bn // 2 >= max(cout, 32) and bn > 32):
            continue
        if bk == 64 and (x.shape[3] % 64 or (kw.get("x2") is not None and kw["x2"].shape[3] % 64)):
            continue
        lib.hd_conv_tune_override(bm, bn, bk, deep)
        try:
            t = timed(lambda: orig(x, w, KH, KW, **kw))
        except Exception:
            continue
        if t < best[0]:
            best = (t, (bm, bn, bk, deep))
    lib.hd_conv_tune_override(-1, -1, -1, -1)
    n = len(lst)
    tot_h += th * n
    tot_b += best[0] * n
    rows.append((th * n - best[0] * n, n, th, best, s))
rows.sort(key=lambda r_: -r_[0])
for gain, n, th, best, s in rows[:40]:
    print("gain %6.1f us  x%2d  dispatcher %6.1f us  best %6.1f us %-18s x=%s x2=%s w=%s k=%d s=%d dil=%d up=%d%s%s%s" % (
        gain, n, th, best[0], best[1], s[0], s[1], s[2], s[3], s[4], s[6], s[7], " stats" if s[8] else "", " res" if s[9] else "", " mask" if s[10] else ""))
print("total dispatcher %.2f ms ; per-shape best %.2f ms (%.1f%% less) over %d launches / %d shapes" % (
    tot_h / 1e3, tot_b / 1e3, 100 * (1 - tot_b / tot_h), sum(r_[1] for r_ in rows), len(rows)))
