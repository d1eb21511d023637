"""The step-split main loop (TS) of the 8-wave patch-staged family (conv3x3_w8.hip) against the sub-step split it replaces, on every
hd_conv2d launch signature of one training step that the family takes, plus synthetic odd-chunk / ragged cases: forced tile 11/12/13 vs
15/16/17 -- max |difference| of every output (relative to the output's scale), error of both against an fp32 ATen convolution where the
launch is a plain one, warm graph-replayed duration.
    python tools/probe_w8_ts.py [--quick]"""
import collections
import os
import sys

os.environ.setdefault("HD_DET_GRAPH", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from hallucidet_amd import _abi, ops, synthetic

lib = _abi.load()
dev = torch.device("cuda:0")
quick = "--quick" in sys.argv

lit = synthetic.make_module()
lit.encoder_decoder.runner.enable_graphs(False)
batch = synthetic.make_batch(8, device="cuda")
lit.fit_step(batch)
rec = []
orig = ops.conv2d


def spy(x, w, KH, KW, **kw):
    out = orig(x, w, KH, KW, **kw)
    if KH == 3 and kw.get("stride", 1) == 1 and kw.get("pool2") is None:
        rec.append((x, w, KH, KW, {k_: v_ for k_, v_ in kw.items() if k_ not in ("_defer", "out")}))
    return out


ops.conv2d = spy
lit.fit_step(batch)
torch.cuda.synchronize()
ops.conv2d = orig


def sig(x, w, kw):
    return (tuple(x.shape), None if kw.get("x2") is None else tuple(kw["x2"].shape), tuple(w.shape), kw.get("in_dil", 1), bool(kw.get("up1")),
            bool(kw.get("want_stats")), kw.get("res") is not None, kw.get("mask") is not None, kw.get("bstat") is not None, kw.get("act", 0))


groups = collections.OrderedDict()
for x, w, KH, KW, kw in rec:
    groups.setdefault(sig(x, w, kw), []).append((x, w, KH, KW, kw))

# synthetic: odd chunk counts (Cin 64, 192 single source; 128 + 64 decoder concat), ragged maps, one image
gen = torch.Generator(device="cuda").manual_seed(1)


def synth(N, H, W, Cin, Cout, C_up=0, **kw):
    if C_up:
        x = (torch.randn(N, H // 2, W // 2, C_up, device=dev, generator=gen) * 0.5).half()
        x2 = (torch.randn(N, H, W, Cin - C_up, device=dev, generator=gen) * 0.5).half()
        kw.update(x2=x2, up1=True)
    else:
        x = (torch.randn(N, H, W, Cin, device=dev, generator=gen) * 0.5).half()
    w = (torch.randn(Cout, 9 * Cin, device=dev, generator=gen) / (9 * Cin) ** 0.5).half()
    kw.update(pad=1)
    return x, w, 3, 3, kw


extra = [synth(8, 32, 40, 64, 128), synth(3, 30, 44, 192, 128, want_stats=True), synth(8, 64, 80, 192, 64, C_up=128, want_stats=True),
         synth(1, 16, 24, 320, 256), synth(2, 50, 70, 128, 136), synth(8, 32, 40, 384, 128, C_up=256, want_stats=True)]
for e in extra:
    groups.setdefault(("synthetic",) + sig(e[0], e[1], e[4]), []).append(e)


def timed(fn, reps=6):
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
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


def flat(o):
    return [t.float() for t in (o if isinstance(o, tuple) else (o,)) if torch.is_tensor(t)]


tot = {"auto": 0.0, "ss": 0.0, "ts": 0.0}
worst = 0.0
nbad = 0
for s, lst in groups.items():
    x, w, KH, KW, kw = lst[0]
    n = len(lst)
    lib.hd_conv_tune_w8(-1, 1)
    ref_out = flat(orig(x, w, KH, KW, **kw))
    t_auto = timed(lambda: orig(x, w, KH, KW, **kw))
    line = []
    best_ss = best_ts = 1e9
    for base in (11, 12, 13):
        res = {}
        for cfg in (base, base + 4):
            lib.hd_conv_tune_w8(cfg, 1)
            try:
                o = flat(orig(x, w, KH, KW, **kw))
                t = timed(lambda: orig(x, w, KH, KW, **kw))
            except RuntimeError as ex:
                res[cfg] = None
                continue
            res[cfg] = (o, t)
        lib.hd_conv_tune_w8(-1, 1)
        if res.get(base) is None or res.get(base + 4) is None:
            continue
        (o_ss, t_ss), (o_ts, t_ts) = res[base], res[base + 4]
        d = 0.0
        for a, b in zip(o_ss, o_ts):
            if a.shape != b.shape:
                d = float("inf")
                break
            sc = float(a.abs().max()) + 1e-6
            d = max(d, float((a - b).abs().max()) / sc)
        worst = max(worst, d)
        if not d < 4e-3:
            nbad += 1
        best_ss, best_ts = min(best_ss, t_ss), min(best_ts, t_ts)
        line.append("%d: %6.1f -> %6.1f us (d %.1e)" % (base, t_ss, t_ts, d))
    err = ""
    plain = kw.get("x2") is None and kw.get("res") is None and kw.get("mask") is None and kw.get("bstat") is None and kw.get("in_dil", 1) == 1 and \
        kw.get("in_scale") is None and kw.get("bias") is None and kw.get("act", 0) == 0
    if plain and not quick:
        lib.hd_conv_tune_w8(15, 1)
        y = flat(orig(x, w, KH, KW, **kw))[0]
        lib.hd_conv_tune_w8(-1, 1)
        yr = F.conv2d(x.permute(0, 3, 1, 2).float(), w.float().reshape(w.shape[0], 3, 3, -1).permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
        if y.shape == yr.shape:
            err = " | vs fp32 %.2e" % (float((y - yr).abs().max()) / (float(yr.abs().max()) + 1e-6))
    if best_ss < 1e9:
        tot["auto"] += n * t_auto
        tot["ss"] += n * best_ss
        tot["ts"] += n * best_ts
    print("%2d x auto %6.1f | %s%s | %s" % (n, t_auto, " | ".join(line), err, s), flush=True)
print("launch-weighted: dispatcher %.1f us, best sub-step tile %.1f us, best step-split tile %.1f us; worst relative difference %.2e; %d bad" % (
    tot["auto"], tot["ss"], tot["ts"], worst, nbad))
