"""Per-shape sweep of hd_conv2d's 8-wave family (conv_igemm_w8.hip) over the conv launches of one real training step.

For every unique launch signature of `fit_step` (U-Net eager + detector): time the shipped heuristic, then every eligible
(tile cfg, split-K) of the 8-wave family, CHECK each result against the heuristic's output (max |diff| relative to the
output's scale) and print / dump the table a dispatcher rule set is derived from.
    python tools/tune_w8.py [--quick]
"""
import collections
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import synthetic, ops, _abi

lib = _abi.load()
TILES = [(256, 128), (128, 128), (256, 64), (128, 64), (128, 256), (64, 128), (64, 256)]
quick = "--quick" in sys.argv

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
lit.encoder_decoder.runner.enable_graphs(False)
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


def run(x, w, KH, KW, kw):
    kw = {k_: v_ for k_, v_ in kw.items() if k_ != "_defer"}
    kw.pop("out", None)
    return orig(x, w, KH, KW, **kw)


def timeit(x, w, KH, KW, kw, reps=6):
    run(x, w, KH, KW, kw)
    e0.record()
    for _ in range(reps):
        run(x, w, KH, KW, kw)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def flat(o):
    return [t.float() for t in (o if isinstance(o, tuple) else (o,))]


rows, dump = [], []
tot_h = tot_b = tot_a = 0.0
bad = 0
for s, lst in groups.items():
    x, w, KH, KW, kw = lst[0]
    n = len(lst)
    N, Hs, Ws, C1 = x.shape
    cout = w.shape[0] if kw.get("cout") is None else kw["cout"]
    lib.hd_conv_tune_w8(-2, 0)
    th = timeit(x, w, KH, KW, kw)
    ref = flat(run(x, w, KH, KW, kw))
    M = ref[0].numel() // cout
    K = w.shape[1]
    best = (th, "4-wave")
    allt = {}
    eligible = (not kw.get("out_nchw_f32")) and cout % 8 == 0 and (kw.get("x2") is None or (C1 % 64 == 0 and kw["x2"].shape[3] % 64 == 0))
    if False:      # (the im2col 8-wave family with split-K, cfg 0-6, was removed in round 3: no faster than the 4-wave kernels on any shape)
        for cfg, (bm, bn) in enumerate(TILES):
            if bn // 2 >= max(cout, 64) and bn > 64:
                continue
            tiles = -(-M // bm) * -(-cout // bn)
            nk = -(-K // 64)
            cands = [1]
            if tiles < 256:
                cands += [sl for sl in (2, 3, 4, 6, 8, 12, 16) if sl <= nk // 2 and tiles * sl <= 1024]
            if os.environ.get("NO_IM2COL"):
                cands = []
            for sl in cands:
                lib.hd_conv_tune_w8(cfg, sl)
                try:
                    t = timeit(x, w, KH, KW, kw)
                    got = flat(run(x, w, KH, KW, kw))
                except Exception as e:  # noqa: BLE001
                    print("FAIL", cfg, sl, s, e)
                    continue
                # parity vs the 4-wave family: output to fp16 rounding of a differently ordered fp32 sum; stats summed over tiles
                scale = max(1e-3, float(ref[0].abs().max()))
                err = float((got[0] - ref[0]).abs().max()) / scale
                if len(ref) > 1:
                    a, b = got[1].sum(0), ref[1].sum(0)
                    err = max(err, float(((a - b).abs() / (b.abs() + 1e-2 * float(b.abs().max()) + 1e-6)).max()) * 0.5)
                if not (err < 4e-3):
                    bad += 1
                    print("MISMATCH cfg %d slices %d err %.3g  %s" % (cfg, sl, err, s))
                allt["%d,%d" % (cfg, sl)] = t
                if t < best[0]:
                    best = (t, (cfg, sl))
    p8 = KH == 3 and kw.get("stride", 1) == 1 and kw.get("pad", 0) == 1 and kw.get("in_dil", 1) == 1 and C1 % 64 == 0 and eligible and (
        (kw.get("x2") is None and not kw.get("up1")) or (kw.get("x2") is not None and kw.get("up1")))
    if p8 and th * n > (40.0 if quick else 0.0):
        for cfg in (10, 11, 12, 13):
            if cfg in (10, 11) and cout <= 64:
                continue
            lib.hd_conv_tune_w8(cfg, 1)
            t = timeit(x, w, KH, KW, kw)
            got = flat(run(x, w, KH, KW, kw))
            scale = max(1e-3, float(ref[0].abs().max()))
            err = float((got[0] - ref[0]).abs().max()) / scale
            if len(ref) > 1:
                a, b = got[1].sum(0), ref[1].sum(0)
                err = max(err, float(((a - b).abs() / (b.abs() + 1e-2 * float(b.abs().max()) + 1e-6)).max()) * 0.5)
            if not (err < 4e-3):
                bad += 1
                print("MISMATCH patch cfg %d err %.3g  %s" % (cfg, err, s))
            allt["%d,1" % cfg] = t
            if t < best[0]:
                best = (t, (cfg, 1))
    lib.hd_conv_tune_w8(-1, 0)
    ta = timeit(x, w, KH, KW, kw)
    tot_a += ta * n
    tot_h += th * n
    tot_b += best[0] * n
    fl = 2.0 * M * cout * K / (kw.get("in_dil", 1) ** 2)
    rows.append((th * n - best[0] * n, n, th, best, s, fl, M, K, cout))
    dump.append(dict(sig=[list(v) if isinstance(v, tuple) else v for v in s], n=n, M=M, K=K, cout=cout, four_wave=th, auto=ta, w8=allt))
rows.sort(key=lambda r_: -r_[0])
for gain, n, th, best, s, fl, M, K, cout in rows[:70]:
    print("gain %7.1f us x%2d  4w %7.1f us (%4.0f TF)  best %7.1f us (%4.0f TF) %-10s M=%6d K=%5d N=%4d k=%d s=%d dil=%d up=%d x2=%s st=%d res=%d mask=%d" % (
        gain, n, th, fl / th / 1e6, best[0], fl / best[0] / 1e6, best[1], M, K, cout, s[3], s[4], s[6], s[7], None if s[1] is None else s[1][3], s[8], s[9], s[10]))
print("total 4-wave %.2f ms ; shipped dispatcher %.2f ms ; per-shape best %.2f ms (%.1f%% less than 4-wave) over %d launches / %d shapes ; mismatches %d" % (
    tot_h / 1e3, tot_a / 1e3, tot_b / 1e3, 100 * (1 - tot_b / tot_h), len(rec), len(groups), bad))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(dump, open("gpurun_out/tune_w8.json", "w"))
