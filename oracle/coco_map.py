"""TEST INFRASTRUCTURE ONLY -- the COCO detection metric restated a second time, in plain Python loops, to cross-check
hallucidet_amd/metrics/metrics.py (reference src/metrics/metrics.py:7-32 wraps torchmetrics' MAP [EXT], which re-implements
pycocotools' COCOeval; neither is installable here: PARITY UNPINNED against them, this file follows the published COCOeval
procedure: evaluateImg -> accumulate -> summarize).

Written differently on purpose (no shared helpers, scalar loops, precision-at-recall taken as the maximum over the suffix of the
ranked list instead of an in-place envelope + searchsorted) so that an agreement between the two is evidence, not an identity."""

# COCOeval's grids are np.linspace(.5, .95, 10) and np.linspace(0, 1, 101): start + i * step with the end point set exactly.  The
# recall grid matters to the last bit: recalls k/n land ON grid points (0.29 vs 29 * 0.01 = 0.29000000000000004 decides a sample).
IOU_THRS = [0.5 + i * ((0.95 - 0.5) / 9) for i in range(9)] + [0.95]
REC_THRS = [i * (1.0 / 100) for i in range(100)] + [1.0]
AREAS = {"all": (0.0, 1e10), "small": (0.0, 32.0 ** 2), "medium": (32.0 ** 2, 96.0 ** 2), "large": (96.0 ** 2, 1e10)}


def _iou(a, b):
    iw = min(a[2], b[2]) - max(a[0], b[0])
    ih = min(a[3], b[3]) - max(a[1], b[1])
    if iw <= 0 or ih <= 0:
        return 0.0
    inter = iw * ih
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)


def _area(b):
    return (b[2] - b[0]) * (b[3] - b[1])


def _match_image(dets, gts, thr, rng, max_det):
    """dets: [(score, box)] of ONE class in ONE image; gts: [box].  Returns [(score, is_tp, ignored)] and #countable gts."""
    dets = sorted(enumerate(dets), key=lambda e: (-e[1][0], e[0]))[:max_det]          # stable descending score
    g_ign = [not (rng[0] <= _area(g) <= rng[1]) for g in gts]
    order = sorted(range(len(gts)), key=lambda i: (g_ign[i], i))                       # countable ground truth first
    taken = [False] * len(gts)
    out = []
    for _, (score, box) in dets:
        best, m = min(thr, 1 - 1e-10), -1
        for gi in order:
            if taken[gi]:
                continue
            if m >= 0 and not g_ign[m] and g_ign[gi]:
                break
            v = _iou(box, gts[gi])
            if v < best:
                continue
            best, m = v, gi
        if m >= 0:
            taken[m] = True
            out.append((score, True, g_ign[m]))
        else:
            out.append((score, False, not (rng[0] <= _area(box) <= rng[1])))
    return out, sum(1 for f in g_ign if not f)


def evaluate(preds, targets):
    """preds: [{'boxes': [[x1,y1,x2,y2]], 'scores': [..], 'labels': [..]}] per image (python lists); targets likewise without scores.
    Returns the twelve COCO summary numbers."""
    classes = sorted({int(l) for p in preds for l in p["labels"]} | {int(l) for t in targets for l in t["labels"]})

    def curve(c, thr, rng, max_det):
        """(AP over the 101 recall points, final recall) for one class / IoU threshold / area range / detection cap, or None."""
        ranked, npos, seen = [], 0, False
        for p, t in zip(preds, targets):
            d = [(float(s), b) for s, b, l in zip(p["scores"], p["boxes"], p["labels"]) if int(l) == c]
            g = [b for b, l in zip(t["boxes"], t["labels"]) if int(l) == c]
            if not d and not g:
                continue
            seen = True
            r, n = _match_image(d, g, thr, rng, max_det)
            base = len(ranked)
            ranked += [(s, base + i, tp, ig) for i, (s, tp, ig) in enumerate(r)]
            npos += n
        if not seen or npos == 0:
            return None
        ranked.sort(key=lambda e: (-e[0], e[1]))
        pts, tp, fp = [], 0, 0
        for _, _, is_tp, ig in ranked:
            if not ig:
                tp += int(is_tp)
                fp += int(not is_tp)
            pts.append((tp / npos, tp / (tp + fp + 2.220446049250313e-16)))
        ap = 0.0
        for r in REC_THRS:
            best = 0.0
            for rc, pr in pts:                      # precision at recall >= r: the best precision from the first point reaching r on
                if rc >= r and pr > best:
                    best = pr
            ap += best
        return ap / len(REC_THRS), (pts[-1][0] if pts else 0.0)

    def mean(vals):
        vals = [v for v in vals if v is not None]
        return sum(vals) / len(vals) if vals else -1.0

    def ap(thrs, area, md=100):
        return mean([(lambda r: None if r is None else r[0])(curve(c, t, AREAS[area], md)) for t in thrs for c in classes])

    def ar(area, md):
        return mean([(lambda r: None if r is None else r[1])(curve(c, t, AREAS[area], md)) for t in IOU_THRS for c in classes])

    return {"map": ap(IOU_THRS, "all"), "map_50": ap([IOU_THRS[0]], "all"), "map_75": ap([IOU_THRS[5]], "all"),
            "map_small": ap(IOU_THRS, "small"), "map_medium": ap(IOU_THRS, "medium"), "map_large": ap(IOU_THRS, "large"),
            "mar_1": ar("all", 1), "mar_10": ar("all", 10), "mar_100": ar("all", 100),
            "mar_small": ar("small", 100), "mar_medium": ar("medium", 100), "mar_large": ar("large", 100)}
