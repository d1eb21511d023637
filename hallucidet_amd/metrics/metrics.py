"""COCO-style mean average precision for the evaluation hooks (reference src/metrics/metrics.py:7-32 wraps
torchmetrics 0.6.0 `MAP` [EXT, not installable offline]; used at train_hallucidet.py:121-131,213-215,330-332,399-416 and
train_detector.py: `.update(detections, targets)`, `.compute()`, `.reset()`, result keys map / map_50 / map_75 /
map_small / map_medium / map_large / mar_1 / mar_10 / mar_100 / mar_small / mar_medium / mar_large / map_per_class /
mar_100_per_class).

This is host-side bookkeeping off the GPU hot path ("CPU-side"): numpy on the host, restating the
published COCO evaluation (pycocotools `evaluateImg` / `accumulate` / `summarize`, which torchmetrics' MAP re-implements):
IoU thresholds .50:.05:.95, 101 recall points, detections per image 1 / 10 / 100, area ranges all / small (<32^2) /
medium / large (>96^2), greedy matching in descending score order, precision envelope, AP = mean over the sampled
precision values.  PARITY UNPINNED against torchmetrics itself (absent); pinned by hand-worked cases in
tests/test_metrics.py.
"""
from typing import Dict, List

import numpy as np
import torch


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def box_iou_np(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """[A,4] x [B,4] xyxy -> [A,B]"""
    if a.size == 0 or b.size == 0:
        return np.zeros((a.shape[0], b.shape[0]), dtype=np.float64)
    a, b = a.astype(np.float64), b.astype(np.float64)
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b[None, :] - inter)


class MeanAveragePrecision:
    IOU_THRS = np.linspace(0.5, 0.95, 10)
    REC_THRS = np.linspace(0.0, 1.0, 101)
    MAX_DETS = (1, 10, 100)
    AREA_RNG = {"all": (0.0, 1e10), "small": (0.0, 32.0 ** 2), "medium": (32.0 ** 2, 96.0 ** 2), "large": (96.0 ** 2, 1e10)}

    def __init__(self, box_format: str = "xyxy", class_metrics: bool = False):
        if box_format != "xyxy":
            raise NotImplementedError("the reference only ever uses xyxy boxes (metrics.py:17)")
        self.class_metrics = class_metrics
        self.reset()

    def to(self, device):
        return self

    def reset(self):
        self._dets: List[Dict[str, np.ndarray]] = []
        self._gts: List[Dict[str, np.ndarray]] = []

    # ------------------------------------------------------------------ update
    def update(self, preds, target):
        preds, target = list(preds), list(target)
        if len(preds) != len(target):
            raise ValueError("Expected argument `preds` and `target` to have the same length")
        for p in preds:
            for k in ("boxes", "scores", "labels"):
                if k not in p:
                    raise ValueError(f"Expected all dicts in `preds` to contain the `{k}` key")
        for t in target:
            for k in ("boxes", "labels"):
                if k not in t:
                    raise ValueError(f"Expected all dicts in `target` to contain the `{k}` key")
        for p, t in zip(preds, target):
            self._dets.append({"boxes": _np(p["boxes"]).reshape(-1, 4).astype(np.float64), "scores": _np(p["scores"]).reshape(-1).astype(np.float64),
                               "labels": _np(p["labels"]).reshape(-1).astype(np.int64)})
            self._gts.append({"boxes": _np(t["boxes"]).reshape(-1, 4).astype(np.float64), "labels": _np(t["labels"]).reshape(-1).astype(np.int64)})

    # ------------------------------------------------------------------ per image / class / area evaluation
    def _evaluate_img(self, i, c, rng, max_det):
        g, d = self._gts[i], self._dets[i]
        gsel, dsel = g["labels"] == c, d["labels"] == c
        gb, db, ds = g["boxes"][gsel], d["boxes"][dsel], d["scores"][dsel]
        if gb.shape[0] == 0 and db.shape[0] == 0:
            return None
        garea = (gb[:, 2] - gb[:, 0]) * (gb[:, 3] - gb[:, 1])
        gig = (garea < rng[0]) | (garea > rng[1])
        gorder = np.argsort(gig, kind="mergesort")                 # non-ignored ground truth first
        gb, gig = gb[gorder], gig[gorder]
        dorder = np.argsort(-ds, kind="mergesort")[:max_det]
        db, ds = db[dorder], ds[dorder]
        ious = box_iou_np(db, gb)
        T, D, G = len(self.IOU_THRS), db.shape[0], gb.shape[0]
        gtm = np.zeros((T, G), dtype=bool)
        dtm = np.zeros((T, D), dtype=bool)
        dig = np.zeros((T, D), dtype=bool)
        for ti, t in enumerate(self.IOU_THRS):
            for di in range(D):
                best, m = min(t, 1 - 1e-10), -1
                for gi in range(G):
                    if gtm[ti, gi]:
                        continue
                    if m > -1 and not gig[m] and gig[gi]:
                        break                                     # matched a regular gt; only ignored ones follow
                    if ious[di, gi] < best:
                        continue
                    best, m = ious[di, gi], gi
                if m == -1:
                    continue
                dig[ti, di] = gig[m]
                dtm[ti, di] = True
                gtm[ti, m] = True
        darea = (db[:, 2] - db[:, 0]) * (db[:, 3] - db[:, 1])
        out = (darea < rng[0]) | (darea > rng[1])
        dig = dig | (~dtm & out[None, :])
        return {"scores": ds, "dtm": dtm, "dig": dig, "npos": int((~gig).sum())}

    def _accumulate(self, classes):
        T, R, K, A, M = len(self.IOU_THRS), len(self.REC_THRS), len(classes), len(self.AREA_RNG), len(self.MAX_DETS)
        precision = -np.ones((T, R, K, A, M))
        recall = -np.ones((T, K, A, M))
        n_img = len(self._gts)
        for ki, c in enumerate(classes):
            for ai, rng in enumerate(self.AREA_RNG.values()):
                for mi, md in enumerate(self.MAX_DETS):
                    ev = [e for e in (self._evaluate_img(i, c, rng, md) for i in range(n_img)) if e is not None]
                    if not ev:
                        continue
                    npig = sum(e["npos"] for e in ev)
                    if npig == 0:
                        continue
                    scores = np.concatenate([e["scores"] for e in ev])
                    order = np.argsort(-scores, kind="mergesort")
                    dtm = np.concatenate([e["dtm"] for e in ev], axis=1)[:, order]
                    dig = np.concatenate([e["dig"] for e in ev], axis=1)[:, order]
                    tps = np.cumsum(dtm & ~dig, axis=1).astype(np.float64)
                    fps = np.cumsum(~dtm & ~dig, axis=1).astype(np.float64)
                    for ti in range(T):
                        tp, fp = tps[ti], fps[ti]
                        nd = tp.shape[0]
                        rc = tp / npig
                        pr = tp / (fp + tp + np.spacing(1))
                        recall[ti, ki, ai, mi] = rc[-1] if nd else 0.0
                        pr = pr.tolist()
                        for j in range(nd - 1, 0, -1):
                            if pr[j] > pr[j - 1]:
                                pr[j - 1] = pr[j]
                        inds = np.searchsorted(rc, self.REC_THRS, side="left")
                        q = np.zeros(R)
                        for ri, pi in enumerate(inds):
                            if pi < nd:
                                q[ri] = pr[pi]
                        precision[ti, :, ki, ai, mi] = q
        return precision, recall

    @staticmethod
    def _mean(x):
        x = x[x > -1]
        return float(x.mean()) if x.size else -1.0

    def compute(self) -> Dict[str, torch.Tensor]:
        classes = sorted(set(np.concatenate([g["labels"] for g in self._gts] + [d["labels"] for d in self._dets]).tolist())) if self._gts else []
        precision, recall = self._accumulate(classes)
        areas = list(self.AREA_RNG)
        a_all, m100 = areas.index("all"), self.MAX_DETS.index(100)
        t50, t75 = 0, 5
        res = {
            "map": self._mean(precision[:, :, :, a_all, m100]),
            "map_50": self._mean(precision[t50, :, :, a_all, m100]),
            "map_75": self._mean(precision[t75, :, :, a_all, m100]),
            "map_small": self._mean(precision[:, :, :, areas.index("small"), m100]),
            "map_medium": self._mean(precision[:, :, :, areas.index("medium"), m100]),
            "map_large": self._mean(precision[:, :, :, areas.index("large"), m100]),
            "mar_1": self._mean(recall[:, :, a_all, self.MAX_DETS.index(1)]),
            "mar_10": self._mean(recall[:, :, a_all, self.MAX_DETS.index(10)]),
            "mar_100": self._mean(recall[:, :, a_all, m100]),
            "mar_small": self._mean(recall[:, :, areas.index("small"), m100]),
            "mar_medium": self._mean(recall[:, :, areas.index("medium"), m100]),
            "mar_large": self._mean(recall[:, :, areas.index("large"), m100]),
        }
        out = {k: torch.tensor(v, dtype=torch.float32) for k, v in res.items()}
        if self.class_metrics and classes:
            out["map_per_class"] = torch.tensor([self._mean(precision[:, :, k, a_all, m100]) for k in range(len(classes))], dtype=torch.float32)
            out["mar_100_per_class"] = torch.tensor([self._mean(recall[:, k, a_all, m100]) for k in range(len(classes))], dtype=torch.float32)
        else:
            out["map_per_class"] = torch.tensor(-1.0)
            out["mar_100_per_class"] = torch.tensor(-1.0)
        return out


MAP = MeanAveragePrecision


class Detection():
    """metrics.py:14-32"""

    def __init__(self, box_format='xyxy', device='cpu', class_metrics=False):
        self.device = device
        self.map = self.metric_map(box_format=box_format, class_metrics=class_metrics)

    def iou_bboxes(self, bbox1, bbox2):
        # the reference slices `[:, 3:]` of its 7-column rows (class, score, ?, x1, y1, x2, y2) and truncates to int
        a = torch.Tensor(bbox1)[:, 3:].int().numpy().astype(np.float64)
        b = torch.Tensor(bbox2)[:, 3:].int().numpy().astype(np.float64)
        return box_iou_np(a, b).astype(np.float32)

    def metric_map(self, box_format='xyxy', class_metrics=False):
        return MeanAveragePrecision(box_format=box_format, class_metrics=class_metrics).to(self.device)
