"""Pins for the torchvision-side INTEGER work of the detector half (NMS keep sets, the `batched_nms` offset trick, `filter_proposals`,
the balanced sampler) that do not rest on this repository's own restatement (oracle/):

  (a) the installed `transformers` wheel ships a plain-PyTorch greedy NMS loop and `box_iou` written outside this repository
      (transformers/models/owlvit/image_processing_pil_owlvit.py: `post_process_image_guided_detection`, the same IoU formula and the
      same strict `>` rule as torchvision's `nms`): it and `oracle.kernels.nms_sorted` are driven with the same boxes -- coarse-grid
      coordinates (ties, duplicates, threshold-exact IoUs) -- and must keep the same sets;
  (b) torchvision's `batched_nms` coordinate-offset trick re-stated on top of THAT loop must equal `oracle.detection.batched_nms`
      (both of its branches);
  (c) `RegionProposalNetwork.filter_proposals` and `BalancedPositiveNegativeSampler` against brute-force definitions written here
      (python loops over numpy float32 scalars; nothing imported from oracle/ on the checking side), on `hypothesis`-generated inputs.

torchvision itself cannot be installed in this environment (no network); these are the nearest externally held statements.
CPU only."""
import math
import types

import numpy as np
import pytest
import torch

from oracle import detection as od
from oracle import kernels as ok

hypothesis = pytest.importorskip("hypothesis")
from hypothesis import given, settings, strategies as st  # noqa: E402


@pytest.fixture
def no_torchvision_stub():
    """Other tests leave stand-in `torchvision*` modules (no __spec__) in sys.modules; transformers probes for torchvision with
    importlib.util.find_spec, which raises on those.  Hidden for the duration of a test, restored afterwards."""
    import sys
    hidden = {k: sys.modules.pop(k) for k in list(sys.modules)
              if (k == "torchvision" or k.startswith("torchvision.")) and getattr(sys.modules[k], "__spec__", None) is None}
    try:
        yield
    finally:
        sys.modules.update(hidden)


def _owlvit():
    return pytest.importorskip("transformers.models.owlvit.image_processing_pil_owlvit")


def owlvit_nms(boxes, logits, thr):
    """Keep set (ascending candidate index) of transformers' greedy NMS loop.  boxes [n, 4] corner format on a 1/4 grid below 2^12 (the
    loop takes centre format and converts back: exact on such coordinates), logits [n] whose sigmoids are distinct and within a factor
    ten of each other (the method reports survivors through display alphas that vanish below a tenth of the best score)."""
    m = _owlvit()
    n = boxes.shape[0]
    cxcywh = torch.stack([(boxes[:, 0] + boxes[:, 2]) / 2, (boxes[:, 1] + boxes[:, 3]) / 2, boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], dim=1)
    back = torch.stack([cxcywh[:, 0] - 0.5 * cxcywh[:, 2], cxcywh[:, 1] - 0.5 * cxcywh[:, 3], cxcywh[:, 0] + 0.5 * cxcywh[:, 2], cxcywh[:, 1] + 0.5 * cxcywh[:, 3]], dim=1)
    assert torch.equal(back, boxes), "coordinates must survive the centre-format round trip exactly"
    out = types.SimpleNamespace(logits=logits.reshape(1, n, 1).clone(), target_pred_boxes=cxcywh.reshape(1, n, 4).clone())
    res = m.OwlViTImageProcessorPil.post_process_image_guided_detection(None, out, threshold=0.0, nms_threshold=thr, target_sizes=None)
    kept_boxes, alphas = res[0]["boxes"], res[0]["scores"]
    # the method returns the survivors in candidate order with alpha = clip((s - 0.1 max) / (0.9 max)); recover their indices from the
    # alphas (strictly increasing in s, all s distinct)
    s = torch.sigmoid(logits)
    mx = s.max() + 1e-6
    alpha_all = torch.clip((s - mx * 0.1) / (mx * 0.9), 0.0, 1.0)
    assert float(alpha_all.min()) > 0 and alpha_all.unique().numel() == n
    idx = []
    for a, b in zip(alphas.tolist(), kept_boxes):
        j = (alpha_all == a).nonzero().flatten()
        assert j.numel() == 1 and torch.equal(boxes[j[0]], b)
        idx.append(int(j[0]))
    assert idx == sorted(idx)
    return idx


def grid_boxes(n, g, extent=64, dup=0.15):
    """n boxes on a HALF-integer grid inside [0, extent]: many exact ties of coordinates, exact duplicates, nested boxes and IoUs that
    land exactly on simple fractions (1/2, 1/3, 2/3 ...)."""
    xy = torch.randint(0, 2 * extent - 8, (n, 2), generator=g).float() / 2
    wh = torch.randint(2, 24, (n, 2), generator=g).float() / 2
    b = torch.cat([xy, xy + wh], dim=1)
    for i in range(1, n):
        if float(torch.rand(1, generator=g)) < dup:
            b[i] = b[int(torch.randint(0, i, (1,), generator=g))]
    return b


def distinct_logits(n, g):
    s = 0.3 + 0.65 * (torch.randperm(n, generator=g).float() + 0.5) / n          # distinct, within [0.3, 0.95]
    lg = torch.log(s / (1 - s))
    assert torch.sigmoid(lg).unique().numel() == n
    return lg


@pytest.mark.parametrize("thr", [0.3, 0.5, 0.7, 1.0 / 3.0])
def test_nms_keep_sets_equal_the_greedy_loop_of_transformers(thr, no_torchvision_stub):
    g = torch.Generator().manual_seed(int(thr * 1000))
    hit_exact = 0
    for trial in range(6):
        n = 120
        boxes = grid_boxes(n, g)
        for k in range(6):                                    # pairs whose IoU is EXACTLY 1/2 and exactly 1/3 (the rule is a strict >)
            x, y = float(torch.randint(0, 40, (1,), generator=g)), float(torch.randint(0, 40, (1,), generator=g))
            boxes[4 * k] = torch.tensor([x, y, x + 4, y + 2]); boxes[4 * k + 1] = torch.tensor([x, y, x + 4, y + 1])
            boxes[4 * k + 2] = torch.tensor([x + 9, y, x + 11, y + 2]); boxes[4 * k + 3] = torch.tensor([x + 10, y, x + 12, y + 2])
        logits = distinct_logits(n, g)
        scores = torch.sigmoid(logits)
        iou = _owlvit().box_iou(boxes, boxes)[0]
        assert torch.equal(iou, ok.box_iou(boxes, boxes))                 # same formula, same association: bit for bit
        hit_exact += int(((iou == thr) & ~torch.eye(n, dtype=torch.bool)).sum())
        theirs = owlvit_nms(boxes, logits, thr)
        ours = sorted(od.nms(boxes, scores, thr).tolist())
        assert ours == theirs, (trial, sorted(set(ours) ^ set(theirs)))
        # and through the sorted-order kernel statement directly
        order = torch.sort(scores, descending=True, stable=True)[1]
        assert sorted(order[ok.nms_sorted(boxes[order], thr)].tolist()) == theirs
    if thr == 0.5:
        assert hit_exact > 0, "the generator is meant to produce IoUs exactly on the threshold"


def batched_nms_on_their_loop(boxes, logits, idxs, thr):
    """torchvision.ops.boxes._batched_nms_coordinate_trick restated over the NMS loop of transformers: boxes of category c are moved by
    c * (max coordinate + 1), one NMS over everything, survivors in descending score order."""
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    keep = owlvit_nms(boxes + offsets[:, None], logits, thr)
    s = torch.sigmoid(logits)
    return sorted(keep, key=lambda i: -float(s[i]))


@pytest.mark.parametrize("n,levels", [(150, 5), (90, 2), (1100, 5)])
def test_batched_nms_offset_trick_equals_the_restated_form(n, levels, no_torchvision_stub, monkeypatch):
    """(1100 boxes = 4 400 coordinates: above torchvision's CPU switch-over to the per-category loop, which oracle.detection follows
    when NMS_VANILLA_NUMEL is set to the CPU value -- both branches must give the one answer.)"""
    g = torch.Generator().manual_seed(n + levels)
    boxes = grid_boxes(n, g, extent=128)
    logits = distinct_logits(n, g)
    idxs = torch.randint(0, levels, (n,), generator=g)
    scores = torch.sigmoid(logits)
    want = batched_nms_on_their_loop(boxes, logits, idxs, 0.7)
    for numel in (20000, 4000):
        monkeypatch.setattr(od, "NMS_VANILLA_NUMEL", numel)
        got = od.batched_nms(boxes, scores, idxs, 0.7).tolist()
        assert got == want, (numel, len(got), len(want))
    # categories never interact: the same as one NMS per category
    per = []
    for c in range(levels):
        cur = (idxs == c).nonzero().flatten()
        if cur.numel():
            per += cur[torch.tensor(sorted(od.nms(boxes[cur], scores[cur], 0.7).tolist()), dtype=torch.int64)].tolist()
    assert sorted(per) == sorted(want)


# ----------------------------------------------------------------------------- (c) brute-force definitions
F32 = np.float32


def iou_f32(a, b):
    """torchvision's IoU on float32 scalars: inter / (area_a + area_b - inter)."""
    aa = F32(F32(a[2] - a[0]) * F32(a[3] - a[1]))
    ab = F32(F32(b[2] - b[0]) * F32(b[3] - b[1]))
    w = max(F32(0), F32(min(a[2], b[2]) - max(a[0], b[0])))
    h = max(F32(0), F32(min(a[3], b[3]) - max(a[1], b[1])))
    inter = F32(w * h)
    return F32(inter / F32(F32(aa + ab) - inter))


def brute_filter_proposals(proposals, objectness, image_shape, per_level, pre_n, post_n, nms_thr, score_thr, min_size):
    """All images.  proposals [n_img, A, 4] float32 numpy, objectness [n_img, A]; per image and level: the pre_n best objectness (equal
    values: lower index first), clip to the image, drop boxes with a side below min_size, drop scores below score_thr, greedy NMS inside
    the level; all levels' survivors by descending score (equal scores: position in the level-major candidate list), the first post_n.
    The scores are ATen's sigmoid of the SELECTED objectness values as one [n_img, K] tensor -- the shape torchvision applies it to (the
    vectorised sigmoid's last bit depends on an element's position in the tensor)."""
    h, w = image_shape
    n_img = proposals.shape[0]
    cands = []
    for n in range(n_img):
        cand, off = [], 0
        for lvl, cnt in enumerate(per_level):
            idx = sorted(range(off, off + cnt), key=lambda i: (-float(objectness[n][i]), i))[:min(pre_n, cnt)]
            cand += [(lvl, i) for i in idx]
            off += cnt
        cands.append(cand)
    sel = torch.tensor([[float(objectness[n][i]) for (_, i) in cands[n]] for n in range(n_img)], dtype=torch.float32)
    prob = torch.sigmoid(sel).numpy()
    out = []
    for n in range(n_img):
        items = []
        for pos, (lvl, i) in enumerate(cands[n]):
            b = proposals[n][i].astype(F32).copy()
            b[0] = min(max(b[0], F32(0)), F32(w)); b[2] = min(max(b[2], F32(0)), F32(w))
            b[1] = min(max(b[1], F32(0)), F32(h)); b[3] = min(max(b[3], F32(0)), F32(h))
            s = prob[n][pos]
            if F32(b[2] - b[0]) >= F32(min_size) and F32(b[3] - b[1]) >= F32(min_size) and s >= F32(score_thr):
                items.append((pos, lvl, b, s))
        survivors = []
        for lvl in range(len(per_level)):
            mine = sorted([it for it in items if it[1] == lvl], key=lambda it: (-float(it[3]), it[0]))
            kept = []
            for it in mine:
                if all(not (iou_f32(k[2], it[2]) > F32(nms_thr)) for k in kept):
                    kept.append(it)
            survivors += kept
        survivors.sort(key=lambda it: (-float(it[3]), it[0]))
        survivors = survivors[:post_n]
        out.append((np.array([it[2] for it in survivors], dtype=F32).reshape(-1, 4), np.array([it[3] for it in survivors], dtype=F32)))
    return out


@settings(max_examples=60, deadline=None)
@given(seed=st.integers(0, 10 ** 6), pre_n=st.integers(1, 9), post_n=st.integers(1, 12), n_img=st.integers(1, 2),
       coarse=st.booleans())
def test_filter_proposals_against_a_brute_force_definition(seed, pre_n, post_n, n_img, coarse):
    g = torch.Generator().manual_seed(seed)
    per_level = [12, 7, 3]
    A = sum(per_level)
    H, W = 40, 48
    xy = torch.rand(n_img, A, 2, generator=g) * 56 - 8                     # some boxes start outside the image
    wh = torch.rand(n_img, A, 2, generator=g) * 30
    wh[torch.rand(n_img, A, generator=g) < 0.15] = 0.0                     # degenerate boxes: removed by the min-size rule
    props = torch.cat([xy, xy + wh], dim=2)
    if coarse:
        props = torch.round(props * 2) / 2                                 # coordinate ties and exact duplicates
    obj = torch.randn(n_img, A, generator=g)
    if coarse:
        obj = torch.round(obj * 2) / 2                                     # many equal scores: the tie rules are exercised
    rpn = od.RegionProposalNetwork().eval()
    rpn._pre_nms_top_n = dict(training=pre_n, testing=pre_n)
    rpn._post_nms_top_n = dict(training=post_n, testing=post_n)
    rpn.min_size = 1.0
    fb, fs = rpn.filter_proposals(props.clone(), obj.clone().reshape(n_img, A, 1), [(H, W)] * n_img, per_level)
    want = brute_filter_proposals(props.numpy(), obj.numpy(), (H, W), per_level, pre_n, post_n, rpn.nms_thresh, rpn.score_thresh, rpn.min_size)
    for n in range(n_img):
        wb, ws = want[n]
        assert fb[n].shape[0] == wb.shape[0] <= post_n
        assert np.array_equal(fb[n].numpy(), wb), (fb[n], wb)
        assert np.array_equal(fs[n].numpy(), ws)
        # properties that hold whatever the arithmetic: inside the image, descending scores, big enough
        if wb.shape[0]:
            b = fb[n]
            assert float(b[:, 0::2].min()) >= 0 and float(b[:, 0::2].max()) <= W and float(b[:, 1::2].min()) >= 0 and float(b[:, 1::2].max()) <= H
            assert bool((fs[n][:-1] >= fs[n][1:]).all())
            assert bool(((b[:, 2] - b[:, 0]) >= 1.0).all() and ((b[:, 3] - b[:, 1]) >= 1.0).all())


@settings(max_examples=80, deadline=None)
@given(labels=st.lists(st.lists(st.sampled_from([-1, 0, 0, 0, 1, 2]), min_size=1, max_size=60), min_size=1, max_size=3),
       batch=st.integers(1, 32), frac=st.sampled_from([0.25, 0.5]), seed=st.integers(0, 10 ** 6))
def test_balanced_sampler_against_its_definition(labels, batch, frac, seed):
    """BalancedPositiveNegativeSampler: at most batch * frac positives (label >= 1), the rest of the batch from the negatives
    (label == 0), ignored entries (label < 0) never drawn, the two masks disjoint; the drawn members are the leading entries of the
    injected permutations (positives first, then negatives -- the order torchvision consumes its generator in)."""
    g = torch.Generator().manual_seed(seed)
    perms = []

    def perm(n):
        p = torch.randperm(n, generator=g)
        perms.append(p)
        return p
    sampler = od.BalancedPositiveNegativeSampler(batch, frac, perm)
    ms = [torch.tensor(l, dtype=torch.int64) for l in labels]
    pos, neg = sampler(ms)
    assert len(perms) == 2 * len(ms)
    for i, m in enumerate(ms):
        P = [j for j, v in enumerate(m.tolist()) if v >= 1]
        N = [j for j, v in enumerate(m.tolist()) if v == 0]
        n_pos = min(len(P), int(batch * frac))
        n_neg = min(len(N), batch - n_pos)
        pi, ni = pos[i].nonzero().flatten().tolist(), neg[i].nonzero().flatten().tolist()
        assert len(pi) == n_pos and len(ni) == n_neg
        assert set(pi) <= set(P) and set(ni) <= set(N) and not (set(pi) & set(ni))
        assert sorted(pi) == sorted(P[k] for k in perms[2 * i][:n_pos].tolist())
        assert sorted(ni) == sorted(N[k] for k in perms[2 * i + 1][:n_neg].tolist())
        assert pos[i].dtype == torch.uint8 and pos[i].shape == m.shape


# ----------------------------------------------------------------------------------------------------------------------------------
# Round 5: the last torchvision pieces the reference reaches at src/utils/eval_forward_fasterrcnn.py:77,122,136 -- the anchor grid,
# the pyramid-level mapper of MultiScaleRoIAlign and RoIHeads.postprocess_detections -- against definitions written HERE from
# torchvision 0.12's published algorithm (python loops over numpy float32 scalars; nothing from oracle/ on the checking side).


def brute_postprocess(boxes_rc, scores_rc, image_hw, score_thr, nms_thr, min_size, top_n):
    """One image.  boxes_rc [R, C, 4] float32 (decoded, NOT yet clipped), scores_rc [R, C] float32 (softmax), class 0 = background.
    torchvision: clip to the image, drop the background column, flatten (proposal-major, class-minor), keep score > score_thr, keep
    both sides >= min_size, greedy NMS among the boxes of ONE class (IoU > nms_thr suppresses; higher score first, equal scores:
    lower flat index first), all classes' survivors by descending score, the first top_n."""
    h, w = image_hw
    R, C = scores_rc.shape
    items = []
    for r in range(R):
        for c in range(1, C):
            b = boxes_rc[r, c].astype(F32).copy()
            b[0] = min(max(b[0], F32(0)), F32(w)); b[2] = min(max(b[2], F32(0)), F32(w))
            b[1] = min(max(b[1], F32(0)), F32(h)); b[3] = min(max(b[3], F32(0)), F32(h))
            s = F32(scores_rc[r, c])
            flat = r * (C - 1) + (c - 1)
            if s > F32(score_thr) and F32(b[2] - b[0]) >= F32(min_size) and F32(b[3] - b[1]) >= F32(min_size):
                items.append((flat, c, b, s))
    survivors = []
    for c in range(1, C):
        mine = sorted([it for it in items if it[1] == c], key=lambda it: (-float(it[3]), it[0]))
        kept = []
        for it in mine:
            if all(not (iou_f32(k[2], it[2]) > F32(nms_thr)) for k in kept):
                kept.append(it)
        survivors += kept
    survivors.sort(key=lambda it: (-float(it[3]), it[0]))
    survivors = survivors[:top_n]
    return (np.array([it[2] for it in survivors], dtype=F32).reshape(-1, 4), np.array([it[3] for it in survivors], dtype=F32),
            np.array([it[1] for it in survivors], dtype=np.int64))


@settings(max_examples=60, deadline=None)
@given(seed=st.integers(0, 10 ** 6), n_cls=st.integers(2, 4), top_n=st.integers(1, 12), n_img=st.integers(1, 2), spread=st.sampled_from([1.0, 4.0]))
def test_postprocess_detections_against_a_brute_force_definition(seed, n_cls, top_n, n_img, spread):
    """RoIHeads.postprocess_detections (torchvision 0.12 roi_heads.py; reached from eval_forward_fasterrcnn.py:122-136): per-class split,
    score and size filters, class-wise NMS through the coordinate-offset trick, top-k.  Proposals on a half-integer grid and
    regression codes whose decode is exact in fp32 (dx, dy multiples of 1/8 of the box size after the (10, 10, 5, 5) weights,
    dw = dh = 0), so the decoded boxes can be written down here without restating BoxCoder; scores are ATen's softmax."""
    g = torch.Generator().manual_seed(seed)
    H, W = 40, 48
    R = 14
    props, logits, codes, want_boxes = [], [], [], []
    for _ in range(n_img):
        xy = torch.randint(-8, 2 * W - 8, (R, 2), generator=g).float() / 2          # some start left / above the image
        wh = torch.randint(0, 40, (R, 2), generator=g).float() / 2                  # zero-size boxes: the min-size rule
        p = torch.cat([xy, xy + wh], dim=1)
        for i in range(1, R):
            if float(torch.rand(1, generator=g)) < 0.3:
                p[i] = p[int(torch.randint(0, i, (1,), generator=g))]                # duplicates: suppressed inside a class only
        j = torch.randint(-2, 3, (R, n_cls, 2), generator=g).float()                 # per-class shift in eighths of the box size
        code = torch.zeros(R, n_cls, 4)
        code[..., 0] = 1.25 * j[..., 0]                                             # / 10 -> j / 8 exactly
        code[..., 1] = 1.25 * j[..., 1]
        wd, ht = (p[:, 2] - p[:, 0]), (p[:, 3] - p[:, 1])
        cx, cy = p[:, 0] + 0.5 * wd, p[:, 1] + 0.5 * ht
        pcx, pcy = (j[..., 0] / 8) * wd[:, None] + cx[:, None], (j[..., 1] / 8) * ht[:, None] + cy[:, None]
        hw_, hh_ = 0.5 * wd[:, None].expand(R, n_cls), 0.5 * ht[:, None].expand(R, n_cls)
        want_boxes.append(torch.stack([pcx - hw_, pcy - hh_, pcx + hw_, pcy + hh_], dim=2))
        props.append(p)
        codes.append(code.reshape(R, n_cls * 4))
        logits.append(torch.randn(R, n_cls, generator=g) * spread)
    heads = od.RoIHeads(num_classes=n_cls)
    heads.detections_per_img = top_n
    cl, br = torch.cat(logits), torch.cat(codes)
    gb, gs, gl = heads.postprocess_detections(cl, br, props, [(H, W)] * n_img)
    scores_all = torch.softmax(cl, -1).split([R] * n_img, 0)
    for n in range(n_img):
        wb, ws, wl = brute_postprocess(want_boxes[n].numpy(), scores_all[n].numpy(), (H, W), heads.score_thresh, heads.nms_thresh, 1e-2, top_n)
        assert gb[n].shape[0] == wb.shape[0] <= top_n
        assert np.array_equal(gb[n].numpy(), wb), (gb[n], wb)
        assert np.array_equal(gs[n].numpy(), ws)
        assert np.array_equal(gl[n].numpy(), wl)
        if wb.shape[0]:
            assert bool((gs[n][:-1] >= gs[n][1:]).all()) and int(gl[n].min()) >= 1 and float(gs[n].min()) > heads.score_thresh


def brute_level(box, k_min=2, k_max=5):
    """torchvision.ops.poolers.LevelMapper (0.12): floor(4 + log2(sqrt(area) / 224) + 1e-6) clamped to [k_min, k_max], minus k_min.
    Evaluated in float64 from the float32 area; returns None when the value sits within 1e-4 of a level boundary without being an
    exact power-of-two ratio (there fp32's log2 and this float64 one may land on different sides)."""
    area = F32(F32(box[2] - box[0]) * F32(box[3] - box[1]))
    s = float(np.sqrt(area))                       # IEEE sqrt: correctly rounded in fp32, then widened
    if s == 0.0:
        return 0                                   # log2(0) = -inf -> floor -> clamp to k_min
    t = 4.0 + math.log2(s / 224.0) + 1e-6
    frac = t - math.floor(t)
    exact = math.log2(s / 224.0) == round(math.log2(s / 224.0))
    if not exact and (frac < 1e-4 or frac > 1 - 1e-4):
        return None
    return int(min(max(math.floor(t), k_min), k_max)) - k_min


@settings(max_examples=200, deadline=None)
@given(boxes=st.lists(st.tuples(st.floats(0, 280, width=32), st.floats(0, 280, width=32), st.floats(0, 600, width=32), st.floats(0, 600, width=32)),
                      min_size=1, max_size=12))
def test_level_mapper_against_its_definition(boxes):
    arr = np.array([[x, y, x + w, y + h] for x, y, w, h in boxes], dtype=F32)
    want = [brute_level(b) for b in arr]
    got = od.MultiScaleRoIAlign().level_map([torch.from_numpy(arr)], 2, 5).tolist()
    for g_, w_ in zip(got, want):
        assert w_ is None or g_ == w_


def test_level_mapper_at_the_exact_level_boundaries():
    """sqrt(area) = 224 * 2^k exactly (k = -3 .. 2): the 1e-6 keeps the value ON the upper level; one fp32 step below the boundary
    stays on the lower level; the clamp takes everything below 112 to level 0 and everything from 896 up to level 3."""
    sides = [28.0, 56.0, 112.0, 224.0, 448.0, 896.0]
    arr = np.array([[3.0, 5.0, 3.0 + s, 5.0 + s] for s in sides], dtype=F32)
    got = od.MultiScaleRoIAlign().level_map([torch.from_numpy(arr)], 2, 5).tolist()
    assert got == [0, 0, 1, 2, 3, 3]
    below = np.array([[0.0, 0.0, s, np.nextafter(F32(s), F32(0))] for s in (112.0, 224.0, 448.0)], dtype=F32)
    # (area shrinks by one part in 2^23 of one side: log2 moves by ~1e-7 * 1.44 / 2 < 1e-6, so the eps lifts it back over the boundary)
    assert od.MultiScaleRoIAlign().level_map([torch.from_numpy(below)], 2, 5).tolist() == [1, 2, 3]
    clearly_below = np.array([[0.0, 0.0, s, s * 0.999] for s in (112.0, 224.0, 448.0)], dtype=F32)
    assert od.MultiScaleRoIAlign().level_map([torch.from_numpy(clearly_below)], 2, 5).tolist() == [0, 1, 2]
    two_lists = od.MultiScaleRoIAlign().level_map([torch.from_numpy(arr[:2]), torch.from_numpy(arr[2:])], 2, 5).tolist()
    assert two_lists == got                        # the mapper concatenates the per-image lists in order


def brute_anchors(image_hw, grids, sizes, ratios):
    """torchvision AnchorGenerator (0.12 anchor_utils.py): base anchors round([-w, -h, w, h] / 2) with h = size * sqrt(ratio),
    w = size / sqrt(ratio) in fp32 (ratio-major, size-minor), shifted by (x * stride_w, y * stride_h) with stride = image // grid
    (integer division), locations row-major, anchors of a location contiguous, levels concatenated."""
    ih, iw = image_hw
    out = []
    for (gh, gw), size, rs in zip(grids, sizes, ratios):
        sh, sw = ih // gh, iw // gw
        base = []
        for r in rs:
            hr = np.sqrt(F32(r))
            wr = F32(1) / hr
            for sz in size:
                w_, h_ = F32(wr * F32(sz)), F32(hr * F32(sz))
                base.append([np.round(F32(-w_) / F32(2)), np.round(F32(-h_) / F32(2)), np.round(w_ / F32(2)), np.round(h_ / F32(2))])
        for y in range(gh):
            for x in range(gw):
                for b in base:
                    out.append([F32(x * sw) + b[0], F32(y * sh) + b[1], F32(x * sw) + b[2], F32(y * sh) + b[3]])
    return np.array(out, dtype=F32)


def test_anchor_generator_at_the_llvip_grid_sizes():
    """The five pyramid levels of the detector on the 300 x 300 transformed image (75, 38, 19, 10, 5: strides 4, 7, 15, 30, 60 by
    integer division) and on a non-square image, against the loop above; plus the base anchors every torchvision user has seen
    printed for size 32, ratios (0.5, 1, 2)."""
    ag = od.AnchorGenerator()
    sizes, ratios = ((32,), (64,), (128,), (256,), (512,)), ((0.5, 1.0, 2.0),) * 5
    assert ag.base_anchors((32,), (0.5, 1.0, 2.0)).tolist() == [[-23.0, -11.0, 23.0, 11.0], [-16.0, -16.0, 16.0, 16.0], [-11.0, -23.0, 11.0, 23.0]]
    assert ag.num_anchors_per_location() == [3] * 5
    for (ih, iw), grids in (((300, 300), [(75, 75), (38, 38), (19, 19), (10, 10), (5, 5)]),
                            ((512, 640), [(128, 160), (64, 80), (32, 40), (16, 20), (8, 10)]),
                            ((300, 200), [(75, 50), (38, 25), (19, 13), (10, 7), (5, 4)])):
        il = od.ImageList(torch.zeros(2, 3, ih, iw), [(ih, iw)] * 2)
        feats = [torch.zeros(2, 1, gh, gw) for gh, gw in grids]
        got = ag(il, feats)
        want = brute_anchors((ih, iw), grids, sizes, ratios)
        assert len(got) == 2 and got[0].dtype == torch.float32 and got[0].shape == (sum(gh * gw * 3 for gh, gw in grids), 4)
        assert np.array_equal(got[0].numpy(), want) and torch.equal(got[0], got[1])
    # RetinaNet's generator: three octave scales per level, ratio-major order (retinanet.py: anchor_sizes x (2^0, 2^(1/3), 2^(2/3)))
    rs = tuple((x, int(x * 2 ** (1.0 / 3)), int(x * 2 ** (2.0 / 3))) for x in (32, 64, 128, 256, 512))
    ag2 = od.AnchorGenerator(rs, ratios)
    grids = [(38, 38), (19, 19), (10, 10), (5, 5), (3, 3)]
    il = od.ImageList(torch.zeros(1, 3, 300, 300), [(300, 300)])
    got = ag2(il, [torch.zeros(1, 1, gh, gw) for gh, gw in grids])
    assert np.array_equal(got[0].numpy(), brute_anchors((300, 300), grids, rs, ratios))


# ----------------------------------------------------------------------------------------------------------------------------------
# Round 5, second batch: the REDUCTION FORMS of the two Faster R-CNN losses, FCOS's centre-sampling assignment and the P6 / P7 block,
# again from torchvision 0.12's published code as python loops in float64 (roi_heads.py `fastrcnn_loss`, rpn.py `compute_loss`,
# fcos.py `FCOS.forward` matching block, feature_pyramid_network.py `LastLevelP6P7`).


def _smooth_l1(d, beta):
    d = abs(d)
    return 0.5 * d * d / beta if d < beta else d - 0.5 * beta


@settings(max_examples=60, deadline=None)
@given(seed=st.integers(0, 10 ** 6), n_cls=st.integers(2, 5), rows=st.lists(st.integers(0, 9), min_size=1, max_size=3))
def test_fastrcnn_loss_reduction_form(seed, n_cls, rows):
    """classification: mean over ALL sampled rows of -log softmax(logits)[label]; box: smooth-L1 (beta 1/9) of the 4 deltas of the
    LABEL's class, summed over the foreground rows only, divided by the number of ALL rows."""
    g = torch.Generator().manual_seed(seed)
    if sum(rows) == 0:
        rows = [1]
    N = sum(rows)
    logits = torch.randn(N, n_cls, generator=g) * 2
    reg = torch.randn(N, n_cls * 4, generator=g)
    labels = [torch.randint(0, n_cls, (r,), generator=g) for r in rows]
    tgt = [torch.randn(r, 4, generator=g) * 0.3 for r in rows]
    got_c, got_b = od.fastrcnn_loss(logits, reg, labels, tgt)
    lab = [int(v) for l in labels for v in l.tolist()]
    t = [row for tt in tgt for row in tt.tolist()]
    lc = lb = 0.0
    for i in range(N):
        z = [float(v) for v in logits[i].tolist()]
        m = max(z)
        lse = m + math.log(sum(math.exp(v - m) for v in z))
        lc += lse - z[lab[i]]
        if lab[i] > 0:
            for k in range(4):
                lb += _smooth_l1(float(reg[i, 4 * lab[i] + k]) - t[i][k], 1.0 / 9)
    assert abs(float(got_c) - lc / N) <= 2e-6 * max(1.0, abs(lc / N))
    assert abs(float(got_b) - lb / N) <= 2e-6 * max(1.0, abs(lb / N))


@settings(max_examples=60, deadline=None)
@given(seed=st.integers(0, 10 ** 6), n_img=st.integers(1, 3), batch=st.integers(2, 12))
def test_rpn_loss_reduction_form(seed, n_img, batch):
    """objectness: mean binary cross-entropy with logits over the SAMPLED anchors (positives then negatives of the whole batch); box:
    smooth-L1 (beta 1/9) summed over the sampled positives, divided by the number of sampled anchors.  The sampler is injected (first
    entries of fixed permutations) so that the brute force can name the sampled set."""
    g = torch.Generator().manual_seed(seed)
    A = 17
    labels = [torch.tensor([[-1.0, 0.0, 0.0, 1.0][int(v)] for v in torch.randint(0, 4, (A,), generator=g).tolist()]) for _ in range(n_img)]
    reg_t = [torch.randn(A, 4, generator=g) * 0.4 for _ in range(n_img)]
    obj = torch.randn(n_img * A, 1, generator=g) * 2
    deltas = torch.randn(n_img * A, 4, generator=g)
    rpn = od.RegionProposalNetwork(randperm_fn=lambda n: torch.arange(n))       # identity "permutation": the first candidates are drawn
    rpn.fg_bg_sampler = od.BalancedPositiveNegativeSampler(batch, 0.5, lambda n: torch.arange(n))
    got_o, got_b = rpn.compute_loss(obj, deltas, labels, reg_t)
    pos_all, neg_all = [], []
    for i, l in enumerate(labels):
        P = [j for j, v in enumerate(l.tolist()) if v >= 1]
        Ng = [j for j, v in enumerate(l.tolist()) if v == 0]
        n_pos = min(len(P), int(batch * 0.5))
        n_neg = min(len(Ng), batch - n_pos)
        pos_all += [i * A + j for j in P[:n_pos]]
        neg_all += [i * A + j for j in Ng[:n_neg]]
    sampled = pos_all + neg_all
    if not sampled:
        return
    flat_l = [v for l in labels for v in l.tolist()]
    flat_t = [row for t_ in reg_t for row in t_.tolist()]
    lo = 0.0
    for j in sampled:
        x, y = float(obj[j, 0]), flat_l[j]
        lo += max(x, 0.0) - x * y + math.log1p(math.exp(-abs(x)))
    lb = sum(_smooth_l1(float(deltas[j, k]) - flat_t[j][k], 1.0 / 9) for j in pos_all for k in range(4))
    assert abs(float(got_o) - lo / len(sampled)) <= 2e-6 * max(1.0, lo / len(sampled))
    assert abs(float(got_b) - lb / len(sampled)) <= 2e-6 * max(1.0, lb / len(sampled))


@settings(max_examples=80, deadline=None)
@given(seed=st.integers(0, 10 ** 6), n_gt=st.integers(0, 4))
def test_fcos_centre_sampling_assignment_against_its_definition(seed, n_gt):
    """A location (the centre of its stride-sized anchor) is a candidate for a ground-truth box when (1) it lies within 1.5 strides of
    the box centre in both axes, (2) strictly inside the box, (3) its largest side distance is inside the level's range (4 s, 8 s), the
    first level starting at 0 and the last open-ended; among the candidates the box of the SMALLEST area wins (ties: the first), no
    candidate: -1.  Half-integer geometry, so that every comparison is exact in fp32."""
    from oracle import fcos as ofc
    g = torch.Generator().manual_seed(seed)
    strides, grids = [8, 16, 32], [(6, 6), (3, 3), (2, 2)]
    anchors, per_level = [], []
    for s, (gh, gw) in zip(strides, grids):
        for y in range(gh):
            for x in range(gw):
                cx, cy = x * s + s // 2, y * s + s // 2
                anchors.append([cx - s / 2, cy - s / 2, cx + s / 2, cy + s / 2])
        per_level.append(gh * gw)
    anchors = torch.tensor(anchors, dtype=torch.float32)
    xy = torch.randint(0, 60, (n_gt, 2), generator=g).float() / 2
    wh = torch.randint(2, 120, (n_gt, 2), generator=g).float() / 2
    gt = torch.cat([xy, xy + wh], dim=1)
    model = types.SimpleNamespace(center_sampling_radius=1.5)
    got = ofc.FCOS.match(model, anchors, {"boxes": gt}, per_level).tolist()
    want = []
    lvl_of = [li for li, n in enumerate(per_level) for _ in range(n)]
    for a, li in zip(anchors.tolist(), lvl_of):
        cx, cy, size = (a[0] + a[2]) / 2, (a[1] + a[3]) / 2, a[2] - a[0]
        lower = 0.0 if li == 0 else 4 * size
        upper = float("inf") if li == len(per_level) - 1 else 8 * size
        best, best_area = -1, None
        for j, b in enumerate(gt.tolist()):
            gx, gy = (b[0] + b[2]) / 2, (b[1] + b[3]) / 2
            d = [cx - b[0], cy - b[1], b[2] - cx, b[3] - cy]
            ok = max(abs(cx - gx), abs(cy - gy)) < 1.5 * size and min(d) > 0 and lower < max(d) < upper
            area = (b[2] - b[0]) * (b[3] - b[1])
            if ok and (best_area is None or area < best_area):
                best, best_area = j, area
        want.append(best)
    assert got == want


def test_last_level_p6p7_block():
    """P6 = conv3x3 / stride 2 / pad 1 of P5 (in == out channels: `use_P5`), P7 = the same kind of conv of relu(P6): checked against
    ATen convolutions composed here; the extents of the detector's 300 x 300 input: 10 -> 5 -> 3."""
    from oracle import retinanet as orr
    torch.manual_seed(3)
    fpn = orr.FPN3(in_channels=(8, 16, 24), out_channels=8)
    xs = [torch.randn(2, 8, 38, 38), torch.randn(2, 16, 19, 19), torch.randn(2, 24, 10, 10)]
    out = fpn(xs, lambda t: t)
    assert fpn.extra_blocks.use_P5 and [tuple(out[k].shape[-2:]) for k in ("0", "1", "2", "p6", "p7")] == [(38, 38), (19, 19), (10, 10), (5, 5), (3, 3)]
    p5 = out["2"]
    p6 = torch.nn.functional.conv2d(p5, fpn.extra_blocks.p6.weight, fpn.extra_blocks.p6.bias, stride=2, padding=1)
    p7 = torch.nn.functional.conv2d(torch.relu(p6), fpn.extra_blocks.p7.weight, fpn.extra_blocks.p7.bias, stride=2, padding=1)
    assert torch.equal(out["p6"], p6) and torch.equal(out["p7"], p7)
    # top-down pathway: P4 = layer(lateral(C4) + nearest-upsampled inner P5) -- the inner map, not the smoothed output, is passed down
    inner5 = fpn.inner_blocks[2](xs[2])
    inner4 = fpn.inner_blocks[1](xs[1]) + torch.nn.functional.interpolate(inner5, size=(19, 19), mode="nearest")
    assert torch.allclose(out["1"], fpn.layer_blocks[1](inner4), atol=1e-6)


@settings(max_examples=60, deadline=None)
@given(seed=st.integers(0, 10 ** 6), sr=st.sampled_from([1, 2, 3]), P=st.sampled_from([1, 2, 7]), scale=st.sampled_from([1.0, 0.5, 0.25]))
def test_roi_align_on_a_linear_ramp_has_a_closed_form(seed, sr, P, scale):
    """A known answer that needs no implementation at all: bilinear interpolation reproduces a LINEAR feature f(y, x) = a y + b x + c
    exactly, and the sampling points of a bin are symmetric about its centre, so RoIAlign (aligned=False, any sampling ratio) of a
    RoI that stays inside [0, H-1] x [0, W-1] is f at the bin centres: a (y0 + (ph + 1/2) bh) + b (x0 + (pw + 1/2) bw) + c with
    (x0, y0) the scaled RoI corner and bh, bw = max(extent, 1) / P.  Both oracle forms (the scalar loop and the vectorised gather the
    detector uses) must reproduce it."""
    g = torch.Generator().manual_seed(seed)
    H, W = 24, 31
    a, b, c = [float(v) for v in torch.randn(3, generator=g)]
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    feat = torch.stack([a * yy + b * xx + c, -b * yy + a * xx - c]).float()[None]            # two channels, one image
    R = 5
    x0 = torch.rand(R, generator=g) * (W - 12) / scale
    y0 = torch.rand(R, generator=g) * (H - 12) / scale
    w = (torch.rand(R, generator=g) * 9 + 0.2) / scale                                       # some narrower than one pixel after scaling
    h = (torch.rand(R, generator=g) * 9 + 0.2) / scale
    rois = torch.stack([torch.zeros(R), x0, y0, x0 + w, y0 + h], dim=1)
    want = torch.zeros(R, 2, P, P, dtype=torch.float64)
    for r in range(R):
        sx0, sy0, sx1, sy1 = [float(torch.tensor(float(v) * scale, dtype=torch.float32)) for v in rois[r, 1:]]
        bw, bh = max(sx1 - sx0, 1.0) / P, max(sy1 - sy0, 1.0) / P
        for ph in range(P):
            for pw in range(P):
                cy, cx = sy0 + (ph + 0.5) * bh, sx0 + (pw + 0.5) * bw
                want[r, 0, ph, pw] = a * cy + b * cx + c
                want[r, 1, ph, pw] = -b * cy + a * cx - c
    got_scalar = ok.roi_align_nchw(feat, rois, P, P, scale, sr)
    got_vec = od.roi_align_autograd(feat, rois, P, scale, sr)
    tol = 2e-4 * (abs(a) + abs(b)) * max(H, W) + 1e-5
    assert float((got_scalar.double() - want).abs().max()) <= tol
    assert float((got_vec.double() - want).abs().max()) <= tol
