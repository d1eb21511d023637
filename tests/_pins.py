"""Test helper: record the product detector's DISCRETE forward decisions (ReLU on/off, max-pool winners) and hand them to the
CPU oracle (oracle.detection.Pins), so that both sides differentiate the same piecewise-linear function.

Why: the product stores activations in fp16.  An fp32-accumulating restatement of the same network sees activations that differ by
~1e-3 relative; wherever a pre-activation lies inside that noise band its ReLU flips, and one flip re-routes the gradient of
everything behind it -- after ~70 layers two CORRECT implementations agree only statistically (cosine 0.97 measured).  A missing
FPN branch would hide in that slack.  With the decisions shared what remains is rounding / summation order, so the gradients are
held to rel-L2 <= 3 % and cosine >= 0.999 and a mis-routed or mis-scaled branch cannot pass."""
import torch

from oracle import detection as od


class _FirstWins(dict):
    def __setitem__(self, k, v):
        if k not in self:
            super().__setitem__(k, v)


class _OpsProxy:
    """Stands in for the `ops` module inside ONE product module for the duration of a recording: chosen functions report their
    results, everything else passes through."""

    def __init__(self, real, **wrapped):
        self._real, self._wrapped = real, wrapped

    def __getattr__(self, name):
        return self._wrapped.get(name) or getattr(self._real, name)


class record:
    """`with record(detector) as r: <product forward>`; afterwards `r.pins()` builds the oracle-side object.  `first=True` keeps the
    FIRST tensor recorded under a tag (a training step runs the detector three times; only the first, hallucinated, pass carries a
    gradient).

    The product carries NO test hook.  The recording is done from here: for the duration of the `with` block the module-level helpers
    every detector convolution goes through (`models.detection._fwd` / `_fwd_many`), the batched proposal filter, and the `ops` name
    of the detection / FCOS modules are wrapped; a call is recognised by the IDENTITY of the packed-weight entry it is given (the
    detector's `pack()` dictionaries are walked to label them: stem, bottleneck (stage, block, conv), RPN head per level, fc6 / fc7,
    the RetinaNet / FCOS tower layers), and the post-ReLU output (or the pool's winner bytes, the post-NMS proposals) is stored under
    the tag the oracle's `Pins.relu(tag, .)` asks for."""

    def __init__(self, model=None, first=False):
        self.model = model
        self.tap = _FirstWins() if first else {}
        self._tags = {}
        self._fcos_tower_ids, self._fcos_nl = set(), 5

    # ---- labels of the packed entries ------------------------------------------------------------------------------------------
    def _label(self):
        m, tags = self.model, {}
        if m is None:
            return tags
        bb = getattr(m, "backbone", None)
        if bb is not None and getattr(bb, "_pack", None) is not None:
            P = bb._pack
            tags[id(P["stem"])] = ("stem",)
            for si, stage in enumerate(P["blocks"]):
                for bi, e in enumerate(stage):
                    for k, name in ((1, "c1"), (2, "c2"), (3, "c3")):
                        tags[id(e[name])] = ("b", si, bi, k)
            if "p7" in P:
                tags[id(P["p7"])] = ("p7in", "input")              # the ReLU in front of P7: its input is what is recorded
        rpn = getattr(m, "rpn", None)
        if rpn is not None and getattr(rpn.head, "_pack", None) is not None:
            tags[id(rpn.head._pack["conv"])] = ("rpn", "level")
        rh = getattr(m, "roi_heads", None)
        if rh is not None and getattr(rh.box_head, "_pack", None) is not None:
            tags[id(rh.box_head._pack["fc6"])] = ("fc6",)
            tags[id(rh.box_head._pack["fc7"])] = ("fc7",)
        head = getattr(m, "head", None)
        if head is not None and getattr(head, "_pack", None) is not None:
            P = head._pack
            for key, name in (("cls_tower", "cls"), ("reg_tower", "reg")):
                for k, e in enumerate(P.get(key, [])):
                    if isinstance(e, dict):                          # RetinaNet: conv + ReLU per tower layer (FCOS: GroupNorm, see below)
                        tags[id(e)] = (name, "level", k)
            if isinstance(P.get("cls_tower", [None])[0], tuple):     # FCOS: bbox_reg output (its ReLU comes after)
                tags[id(P["reg_out"])] = ("reg_out", "level")
                self._fcos_tower_ids = {id(t[0]) for t in P["cls_tower"]}
        return tags

    def _tag(self, e):
        t = self._tags.get(id(e))
        if t is None:
            self._tags = self._label()
            t = self._tags.get(id(e))
        return t

    def _put(self, tag, t, li=None):
        if tag is None:
            return
        tag = tuple(li if x == "level" else x for x in tag)
        self.tap[tag] = t

    def __enter__(self):
        from hallucidet_amd.models import detection as D
        from hallucidet_amd.models import fcos as F
        from hallucidet_amd.utils import eval_forward_fasterrcnn as G
        self._saved = (D._fwd, D._fwd_many, D.filter_proposals_padded, D.ops, F.ops, D.RegionProposalNetwork.filter_proposals)
        o_fwd, o_many, o_filter, o_ops, f_ops, o_list_filter = self._saved
        rec = self

        def fwd(e, x, **kw):
            out = o_fwd(e, x, **kw)
            tag = rec._tag(e)
            if tag is not None:
                if tag[-1] == "input":
                    rec._put(tag[:-1], x)
                else:
                    rec._put(tag, out)
            return out

        def many(es, xs, **kw):
            outs = o_many(es, xs, **kw)
            tagged = [rec._tag(e) for e in es]
            if id(es[0]) in rec._fcos_tower_ids:
                rec._fcos_nl = len(es) // 2
            # the same layer on every feature level: the level index is the position among the calls with that entry
            seen = {}
            for e, tag, o in zip(es, tagged, outs):
                if tag is not None:
                    li = seen.get(id(e), 0)
                    seen[id(e)] = li + 1
                    rec._put(tag, o, li)
            return outs

        def filt(*a, **k):
            out = o_filter(*a, **k)
            rec.tap[("proposals",)] = (out[0], out[2])
            return out

        def list_filt(self_, *a, **k):
            boxes, scores = o_list_filter(self_, *a, **k)
            rec.tap[("proposals",)] = boxes
            return boxes, scores

        def pool(x):
            y, idx = o_ops.maxpool3x3s2_idx(x)
            rec.tap[("pool",)] = idx
            return y, idx

        gn_calls = [0]

        def gn(x, ga, be, eps=1e-5, **kw):
            # FCOS tower: layer k, then cls levels 0..L-1, then reg levels 0..L-1 (models/fcos.py: _HeadFn.forward)
            z, stat = f_ops.groupnorm8_relu(x, ga, be, eps, **kw)
            nl = rec._fcos_nl
            c = gn_calls[0] % (8 * nl)
            gn_calls[0] += 1
            k, idx = divmod(c, 2 * nl)
            rec.tap[("cls" if idx < nl else "reg", idx % nl, k)] = z
            return z, stat

        D._fwd, D._fwd_many, D.filter_proposals_padded = fwd, many, filt
        D.RegionProposalNetwork.filter_proposals = list_filt
        D.ops = _OpsProxy(o_ops, maxpool3x3s2_idx=pool)
        F.ops = _OpsProxy(f_ops, groupnorm8_relu=gn)
        return self

    def __exit__(self, *exc):
        from hallucidet_amd.models import detection as D
        from hallucidet_amd.models import fcos as F
        D._fwd, D._fwd_many, D.filter_proposals_padded, D.ops, F.ops, D.RegionProposalNetwork.filter_proposals = self._saved
        return False

    def pins(self, proposals=None, n_images=None):
        """`n_images`: keep the first n images of every recorded map (a fused three-pass evaluation runs hall + RGB + IR as one
        batch; only the leading, hallucinated, images carry a gradient)."""
        cut = (lambda t: t) if n_images is None else (lambda t: t[:n_images])
        masks, pool, values = {}, None, {}
        for tag, t in self.tap.items():
            if t is None:
                continue
            if tag == ("proposals",):                       # post-NMS proposal sets: list of [n_i, 4] or (padded [N, K, 4], counts)
                if proposals is None:
                    if isinstance(t, tuple):
                        proposals = [b[: int(c)].detach().float().cpu() for b, c in zip(t[0], t[1])]
                    else:
                        proposals = [b.detach().float().cpu() for b in t]
            elif tag == ("pool",):
                pool = cut(t).permute(0, 3, 1, 2).long().cpu().contiguous()
            elif tag[0] == "reg_out":                       # NCHW view of an fp32 head output
                values[tag] = cut(t).detach().float().cpu().contiguous()
            elif tag[0] in ("fc6", "fc7"):                  # [R, 1, 1, C]: one row per RoI, not per image
                values[tag] = t.detach().permute(0, 3, 1, 2).float().cpu().contiguous()
            else:                                            # NHWC fp16 activation
                values[tag] = cut(t).detach().permute(0, 3, 1, 2).float().cpu().contiguous()
        masks = {tag: (v > 0).float() for tag, v in values.items()}
        return od.Pins(masks, pool, proposals, values)


# Bounds on BORROWED decisions.  A decision taken from the product may differ from the oracle's own only where the
# oracle's own value sits inside the NOISE BAND of the decision boundary; a systematically wrong mask in the product (which the oracle
# would otherwise copy and pass) flips elements far outside it.  The noise of a layer is MEASURED, not assumed, on the elements both
# evaluations pass: sigma = RMS, dmax = largest |product activation - oracle's own| (fp16 storage of everything upstream; in the
# end-to-end tier also the drift of the hallucinated image the detector starts from; per-element errors scale with the raw conv
# output in front of a BatchNorm, so the tail is heavy and the bound uses the observed maximum, not a multiple of sigma).
#   * magnitude: every differing ReLU element has |own pre-activation| <= NOISE_C * dmax of its layer -- no further from the boundary
#     than the two evaluations are SEEN to differ elsewhere in that layer; differing max-pool winners: the oracle's two candidates are
#     within 2 * NOISE_C * dmax(stem) of each other (each side moves by at most dmax);
#   * share: at most SHARE_K * sigma / RMS of a layer's elements differ (the mass of a unit-scale density inside a band of width
#     sigma; measured 0.3 - 0.75 in fp16 storage; in fp32 storage a layer has a handful of flips and the ratio is a small-count
#     statistic: SHARE_FLOOR of the elements and SMALL_COUNT flips per layer are always allowed -- one flip in a 4 608-element pyramid
#     level is 2.2e-4 --), never more than MAX_FLIP_SHARE;
#   * proposals: at least MIN_PROPOSAL_MATCH of the borrowed post-NMS boxes coincide (IoU >= 0.5; reported at 0.9 / 0.7 / 0.5) with a box
#     of the oracle's own candidate set (every anchor decoded with the oracle's deltas, clipped) -- WHICH candidates survive top-k / NMS
#     is decided by near-ties of a randomly initialised RPN and is what the pin is for; that the boxes ARE the oracle's boxes is
#     checked here.  (fp32 storage: 100 % at IoU 0.9.  fp16 storage end to end: 70 - 88 % at 0.9, 99.5 - 100 % at 0.5 -- the regression
#     deltas carry the fp16 noise of the trunk and a pixel's shift of a five-pixel box is already IoU 0.7.)
# One set of constants for every pinned test, set from the worst cases over the whole -m gpu suite (HD_PINS_AUDIT=1 prints every
# layer; tools/pins_measure.py collects the worst cases without asserting -- nothing in the environment can switch the assertion
# off inside pytest) with a margin of about two (largest seen: 2.24 dmax on a 6 000-element layer, 0.75 sigma / RMS).
NOISE_C = 4.0
SHARE_K = 2.0
SHARE_FLOOR = 2e-4
SMALL_COUNT = 2
MAX_FLIP_SHARE = 0.10
MIN_PROPOSAL_MATCH = 0.98


def audit_borrowed_decisions(holder, label=""):
    """`holder`: an oracle.detection.Pins or an oracle.unet.Ctx (built WITH the recorded values) after the oracle's forward pass.
    -> (one-line summary, list of the decisions outside the bounds above).  No assertion here: the tests call
    assert_borrowed_decisions_are_noise, tools/pins_measure.py (setting the constants) calls this."""
    import os
    audit = holder.audit
    assert audit, "no borrowed decision was audited (%s)" % label
    worst = dict(share=0.0, kmax=0.0, ksig=0.0, sratio=0.0, tag=None)
    prop, problems = [], []
    stem_dmax = stem_sigma = None
    for tag, rec in audit.items():
        if isinstance(tag, tuple) and tag and tag[0] == "proposals":
            pinned, own, hit = rec[0], rec[1], rec[2:]
            prop.append((pinned, own) + tuple(hit))
            if pinned and hit[-1] < MIN_PROPOSAL_MATCH * pinned:
                problems.append(("proposals", tag, rec))
            continue
        n, nf, big, rms, sigma, dmax = rec
        if tag == ("stem",):
            stem_dmax, stem_sigma = dmax, sigma
        limit = NOISE_C * dmax if dmax is not None else None
        if tag == ("pool",):
            sigma, dmax = stem_sigma, stem_dmax
            limit = 2 * NOISE_C * dmax if dmax is not None else None
        assert sigma is not None and dmax is not None, "decision %s was borrowed without the recorded values (%s)" % (tag, label)
        share = nf / max(n, 1)
        kmax = big / dmax if dmax else (0.0 if big == 0 else float("inf"))
        ksig = big / sigma if sigma else 0.0
        sratio = share / (sigma / max(rms, 1e-30)) if sigma else (0.0 if nf == 0 else float("inf"))
        worst["share"] = max(worst["share"], share)
        worst["ksig"] = max(worst["ksig"], ksig)
        worst["sratio"] = max(worst["sratio"], sratio)
        if kmax > worst["kmax"]:
            worst["kmax"], worst["tag"] = kmax, tag
        if os.environ.get("HD_PINS_AUDIT"):
            print("   audit %-28s n %9d differ %7d (%.4f %%) worst |x| %.3e = %.2f dmax = %.1f sigma; sigma/RMS %.2e share/(sigma/RMS) %.2f" % (
                str(tag), n, nf, 100 * share, big, kmax, ksig, sigma / max(rms, 1e-30), sratio))
        # (one or two flips in a layer are a COUNT, not a share: a 4 608-element pyramid level with one flipped element reads 2.2e-4;
        #  the magnitude bound applies to them all the same)
        if big > limit or (nf > SMALL_COUNT and share > max(SHARE_K * sigma / max(rms, 1e-30), SHARE_FLOOR)) or share > MAX_FLIP_SHARE:
            problems.append((tag, "share %.5f (%.2f sigma/RMS)" % (share, sratio), "worst %.2f dmax" % kmax))
    summary = "borrowed decisions %s: %d audited, worst share %.4f %% = %.2f sigma/RMS, worst |x| %.2f dmax (%s; %.1f sigma)%s" % (
        label, len(audit), 100 * worst["share"], worst["sratio"], worst["kmax"], worst["tag"], worst["ksig"],
        "" if not prop else ", proposals among the oracle's candidates at IoU 0.9 / 0.7 / 0.5: %d / %d / %d of %d" % (
            sum(p[2] for p in prop), sum(p[3] for p in prop), sum(p[4] for p in prop), sum(p[0] for p in prop)))
    print(summary)
    return summary, problems


def assert_borrowed_decisions_are_noise(holder, label=""):
    """Asserts the bounds above for every audited decision (fail-closed: no environment switch) and returns the one-line summary."""
    summary, problems = audit_borrowed_decisions(holder, label)
    assert not problems, "borrowed decisions outside the noise band: %s" % problems[:8]
    return summary


def unet_decisions(runner, device="cpu"):
    """-> (masks, values): the ReLU decisions of the hallucination network's last forward pass and the activations they were taken
    from (NCHW fp32), for oracle.unet.Ctx(q, masks, values)."""
    values = {k: v.permute(0, 3, 1, 2).float().to(device) for k, v in runner.saved_activations().items() if not k.endswith("downsample")}
    return {k: (v > 0).float() for k, v in values.items()}, values


def grad_agreement(got, want):
    a, b = got.flatten().double().cpu(), want.flatten().double().cpu()
    cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
    rel = float((a - b).norm() / (b.norm() + 1e-30))
    return cos, rel


def product_weight_numerics_(oracle):
    """Give an UNFOLDED oracle detector (conv weights and FrozenBN kept as separate tensors, as parameter-gradient tests need)
    the product's weight numerics: the product multiplies the FrozenBN scale into the conv weight and rounds the result to fp16;
    here every conv weight becomes half(w * s) / s, so that w * s reproduces that fp16 value, and convs with a bias (FPN, heads)
    and FC layers are rounded to fp16 directly.  dL/dw is then evaluated at the same effective weights on both sides."""
    from oracle import detection as od
    with torch.no_grad():
        done = set()
        for mod in oracle.modules():
            if isinstance(mod, od.Bottleneck):
                pairs = [(mod.conv1, mod.bn1), (mod.conv2, mod.bn2), (mod.conv3, mod.bn3)]
                if mod.downsample is not None:
                    pairs.append((mod.downsample[0], mod.downsample[1]))
            elif isinstance(mod, od.ResNet50Body):
                pairs = [(mod.conv1, mod.bn1)]
            else:
                continue
            for conv, bn in pairs:
                s = bn.scale_shift()[0][:, None, None, None]
                conv.weight.copy_((conv.weight * s).half().float() / s)
                done.add(id(conv))
        for mod in oracle.modules():
            if isinstance(mod, (torch.nn.Conv2d, torch.nn.Linear)) and id(mod) not in done:
                mod.weight.copy_(mod.weight.half().float())
