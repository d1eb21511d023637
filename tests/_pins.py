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


class record:
    """`with record() as r: <product forward>`; afterwards `r.pins()` builds the oracle-side object.  `first=True` keeps the
    FIRST tensor recorded under a tag (a training step runs the detector three times; only the first, hallucinated, pass
    carries a gradient)."""

    def __init__(self, first=False):
        self.tap = _FirstWins() if first else {}

    def __enter__(self):
        from hallucidet_amd.models import detection as D
        assert D._TAP is None
        D._TAP = self.tap
        return self

    def __exit__(self, *exc):
        from hallucidet_amd.models import detection as D
        D._TAP = None
        return False

    def pins(self, proposals=None, n_images=None):
        """`n_images`: keep the first n images of every recorded map (a fused three-pass evaluation runs hall + RGB + IR as one
        batch; only the leading, hallucinated, images carry a gradient)."""
        cut = (lambda t: t) if n_images is None else (lambda t: t[:n_images])
        masks, pool = {}, None
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
                masks[tag] = (cut(t).detach() > 0).float().cpu().contiguous()
            elif tag[0] in ("fc6", "fc7"):                  # [R, 1, 1, C]: one row per RoI, not per image
                masks[tag] = (t.detach().permute(0, 3, 1, 2) > 0).float().cpu().contiguous()
            else:                                            # NHWC fp16 activation
                masks[tag] = (cut(t).detach().permute(0, 3, 1, 2) > 0).float().cpu().contiguous()
        return od.Pins(masks, pool, proposals)


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
