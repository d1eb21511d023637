import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd.models.detector import Detector
from hallucidet_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(11)
det = Detector(name="fasterrcnn", pretrained=False, n_classes=2, size=300).detector.to(dev).eval()
images = torch.rand(6, 3, 96, 128, device=dev)
il, _ = det.transform(images, None)
det.backbone.calibrate_(il.tensors[:2].contiguous())
with torch.no_grad():
    f6 = det.backbone(il.tensors)
    f2 = det.backbone(il.tensors[:2].contiguous())
    for k in f6:
        print(k, torch.equal(f6[k][:2], f2[k]), float((f6[k][:2].float() - f2[k].float()).abs().max()))
    o6, r6 = det.rpn.head(list(f6.values()))
    o2, r2 = det.rpn.head(list(f2.values()))
    for a, b in zip(o6 + r6, o2 + r2):
        print("head", torch.equal(a[:2], b), float((a[:2] - b).abs().max()))
    # layer by layer inside the body
    P = det.backbone.pack()
    from hallucidet_amd.models.detection import _fwd
    from hallucidet_amd.ops import ACT_RELU
    x6, x2 = il.tensors, il.tensors[:2].contiguous()
    s6, s2 = _fwd(P["stem"], x6, act=ACT_RELU), _fwd(P["stem"], x2, act=ACT_RELU)
    print("stem", torch.equal(s6[:2], s2))
    p6, p2 = ops.maxpool3x3s2(s6), ops.maxpool3x3s2(s2)
    print("pool", torch.equal(p6[:2], p2))
    e = P["blocks"][0][0]
    a6, a2 = _fwd(e["c1"], p6, act=ACT_RELU), _fwd(e["c1"], p2, act=ACT_RELU)
    print("l1.0.c1", torch.equal(a6[:2], a2))
    b6, b2 = _fwd(e["c2"], a6, act=ACT_RELU), _fwd(e["c2"], a2, act=ACT_RELU)
    print("l1.0.c2", torch.equal(b6[:2], b2))
    d6, d2 = _fwd(e["ds"], p6), _fwd(e["ds"], p2)
    print("l1.0.ds", torch.equal(d6[:2], d2))
    c6, c2 = _fwd(e["c3"], b6, act=ACT_RELU, res=d6), _fwd(e["c3"], b2, act=ACT_RELU, res=d2)
    print("l1.0.c3", torch.equal(c6[:2], c2))
