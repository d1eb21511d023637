"""Bisect the fp32 mode against the plain oracle: (1) detector image gradient, (2) U-Net parameter gradients for a fixed output gradient."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import ops, synthetic
from oracle import detection as od, unet as ou

dev = torch.device("cuda:0")
rel = lambda a, b: float((a.double().cpu().flatten() - b.double().cpu().flatten()).norm() / (b.double().flatten().norm() + 1e-30))

lit = synthetic.make_module(seed=5, device="cuda:0", precision=32, detector_name="fasterrcnn")
det = lit.detector
oracle = od.FasterRCNN(num_classes=2, size=300)
oracle.load_state_dict({k: v.cpu() for k, v in det.state_dict().items()})
oracle.eval()
g = torch.Generator().manual_seed(1)
images = torch.rand(2, 3, 96, 128, generator=g)
props = []
for i in range(2):
    xy = torch.rand(30, 2, generator=g) * 200
    wh = torch.rand(30, 2, generator=g) * 90 + 8
    props.append(torch.cat([xy, xy + wh], 1))
with ops.storage(torch.float32):
    x = images.to(dev).requires_grad_(True)
    il, _ = det.transform(x, None)
    f = det.backbone(il.tensors)
    obj, reg = det.rpn.head(list(f.values()))
    bf = det.roi_heads.box_roi_pool(f, [p.to(dev) for p in props], il.image_sizes)
    logits, regs = det.roi_heads.box_predictor(det.roi_heads.box_head(bf))
    ws = [torch.randn(o.shape, generator=g) for o in list(obj) + list(reg)] + [torch.randn(logits.shape, generator=g), torch.randn(regs.shape, generator=g)]
    outs = list(obj) + list(reg) + [logits, regs]
    sum((o * w.to(dev)).sum() for o, w in zip(outs, ws)).backward()
xo = images.clone().requires_grad_(True)
ol, _ = oracle.transform(xo, None)
of = oracle.backbone(ol.tensors)
oobj, oreg = oracle.rpn.head(list(of.values()))
obf = oracle.roi_heads.box_roi_pool(of, props, ol.image_sizes)
ologits, oregs = oracle.roi_heads.box_predictor(oracle.roi_heads.box_head(obf))
oouts = list(oobj) + list(oreg) + [ologits, oregs]
for i, (a, b) in enumerate(zip(outs, oouts)):
    print("head output %d rel %.2e" % (i, rel(a.detach().reshape(b.shape) if a.numel() == b.numel() else a.detach(), b.detach())))
for k, (a, b) in zip(f.keys(), zip(f.values(), of.values())):
    print("feature %s rel %.2e" % (k, rel(a.detach().permute(0, 3, 1, 2), b.detach())))
sum((o * w).sum() for o, w in zip(oouts, ws)).backward()
print("detector image gradient rel %.3e" % rel(x.grad, xo.grad))

# per-branch
for name, sel in (("rpn only", range(0, 10)), ("box only", range(10, 12))):
    with ops.storage(torch.float32):
        x = images.to(dev).requires_grad_(True)
        il, _ = det.transform(x, None)
        f = det.backbone(il.tensors)
        obj, reg = det.rpn.head(list(f.values()))
        bf = det.roi_heads.box_roi_pool(f, [p.to(dev) for p in props], il.image_sizes)
        logits, regs = det.roi_heads.box_predictor(det.roi_heads.box_head(bf))
        outs = list(obj) + list(reg) + [logits, regs]
        sum((outs[i] * ws[i].to(dev)).sum() for i in sel).backward()
    xo = images.clone().requires_grad_(True)
    ol, _ = oracle.transform(xo, None)
    of = oracle.backbone(ol.tensors)
    oobj, oreg = oracle.rpn.head(list(of.values()))
    obf = oracle.roi_heads.box_roi_pool(of, props, ol.image_sizes)
    ologits, oregs = oracle.roi_heads.box_predictor(oracle.roi_heads.box_head(obf))
    oouts = list(oobj) + list(oreg) + [ologits, oregs]
    sum((oouts[i] * ws[i]).sum() for i in sel).backward()
    print("%s: image gradient rel %.3e" % (name, rel(x.grad, xo.grad)))

# U-Net
net = lit.encoder_decoder
ref = ou.Unet(classes=3)
ref.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
net.train(); ref.train()
xi = torch.rand(2, 3, 128, 160, generator=g)
gout = torch.randn(2, 3, 128, 160, generator=g) * 1e-2
with ops.storage(torch.float32):
    out = net(xi.to(dev))
    (out * gout.to(dev)).sum().backward()
w = ref(xi)
(w * gout).sum().backward()
print("unet output rel %.2e" % rel(out.detach(), w.detach()))
bad = [(n, rel(p.grad, q.grad)) for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters())]
bad.sort(key=lambda t: -t[1])
print("unet worst grads:", bad[:6])
print("unet median grad rel %.2e" % sorted(b[1] for b in bad)[len(bad) // 2])
