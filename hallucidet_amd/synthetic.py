"""Synthetic workload of the BASELINE metric: seed 123, IR [N,1,512,640] / RGB [N,3,512,640] uniform[0,1),
1-8 pedestrian-like boxes per image (label 1), identical IR/RGB targets (LLVIP is aligned)."""
import torch


def make_batch(n=8, h=512, w=640, seed=123, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    imgs_ir = torch.rand(n, 1, h, w, generator=g)
    imgs_rgb = torch.rand(n, 3, h, w, generator=g)
    targets = []
    for _ in range(n):
        k = int(torch.randint(1, 9, (1,), generator=g))
        x1 = torch.rand(k, generator=g) * (w - 80)
        y1 = torch.rand(k, generator=g) * (h - 112)
        bw = torch.rand(k, generator=g) * 64 + 16
        bh = torch.rand(k, generator=g) * 80 + 32
        boxes = torch.stack([x1, y1, (x1 + bw).clamp(max=w), (y1 + bh).clamp(max=h)], dim=1)
        targets.append({"boxes": boxes, "labels": torch.ones(k, dtype=torch.int64)})
    dev = torch.device(device)
    imgs_ir, imgs_rgb = imgs_ir.to(dev), imgs_rgb.to(dev)
    targets = [{k: v.to(dev) for k, v in t.items()} for t in targets]
    return imgs_rgb, targets, imgs_ir, [dict(t) for t in targets]


def make_module(seed=123, device="cuda", precision=16, calibrate_on=None, detector_name="fasterrcnn"):
    """Random-init U-Net (reference init rules) + random-init detector whose FrozenBN statistics are calibrated on a
    synthetic batch (there are no checkpoints offline)."""
    from .train_hallucidet import EncoderDecoderLit
    torch.manual_seed(seed)
    lit = EncoderDecoderLit(batch_size=8, model_name="resnet34", detector_name=detector_name, precision=precision, device=device)
    lit.prepare()
    if calibrate_on is None:
        calibrate_on = make_batch(2, seed=seed + 1, device=device)[0]
    with torch.no_grad():
        il, _ = lit.detector.transform(calibrate_on, None)
        lit.detector.backbone.calibrate_(il.tensors)
    return lit
