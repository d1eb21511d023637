"""TEST INFRASTRUCTURE ONLY -- CPU restatement of one `train_hallucidet` step
(train_hallucidet.py:161-240 forward_step + Lightning's backward / clip_grad_value_(0.5) / Adam(lr) [EXT]):
U-Net forward, three detector passes (hallucinated with gradient; RGB and IR for their detections), the 0.1-weighted
sum of the four Faster R-CNN losses, backward into the U-Net only, value clipping, one Adam update.
Used by tests (end-to-end parity) and by bench.py's `cpu_baseline` leg; never by the product."""
import time

import torch

from . import detection as od
from . import retinanet as orn
from . import unet as ou

LOSS_WEIGHTS = {"loss_box_reg": 0.1, "loss_classifier": 0.1, "loss_objectness": 0.1, "loss_rpn_box_reg": 0.1}
LOSS_WEIGHTS_RETINANET = {"classification": 0.1, "bbox_regression": 0.1}      # train_hallucidet.py:196-197 + config.py weights


class OracleTrainer:
    def __init__(self, unet=None, detector=None, lr=1e-4, clip=0.5, seed=123):
        torch.manual_seed(seed)
        self.unet = unet if unet is not None else ou.Unet(classes=3)
        self.det = detector if detector is not None else od.FasterRCNN(num_classes=2, size=300)
        self.det.eval()
        for p in self.det.parameters():
            p.requires_grad = False
        self.opt = torch.optim.Adam(self.unet.parameters(), lr=lr)
        self.clip = clip
        self.unet_q = None          # optional activation rounding schedule (tests: oracle.unet.fp16_round)

    def forward_step(self, imgs_rgb, targets_rgb, imgs_ir, targets_ir, det_pins=None):
        """`det_pins` (tests, oracle.detection.Pins): discrete decisions of the HALLUCINATED pass -- the only one that carries a
        gradient -- taken from a recording of the product; the RGB / IR passes always run plain."""
        ir3 = imgs_ir.repeat(1, 3, 1, 1) if imgs_ir.shape[1] == 1 else imgs_ir
        hall = self.unet(ir3) if self.unet_q is None else self.unet(ir3, q=self.unet_q)
        self.last_hall = hall.detach()
        retina = isinstance(self.det, orn.RetinaNet)
        fwd = orn.eval_forward_retinanet if retina else od.eval_forward_fasterrcnn
        if det_pins is not None:
            self.det.set_pins(det_pins)
        try:
            losses, det_h = fwd(self.det, hall, targets_ir)
        finally:
            if det_pins is not None:
                self.det.set_pins(None)
        with torch.no_grad():
            _, det_rgb = fwd(self.det, imgs_rgb, targets_rgb)
            _, det_ir = fwd(self.det, ir3, targets_ir)
        total = sum(losses[k] * w for k, w in (LOSS_WEIGHTS_RETINANET if retina else LOSS_WEIGHTS).items())
        return total, losses, (det_h, det_rgb, det_ir)

    def train_step(self, batch):
        imgs_rgb, targets_rgb, imgs_ir, targets_ir = batch
        self.unet.train()
        total, losses, _ = self.forward_step(imgs_rgb, targets_rgb, imgs_ir, targets_ir)
        self.opt.zero_grad(set_to_none=True)
        total.backward()
        torch.nn.utils.clip_grad_value_(self.unet.parameters(), self.clip)
        self.opt.step()
        return total.detach(), {k: v.detach() for k, v in losses.items()}


def time_cpu_step(batch, threads=None, budget_s=30.0, max_steps=3):
    """Bounded sample: at least one full step, more while the budget lasts.  Returns (images/s, steps, threads)."""
    if threads:
        torch.set_num_threads(threads)
    tr = OracleTrainer()
    n = batch[0].shape[0]
    t0 = time.time()
    steps = 0
    while steps < max_steps and (steps == 0 or time.time() - t0 < budget_s):
        tr.train_step(batch)
        steps += 1
    dt = time.time() - t0
    return n * steps / dt, steps, torch.get_num_threads()
