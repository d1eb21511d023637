"""train_detector.py step (BASELINE configs[4]: fasterrcnn, RGB, batch 16/GPU) on synthetic 512x640 images."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hallucidet_amd import synthetic
from hallucidet_amd.models.detector import Detector
from hallucidet_amd.train_detector import DetectorLit

N = int(os.environ.get("N", 16)); steps = int(os.environ.get("STEPS", 10))
torch.manual_seed(1)
det = Detector(name="fasterrcnn", pretrained=False, n_classes=2, size=300).detector.to("cuda")
rgb, trgb, _, _ = synthetic.make_batch(N, device="cuda")
il, _ = det.transform(rgb[:2], None)
det.backbone.calibrate_(il.tensors)
lit = DetectorLit(batch_size=N, detector=det, pretrained=False).prepare()
for _ in range(3):
    l = lit.fit_step((rgb, trgb))
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    l = lit.fit_step((rgb, trgb))
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print("train_detector step: %.2f ms -> %.1f img/s (batch %d) loss %.4f found_inf %g" % (dt * 1e3, N / dt, N, float(l), float(lit.optimizer.found_inf)))
