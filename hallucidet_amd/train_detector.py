"""Detector-only training (reference train_detector.py:86-345, BASELINE configs[4]): `DetectorLit` with the reference's
constructor keywords and hook names (`training_step`, `validation_step`, `test_step`, `configure_optimizers`), driven by
`fit_step` instead of `pl.Trainer` (Lightning is not installable offline; its loop around training_step is: zero_grad ->
backward -> optimizer.step, no gradient clipping in this script, train_detector.py:377-390).

What runs where: the detector forward is the same HIP path as the frozen detector; `FasterRCNN.set_trainable(True)` makes
the hand-written backward chains also emit PARAMETER gradients (hd_wgrad / hd_wgrad_reduce for every trainable conv and
FC, hd_channel_sum_f16 for biases) for what torchvision leaves trainable (body.layer2-4, FPN, RPN, RoI heads; FrozenBN,
conv1, layer1 fixed; RetinaNet / FCOS: body.layer2-4, FPN incl. P6/P7, the head towers -- FCOS with hd_groupnorm8_param_grad for its
GroupNorm affines), accumulated into one flat fp32 arena; Adam is the fused hd_adam_step over that arena; the data-
parallel exchange is the all-reduce of the arena (same GradientAverager as the hallucination network).
fp16 storage needs a loss scale for the small detector gradients: a fixed power of two (default 1024) that is removed
inside hd_wgrad_reduce; steps with non-finite gradients are skipped (hd_check_finite), as GradScaler would.
"""
import torch

from . import ops
from .config import Config
from .distributed import GradientAverager, broadcast_parameters
from .models.detector import Detector
from .optim import FusedAdam, LossScaler, ParamArena
from .utils.utils import Utils


class DetectorLit:
    def __init__(self, batch_size=4, wandb_logger=None, lr=0.0001, detector_name='fasterrcnn', pretrained=True, optimizer_name='adam',
                 modality=None, directly_coco=False, detector=None, device='cuda', loss_scale=1024.0, precision=16):
        if not any(k in detector_name for k in ('fasterrcnn', 'retinanet', 'fcos')):
            raise ValueError("unknown detector %r (fasterrcnn / retinanet / fcos)" % (detector_name,))
        self.wandb_logger, self.lr, self.batch_size = wandb_logger, lr, batch_size
        self.optimizer_name, self.detector_name, self.modality = optimizer_name, detector_name, modality
        self.dev = device
        self.detector = detector if detector is not None else Detector(name=detector_name, pretrained=pretrained,
                                                                       n_classes=getattr(getattr(Config, 'Dataset', None), 'n_classes', 2), size=Config.Detector.input_size,
                                                                       modality=modality, directly_coco=directly_coco).detector
        self.detector.fused_passes = False
        # precision 32 (the reference's default, src/config/config.py:149 -> train_detector.py:387): fp32 storage, no loss scaling
        if int(precision) not in (16, 32):
            raise ValueError("precision must be 16 or 32 (got %r)" % (precision,))
        self.precision = int(precision)
        self.act_dtype = torch.float32 if self.precision == 32 else torch.float16
        self.loss_scale = float(loss_scale) if self.precision == 16 else 1.0
        self.optimizer = self.scaler = self.arena = self.averager = None
        self._last_detections = None

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, strict=True, **kwargs):
        """`detectorLit.load_from_checkpoint(checkpoint_path=args.detector_path, ...).detector` (train_hallucidet.py:107-115)."""
        from .checkpoint import load_detector
        lit = cls(**kwargs)
        load_detector(lit.detector, checkpoint_path, strict=strict)
        return lit

    def save_checkpoint(self, path, epoch=0, global_step=0):
        from .checkpoint import save_lightning_checkpoint
        return save_lightning_checkpoint(path, {"detector": self.detector}, epoch, global_step)

    # ------------------------------------------------------------------ setup
    def configure_optimizers(self):
        if self.optimizer_name != 'adam':
            raise NotImplementedError("the reference runs train_detector.py with Config.Optimizer.name == 'adam'")
        self.detector.to(self.dev)
        self.detector.set_trainable(True, grad_scale=self.loss_scale)
        self.arena = ParamArena(self.detector.trainable_parameters())
        self.optimizer = FusedAdam(self.arena, lr=self.lr, clip_value=0.0)
        self.scaler = LossScaler(self.arena, init_scale=self.loss_scale, growth_interval=1 << 30)   # fixed scale
        # train_detector.py:336-343: ReduceLROnPlateau(optimizer, mode='min') monitored on val_loss
        self.lr_scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode='min')
        return {"optimizer": self.optimizer, "lr_scheduler": {"scheduler": self.lr_scheduler, "monitor": "val_loss"}}

    def lr_scheduler_step(self, val_loss):
        self.lr_scheduler.step(float(val_loss))
        return self.optimizer.param_groups[0]["lr"]

    def prepare(self):
        self.configure_optimizers()
        broadcast_parameters(self.arena.flat_params)
        self.averager = GradientAverager()
        return self

    # ------------------------------------------------------------------ hooks (train_detector.py:147-203, 205-254)
    def _weighted(self, losses_det):
        w = Config.Losses.hparams_losses_weights
        losses_det = dict(losses_det)
        frcnn = 'fasterrcnn' in self.detector_name
        if frcnn:                                                     # train_detector.py:162-164
            losses_det['classification'] = losses_det['loss_classifier']
            losses_det['bbox_regression'] = losses_det['loss_box_reg']
        losses_det['classification'] = losses_det['classification'] * w['det_classification']
        losses_det['bbox_regression'] = losses_det['bbox_regression'] * w['det_regression']
        losses_det['loss_objectness'] = losses_det['loss_objectness'] * w['det_objectness'] if frcnn else 0.0
        losses_det['loss_rpn_box_reg'] = losses_det['loss_rpn_box_reg'] * w['det_rpn_box_reg'] if frcnn else 0.0
        losses_det['bbox_ctrness'] = losses_det['bbox_ctrness'] * w['det_bbox_ctrness'] if 'fcos' in self.detector_name else 0.0   # train_detector.py:174-175
        total = losses_det['bbox_regression'] + losses_det['classification'] + losses_det['loss_objectness'] + \
            losses_det['loss_rpn_box_reg'] + losses_det['bbox_ctrness']
        return total, losses_det

    def _unpack(self, batch):
        if len(batch) == 2:
            imgs, targets = batch
            if self.modality == 'ir':
                imgs = Utils.expand_one_channel_to_output_channels(imgs, 3) if imgs.shape[1] == 1 else imgs
        else:
            imgs, targets, _, _ = batch
        return imgs, Utils.batch_targets_for_detector(targets=targets, device=self.dev, detector_name=self.detector_name)

    def training_step(self, train_batch, batch_idx):
        imgs, targets = self._unpack(train_batch)
        with ops.storage(self.act_dtype):
            losses_det, detections = Detector.calculate_loss(self.detector, imgs, targets, train_det=True, model_name=self.detector_name)
        total_loss, self._last_losses = self._weighted(losses_det)
        self._last_detections = detections
        return total_loss

    def validation_step(self, val_batch, batch_idx):
        imgs, targets = self._unpack(val_batch)
        with torch.no_grad(), ops.storage(self.act_dtype):
            losses_det, detections = Detector.calculate_loss(self.detector, imgs, targets, train_det=False, model_name=self.detector_name)
        self._last_detections = detections
        self._metric('val').update(detections, targets)          # train_detector.py:220
        # train_detector.py:224-229: the validation total is the UNWEIGHTED sum
        if 'fasterrcnn' in self.detector_name:
            return losses_det['loss_box_reg'] + losses_det['loss_classifier'] + losses_det['loss_objectness'] + losses_det['loss_rpn_box_reg']
        return losses_det['bbox_regression'] + losses_det['classification'] + (losses_det['bbox_ctrness'] if 'fcos' in self.detector_name else 0.0)

    def test_step(self, test_batch, batch_idx):
        imgs, targets = self._unpack(test_batch)
        with torch.no_grad(), ops.storage(self.act_dtype):
            _, detections = Detector.calculate_loss(self.detector, imgs, targets, train_det=False, model_name=self.detector_name)
        self._last_detections = detections
        self._metric('test').update(detections, targets)         # train_detector.py:300
        return detections

    def _metric(self, split):
        from .metrics import Detection
        store = self.__dict__.setdefault("_map_metrics", {})
        if split not in store:
            store[split] = Detection(class_metrics=True).map      # train_detector.py:115-116
        return store[split]

    def _epoch_end(self, split):
        m = self._metric(split)
        out = Utils.filter_dictionary(m.compute(), {'map_50', 'map_75', 'map', 'map_per_class'})
        m.reset()
        return out

    def on_validation_epoch_end(self):
        return self._epoch_end('val')

    def on_test_epoch_end(self):
        return self._epoch_end('test')

    # ------------------------------------------------------------------ what Lightning does around training_step
    def fit_step(self, batch, batch_idx=0):
        if self.optimizer is None:
            self.prepare()
        self.detector.train()
        self.detector.invalidate_packs()              # the fp32 masters moved: rebuild the fp16 GEMM layouts
        g = self.arena.flat_grads
        g.zero_()                                     # optimizer_zero_grad (train_detector.py:344-345); kernels accumulate
        with ops.storage(self.act_dtype):
            loss = self.training_step(batch, batch_idx)
            (loss * self.loss_scale).backward()
        if hasattr(self._last_detections, "flush"):          # deferred post-processing: queue it behind the backward pass
            self._last_detections.flush()
        self.averager.begin(g)
        self.averager.start(g)
        self.averager.finish(g)
        self.scaler.step(self.optimizer)
        return loss.detach()
