"""`EncoderDecoderLit` -- the training/eval module of the reference's train_hallucidet.py / eval_hallucidet.py
(train_hallucidet.py:59-445), without the Lightning / wandb hard dependencies (absent from the toolchain).

Kept: constructor keywords, `forward_step` (one U-Net pass, three detector passes, the 11-key loss dict and the 3 output
batches, :161-240), `training_step / validation_step / test_step`, `configure_optimizers`.  The RGB and IR detector passes
run without autograd (their losses are discarded by the reference, App. D.2 -- same results).
"""
import os

import torch
import torch.nn as nn

from . import ops
from .config import Config
from .distributed import GradientAverager, broadcast_parameters, exchange_and_step
from .models.detector import Detector
from .models.encoder_decoder import EncoderDecoder
from .optim import LossScaler
from .utils.utils import Utils
from .utils.eval_forward_fasterrcnn import eval_forward_fasterrcnn_multi, flush_degenerate
from .utils.eval_forward_retinanet import eval_forward_retinanet_multi
from .utils.eval_forward_fcos import eval_forward_fcos_multi


_SEED_WITH_SCALE = os.environ.get("HD_SEED_WITH_SCALE", "1") != "0"     # A/B knob: LossScaler.backward (no scale / seed launches, image gradient read in place)


class _WeightedLosses(torch.autograd.Function):
    """(weighted = stack(losses) * w, total = weighted.sum()) with a ONE-launch backward: d total / d loss_i = w_i, so the incoming
    gradients become one vector and the per-loss gradients are views of it.  Autograd's own backward of stack / mul / sum is seven
    launches of ~5 us each inside a GPU-bound step (train_hallucidet.py:193-210 has one scalar kernel per key and per '+')."""

    @staticmethod
    def forward(ctx, wvec, *vals):
        weighted = torch.stack(vals) * wvec
        ctx.save_for_backward(wvec)
        ctx.set_materialize_grads(False)          # an unused output arrives as None, not as a zero tensor to add
        return weighted, weighted.sum()

    @staticmethod
    def backward(ctx, g_weighted, g_total):
        wvec, = ctx.saved_tensors
        if g_weighted is None and g_total is None:
            return (None,) * (1 + wvec.shape[0])
        g = wvec * g_total if g_weighted is None else (wvec * g_weighted if g_total is None else wvec * (g_weighted + g_total))
        return (None,) + tuple(g[i] for i in range(g.shape[0]))


class EncoderDecoderLit(nn.Module):
    def __init__(self, batch_size=4, wandb_logger=None, model_name='resnet34', in_channels=3, output_channels=3, lr=0.0001,
                 loss_pixel=None, loss_perceptual=None, detector_name='fasterrcnn', train_det=False, fuse_data='none',
                 scheduler_on=False, detector=None, precision=16, device='cuda', use_graphs=True):
        super().__init__()
        self.model_name, self.wandb_logger = model_name, wandb_logger
        self.in_channels, self.output_channels = in_channels, output_channels
        self.lr, self.batch_size, self.train_det, self.fuse_data = lr, batch_size, train_det, fuse_data
        self.optimizer_name = Config.Optimizer.name
        self.segmentation_head = Config.EncoderDecoder.decoder_head
        self.scheduler_on = scheduler_on
        self.detector_name = detector_name
        self.dev = torch.device(device)
        if loss_pixel is not None or loss_perceptual is not None:
            raise NotImplementedError("pixel / LPIPS losses have weight 0.0 and selectors returning None in every BASELINE "
                                      "config (SURVEY #17); their four keys are reported as 0.0")
        self.encoder_decoder = EncoderDecoder(name=self.model_name, encoder_depth=5, encoder_weights=None,
                                              decoder_attention_type=None, in_channels=self.in_channels,
                                              output_channels=self.output_channels,
                                              segmentation_head=Config.EncoderDecoder.decoder_head).encoder_decoder
        self.detector = detector if detector is not None else Detector(name=detector_name, pretrained=False, n_classes=2,
                                                                       size=Config.Detector.input_size).detector
        self.detector.eval()
        for p in self.detector.parameters():
            p.requires_grad = False
        # precision 16: fp16 storage + dynamic loss scaling (BASELINE configs[1]); 32: fp32 storage, no scaling -- the arithmetic of
        # the reference's default `--precision 32` (src/config/config.py:149 -> pl.Trainer, train_hallucidet.py:507): every kernel of
        # the step in its _f32 form (ops.storage); a parity mode, not a fast path
        if int(precision) not in (16, 32):
            raise ValueError("precision must be 16 or 32 (got %r)" % (precision,))
        self.precision = int(precision)
        self.act_dtype = torch.float32 if self.precision == 32 else torch.float16
        self.encoder_decoder.runner.set_precision(self.precision)
        self.use_graphs = use_graphs
        self.batch_detector_passes = True
        # The reference evaluates the RGB and IR detector passes in EVERY training step (train_hallucidet.py:183,186) and uses
        # their detections only in validation: two thirds of the detector forward of a training step are dead work.  False
        # (default) reproduces the reference's step exactly -- and is what bench.py measures; True skips the two passes when
        # step == 'train' (their detections are then empty lists).  Opt-in: it changes the work per step, not the result.
        self.skip_unused_train_passes = False
        self.detector.fused_passes = True    # one head evaluation for the three passes (see eval_forward_fasterrcnn._multi_fused)
        self.scaler = None
        self.optimizer = None
        self.averager = GradientAverager()
        self.overlap_allreduce = True        # False: one exchange after the whole backward (A/B knob)
        self.use_detector_graph = os.environ.get("HD_DET_GRAPH", "1") != "0"

    # ------------------------------------------------------------------------------------------------------------
    def forward_step(self, imgs_rgb, targets_rgb, imgs_ir, targets_ir, batch_idx, step='train'):
        with ops.storage(self.act_dtype):
            return self._forward_step(imgs_rgb, targets_rgb, imgs_ir, targets_ir, batch_idx, step)

    def _forward_step(self, imgs_rgb, targets_rgb, imgs_ir, targets_ir, batch_idx, step='train'):
        device = self.dev
        imgs_ir = Utils.batch_images_for_encoder_decoder(imgs=imgs_ir, device=device)
        imgs_rgb = Utils.batch_images_for_encoder_decoder(imgs=imgs_rgb, device=device)
        targets_rgb = Utils.batch_targets_for_detector(targets=targets_rgb, device=device, detector_name=self.detector_name)
        targets_ir = Utils.batch_targets_for_detector(targets=targets_ir, device=device, detector_name=self.detector_name)

        imgs_ir_three_channel = Utils.expand_one_channel_to_output_channels(imgs_ir, self.output_channels) if imgs_ir.shape[1] == 1 else imgs_ir
        imgs_hallucinated = self.encoder_decoder(imgs_ir_three_channel)

        loss_pixel_rgb = loss_perceptual_rgb = loss_pixel_ir = loss_perceptual_ir = 0.0

        train_det = True if (self.train_det is True and step == 'train') else False
        graph = self._detector_graph() if (step == 'train' and not train_det and imgs_hallucinated.requires_grad) else None
        if graph is not None:
            # the detector half of the step (three passes, losses, the backward pass down to the hallucinated image, the deferred
            # post-processing) as ONE hipGraph replay (det_graph.py); the U-Net's backward continues from the image gradient
            losses_det, loss_det_total, (detections_hall, detections_rgb, detections_ir) = graph.step(
                imgs_hallucinated, imgs_rgb, imgs_ir_three_channel, targets_rgb, targets_ir)
        else:
            losses_det, loss_det_total, (detections_hall, detections_rgb, detections_ir) = self._detector_section(
                imgs_hallucinated, imgs_rgb, imgs_ir_three_channel, targets_rgb, targets_ir, step, train_det)
        total_loss = loss_det_total
        for extra in (loss_pixel_rgb, loss_perceptual_rgb, loss_pixel_ir, loss_perceptual_ir):
            if torch.is_tensor(extra) or extra != 0.0:
                total_loss = total_loss + extra

        self._last_detections = dict(hall=detections_hall, rgb=detections_rgb, ir=detections_ir)
        return {
            'loss': {'total': total_loss, 'pixel_rgb': loss_pixel_rgb, 'perceptual_rgb': loss_perceptual_rgb,
                     'pixel_ir': loss_pixel_ir, 'perceptual_ir': loss_perceptual_ir,
                     'det_regression': losses_det['bbox_regression'], 'det_classification': losses_det['classification'],
                     'det_objectness': losses_det['loss_objectness'], 'det_rpn_box_reg': losses_det['loss_rpn_box_reg'],
                     'det_bbox_ctrness': losses_det['bbox_ctrness'], 'det_total': loss_det_total},
            # train_hallucidet.py:218 min-max normalises a detached clone for plotting only; kept lazy
            'output': {'imgs_rgb': imgs_rgb, 'imgs_ir': imgs_ir, 'imgs_hallucinated': imgs_hallucinated.detach()},
        }

    def _detector_section(self, imgs_hallucinated, imgs_rgb, imgs_ir_three_channel, targets_rgb, targets_ir, step, train_det,
                          targets_ir_pass=None):
        """train_hallucidet.py:180-210: the three detector passes and the weighted detector losses.
        -> (losses_det with the weighted keys, their sum, (detections_hall, detections_rgb, detections_ir)).
        `targets_ir_pass`: the IR pass's own copy of the IR targets (det_graph.py stages one, so that the three passes' targets are
        consecutive rows of one buffer); default: the same list as the hallucinated pass."""
        t_ir3 = targets_ir if targets_ir_pass is None else targets_ir_pass
        if step == 'train' and self.skip_unused_train_passes and not train_det:
            losses_det, detections_hall = Detector.calculate_loss(self.detector, imgs_hallucinated, targets_ir, train_det=False, model_name=self.detector_name)
            detections_rgb, detections_ir = [], []
        elif self.batch_detector_passes and not train_det and 'fasterrcnn' in self.detector_name:
            # one trunk evaluation for the three passes (frozen, eval-mode detector: images are independent); the RGB / IR
            # losses are discarded by the reference (train_hallucidet.py:183,186) and carry no gradient
            (losses_det, detections_hall), (_, detections_rgb), (_, detections_ir) = eval_forward_fasterrcnn_multi(
                self.detector, [imgs_hallucinated, imgs_rgb, imgs_ir_three_channel], [targets_ir, targets_rgb, t_ir3])
        elif self.batch_detector_passes and not train_det and 'retinanet' in self.detector_name:
            (losses_det, detections_hall), (_, detections_rgb), (_, detections_ir) = eval_forward_retinanet_multi(
                self.detector, [imgs_hallucinated, imgs_rgb, imgs_ir_three_channel], [targets_ir, targets_rgb, t_ir3])
        elif self.batch_detector_passes and not train_det and 'fcos' in self.detector_name:
            (losses_det, detections_hall), (_, detections_rgb), (_, detections_ir) = eval_forward_fcos_multi(
                self.detector, [imgs_hallucinated, imgs_rgb, imgs_ir_three_channel], [targets_ir, targets_rgb, t_ir3])
        else:
            losses_det, detections_hall = Detector.calculate_loss(self.detector, imgs_hallucinated, targets_ir, train_det=train_det, model_name=self.detector_name)
            with torch.no_grad():
                _, detections_rgb = Detector.calculate_loss(self.detector, imgs_rgb, targets_rgb, train_det=False, model_name=self.detector_name)
                _, detections_ir = Detector.calculate_loss(self.detector, imgs_ir_three_channel, targets_ir, train_det=False, model_name=self.detector_name)

        w = Config.Losses.hparams_losses_weights
        if 'fasterrcnn' in self.detector_name:
            losses_det['classification'] = losses_det['loss_classifier']
            losses_det['bbox_regression'] = losses_det['loss_box_reg']
        # train_hallucidet.py:193-210: every detector loss times its weight, their sum, plus the (weight 0.0) pixel / perceptual terms.
        # One stack, one multiply by a cached weight vector, one sum (the per-key results are views of the product) instead of a
        # scalar kernel per key and per '+': each of those is a serialised ~3 us launch in a step that is GPU-bound.
        keys = self._loss_keys()
        vals = [losses_det[k] for k, _ in keys]
        wkey = (tuple(float(w[wk]) for _, wk in keys), str(vals[0].device))
        wvec = self._wvec_cache.get(wkey) if hasattr(self, "_wvec_cache") else None
        if wvec is None:
            self.__dict__.setdefault("_wvec_cache", {})[wkey] = wvec = torch.tensor(wkey[0], dtype=torch.float32, device=vals[0].device)
        weighted, total = _WeightedLosses.apply(wvec, *[v.reshape(()).float() for v in vals])
        for i, (k, _) in enumerate(keys):
            losses_det[k] = weighted[i]
        for k in ('loss_objectness', 'loss_rpn_box_reg', 'bbox_ctrness'):
            if not any(k == kk for kk, _ in keys):
                losses_det[k] = 0.0
        return losses_det, total, (detections_hall, detections_rgb, detections_ir)

    def _loss_keys(self):
        frcnn, fcos_ = 'fasterrcnn' in self.detector_name, 'fcos' in self.detector_name
        keys = [('bbox_regression', 'det_regression'), ('classification', 'det_classification')]
        keys += [('loss_objectness', 'det_objectness'), ('loss_rpn_box_reg', 'det_rpn_box_reg')] if frcnn else []
        keys += [('bbox_ctrness', 'det_bbox_ctrness')] if fcos_ else []
        return keys

    def _detector_graph(self):
        """The captured detector half (det_graph.DetectorStepGraph) when it applies: graphs enabled, frozen detector, the fused
        three-pass evaluation, device tensors.  HD_DET_GRAPH=0 keeps the eager issue order (A/B knob)."""
        if not (self.use_graphs and self.use_detector_graph and self.batch_detector_passes and not self.skip_unused_train_passes
                and self.dev.type == "cuda" and self.scaler is not None):
            return None
        g = self.__dict__.get("_det_graph")
        if g is None:
            from .det_graph import DetectorStepGraph
            g = self.__dict__["_det_graph"] = DetectorStepGraph(self)
        return g if g.usable else None

    def training_step(self, train_batch, batch_idx):
        imgs_rgb, targets_rgb, imgs_ir, targets_ir = train_batch
        out = self.forward_step(imgs_rgb, targets_rgb, imgs_ir, targets_ir, batch_idx, step='train')
        return out['loss']['total']

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, strict=True, **kwargs):
        """train_hallucidet.py:467-481 / eval_hallucidet.py:199 (Lightning's classmethod): construct with the given
        keywords, then restore `encoder_decoder.*` and `detector.*` from the checkpoint's state_dict."""
        from .checkpoint import load_encoder_decoder_lit
        return load_encoder_decoder_lit(cls(**kwargs), checkpoint_path, strict=strict)

    def save_checkpoint(self, path, epoch=0, global_step=0):
        """trainer.save_checkpoint (:353-356, :544-545)."""
        from .checkpoint import save_lightning_checkpoint
        return save_lightning_checkpoint(path, {"encoder_decoder": self.encoder_decoder, "detector": self.detector}, epoch, global_step)

    def _metrics(self, split):
        """train_hallucidet.py:121-131: one COCO-style mAP accumulator per (split, stream)."""
        from .metrics import Detection
        store = self.__dict__.setdefault("_map_metrics", {})
        if split not in store:
            store[split] = {k: Detection().map for k in ("hall", "rgb", "ir")}
        return store[split]

    def _eval_step(self, batch, batch_idx, split):
        imgs_rgb, targets_rgb, imgs_ir, targets_ir = batch
        with torch.no_grad():
            out = self.forward_step(imgs_rgb, targets_rgb, imgs_ir, targets_ir, batch_idx, step=split)
        d, m = self._last_detections, self._metrics(split)
        m["rgb"].update(d["rgb"], targets_rgb)          # :213-215 / :399-401
        m["hall"].update(d["hall"], targets_ir)
        m["ir"].update(d["ir"], targets_ir)
        return out['loss']['total'], d

    def validation_step(self, val_batch, batch_idx):
        return self._eval_step(val_batch, batch_idx, 'val')

    def test_step(self, test_batch, batch_idx):
        return self._eval_step(test_batch, batch_idx, 'test')

    def _epoch_end(self, split):
        m = self._metrics(split)
        out = {"map_" + k: Utils.filter_dictionary(m[k].compute(), {'map_50', 'map_75', 'map'}) for k in ("rgb", "hall", "ir")}
        for v in m.values():
            v.reset()
        return out

    def on_validation_epoch_end(self):
        """:328-362 without the wandb / checkpoint side effects: returns {'map_rgb','map_hall','map_ir'} -> {map, map_50, map_75}."""
        return self._epoch_end('val')

    def on_test_epoch_end(self):
        """:412-427"""
        return self._epoch_end('test')

    def configure_optimizers(self):
        self.encoder_decoder.to(self.dev)
        self.optimizer = Config.config_optimizer(self.encoder_decoder, learning_rate=self.lr, name=self.optimizer_name)
        # precision 16 = Lightning's native AMP: GradScaler policy (fp16 gradient maps underflow without it); precision 32: fp32
        # storage end to end, no scaler (scale 1.0, no overflow check), as pl.Trainer(precision=32) steps the optimizer
        self.scaler = LossScaler(self.encoder_decoder, enabled=self.precision == 16)
        # train_hallucidet.py:436-444: ReduceLROnPlateau(optimizer, mode='min') monitored on val_loss (torch defaults:
        # factor 0.1, patience 10); the fused optimizer reads param_groups[0]['lr'] at every step
        self.lr_scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode='min')
        return {"optimizer": self.optimizer, "lr_scheduler": {"scheduler": self.lr_scheduler, "monitor": "val_loss"}}

    def lr_scheduler_step(self, val_loss):
        """What Lightning does with the monitored metric at the end of a validation epoch."""
        self.lr_scheduler.step(float(val_loss))
        return self.optimizer.param_groups[0]["lr"]

    # ------------------------------------------------------------------------------------------------------------
    def prepare(self):
        """Move to the device, build the optimizer, and (data parallel) start every rank from rank 0's weights."""
        self.to(self.dev)
        self.configure_optimizers()
        r = self.encoder_decoder.runner
        r.enable_graphs(self.use_graphs)
        bufs = [b for b in self.encoder_decoder.buffers() if b.dtype.is_floating_point]
        broadcast_parameters(r.flat_params, bufs)
        return self

    def eval(self):
        """Lightning runs validation / test under `model.eval()`: U-Net BatchNorm on running statistics (and no buffer
        updates), detector in eval mode."""
        return super().eval()

    def fit_step(self, batch, batch_idx=0):
        """What Lightning does around training_step: scale -> backward -> all-reduce -> (unscale+clip+Adam fused)."""
        with ops.storage(self.act_dtype):
            return self._fit_step(batch, batch_idx)

    def _fit_step(self, batch, batch_idx=0):
        self.encoder_decoder.train()
        loss = self.training_step(batch, batch_idx)
        r = self.encoder_decoder.runner
        g = r.flat_grads
        from .distributed import is_dist
        # data parallel: the runner reports finished arena slices during backward (decoder first), their all-reduces overlap the
        # rest of the backward pass; start() covers what no hook reported, finish() waits right before the optimizer
        r.bucket_hook = self.averager.bucket_ready if (is_dist() and self.overlap_allreduce) else None
        self.averager.begin(g)
        if _SEED_WITH_SCALE:
            self.scaler.backward(loss)
        else:
            self.scaler.scale(loss).backward()
        for d in (self._last_detections or {}).values():       # deferred detector post-processing: queue it behind the backward pass
            if hasattr(d, "flush"):
                d.flush()
        flush_degenerate(self.detector, block=False)      # the reference's degenerate-box assertion, before the update when it is known by now
        exchange_and_step(self.averager, g, self.scaler, self.optimizer)
        return loss.detach()
