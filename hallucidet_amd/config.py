"""Subset of the reference's global `Config` that the hot path reads (src/config/config.py:6-93,204-245,310-357):
same attribute names and default values (pinned by tests/golden/config_defaults.json)."""
import os

import torch


class Config:

    class Environment:
        N_CORE = "8"
        N_THREADS_TORCH = 8
        N_GPUS = 1
        CUDNN_BENCHMARK = True
        DEBUG = False

    class Optimizer:
        name = 'adam'
        scheduler_step_size = 10
        scheduler_gamma = 0.1
        scheduler_on = True
        gradient_clip_val = 0.5

    class Losses:
        hparams_losses_weights = {
            'pixel_rgb': 0.0,
            'pixel_ir': 0.0,
            'perceptual_rgb': 0.0,
            'perceptual_ir': 0.0,
            'det_regression': 0.1,
            'det_classification': 0.1,
            'det_objectness': 0.1,
            'det_rpn_box_reg': 0.1,
            'det_bbox_ctrness': 0.1,
            'det_masked': 0.0,
        }
        pixel = None
        perceptual = None

    class EncoderDecoder:
        in_channels_encoder = 3
        out_channels_decoder = 3
        decoder_head = 'sigmoid'
        load_encoder_decoder = False

    class Detector:
        train_det = False
        name = 'fasterrcnn'
        pretrained = True
        input_size = 300   # 640 for flir (config.py:317)
        batch_norm_eps = 0.001
        batch_norm_momentum = 0.03
        eval_path = None
        modality = None
        score_threshold = 0.5

    @staticmethod
    def set_detector(detector_name='fasterrcnn', train_det=False, pretrained=True, dataset='llvip'):
        Config.Detector.name = detector_name
        Config.Detector.train_det = train_det
        Config.Detector.pretrained = pretrained
        Config.Detector.input_size = 640 if dataset == 'flir' else 300

    @staticmethod
    def set_environment():
        """config.py:262-271: eight host threads for OpenMP / BLAS / torch's intra-op pool (cudnn.benchmark has no counterpart
        here).  Matters on the GPU path too: torch's default is one thread per HARDWARE thread of the host (256 on the MI355X
        boxes) while a container is throttled to a fraction of them -- the step's small host-side tensor ops then spin against
        each other (measured: mean step 14.5-16.7 ms instead of 14.0-14.2 ms)."""
        for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "VECLIB_MAXIMUM_THREADS", "NUMEXPR_NUM_THREADS"):
            os.environ[k] = Config.Environment.N_CORE
        torch.set_num_threads(Config.Environment.N_THREADS_TORCH)

    @staticmethod
    def config_optimizer(unet, learning_rate=1e-4, name='adam'):
        """config.py:204-245 builds torch.optim.Adam(params, lr); here the same update runs as one fused kernel over the
        flat parameter arena, with Lightning's value clipping (Config.Optimizer.gradient_clip_val) folded in."""
        from .optim import FusedAdam
        if name != 'adam':
            raise NotImplementedError("the reference always uses Config.Optimizer.name == 'adam' (SURVEY 5.6)")
        return FusedAdam(unet, lr=learning_rate, clip_value=Config.Optimizer.gradient_clip_val)

    @staticmethod
    def argument_parser(argv=None):
        """config.py:96-197 -- the reference's flags and defaults (one parser for the three scripts)."""
        import argparse
        p = argparse.ArgumentParser(description='HalluciDet')
        p.add_argument('--dataset', type=str, default=None, help='llvip/flir')
        p.add_argument('--train', type=str, default=None, help='Train Dataset Path')
        p.add_argument('--valid', type=str, default=None, help='Valid Dataset Path')
        p.add_argument('--test', type=str, default=None, help='Test Dataset Path')
        p.add_argument('--n-classes', '--n_classes', '--num-classes', '--nclasses', type=int, default=2)
        p.add_argument('--detector', type=str, default='fasterrcnn', help="choices=['fasterrcnn', 'fcos', 'retinanet']")
        p.add_argument('--pretrained', action='store_true')
        p.add_argument('--fine-tuning', action='store_true')
        p.add_argument('--fine-tuning-lp', action='store_true')
        p.add_argument('--modality', type=str, default='rgb')
        p.add_argument('--threshold', type=float, default=0.5)
        p.add_argument('--epochs', type=int, default=10)
        p.add_argument('--lr', type=float, default=None)
        p.add_argument('--seed', type=int, default=123)
        p.add_argument('--wandb-project', type=str, default="hallucidet")
        p.add_argument('--wandb-name', type=str, default="detector")
        p.add_argument("--batch", type=int, default=16)
        p.add_argument("--num-workers", type=int, default=4)
        p.add_argument("--ext", "--input-ext", type=str, default=None)
        p.add_argument("--output-model", type=str, default="example.ckpt")
        p.add_argument("--detector-path", type=str, default=None)
        p.add_argument("--device", type=str, default=None)
        p.add_argument("--fuse-data", type=str, default='none')
        p.add_argument("--decoder-backbone", type=str, default='resnet34')
        p.add_argument("--precision", type=int, default=32)
        p.add_argument("--optimizer", type=str, default='adamw')
        p.add_argument("--path", type=str, default=None)
        p.add_argument("--segmentation-head", type=str, default='sigmoid')
        p.add_argument("--pixel", type=str, default=None)
        p.add_argument("--weight-pixel-rgb", type=float, default=0.0)
        p.add_argument("--weight-pixel-ir", type=float, default=0.0)
        p.add_argument("--perceptual", type=str, default=None)
        p.add_argument("--weight-perceptual-rgb", type=float, default=0.0)
        p.add_argument("--weight-perceptual-ir", type=float, default=0.0)
        p.add_argument("--weight-det-regression", type=float, default=0.1)
        p.add_argument("--weight-det-classification", type=float, default=0.1)
        p.add_argument("--weight-det-masked", type=float, default=0.0)
        p.add_argument("--weight-det-objectness", type=float, default=0.1)
        p.add_argument("--weight-det-rpn-box-reg", type=float, default=0.1)
        p.add_argument("--weight-det-bbox-ctrness", type=float, default=0.1)
        p.add_argument("--image2image-model", type=str, default=None)
        p.add_argument('--directly-coco', action='store_true')
        p.add_argument('--limit-train-batches', type=float, default=1.0)
        p.add_argument('--ablation-flag', action='store_true')
        p.add_argument("--pre-train-path", type=str, default=None)
        p.add_argument("--encoder-depth", type=int, default=5)
        p.add_argument('--hallucidet-path', type=str)
        return p.parse_args(argv)

    @staticmethod
    def set_loss_weights(args):
        """config.py: `set_loss_weights(args)` copies the --weight-* flags into Losses.hparams_losses_weights."""
        if args.pixel is not None:
            Config.Losses.pixel = args.pixel
        if args.perceptual is not None:
            Config.Losses.perceptual = args.perceptual
        w = Config.Losses.hparams_losses_weights
        w.update(pixel_rgb=args.weight_pixel_rgb, pixel_ir=args.weight_pixel_ir, perceptual_rgb=args.weight_perceptual_rgb,
                 perceptual_ir=args.weight_perceptual_ir, det_regression=args.weight_det_regression,
                 det_classification=args.weight_det_classification, det_objectness=args.weight_det_objectness,
                 det_rpn_box_reg=args.weight_det_rpn_box_reg, det_bbox_ctrness=args.weight_det_bbox_ctrness, det_masked=args.weight_det_masked)

    @staticmethod
    def cuda_or_cpu():
        return 'cuda' if torch.cuda.is_available() else 'cpu'
