"""Subset of the reference's global `Config` that the hot path reads (src/config/config.py:6-93,204-245,310-357):
same attribute names and default values (pinned by tests/golden/config_defaults.json)."""
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
    def config_optimizer(unet, learning_rate=1e-4, name='adam'):
        """config.py:204-245 builds torch.optim.Adam(params, lr); here the same update runs as one fused kernel over the
        flat parameter arena, with Lightning's value clipping (Config.Optimizer.gradient_clip_val) folded in."""
        from .optim import FusedAdam
        if name != 'adam':
            raise NotImplementedError("the reference always uses Config.Optimizer.name == 'adam' (SURVEY 5.6)")
        return FusedAdam(unet, lr=learning_rate, clip_value=Config.Optimizer.gradient_clip_val)

    @staticmethod
    def cuda_or_cpu():
        return 'cuda' if torch.cuda.is_available() else 'cpu'
