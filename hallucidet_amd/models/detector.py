"""Operator API, factory 2 + the operator itself (reference src/models/detector.py:24-141).

`Detector(name, pretrained, n_classes, size, ...)` builds the detector (`.detector`), swaps in the fixed-size transform
and re-heads the box predictor to `n_classes`; `Detector.calculate_loss(detector, outs, targets, train_det, model_name)`
returns `(losses: dict[str, 0-dim Tensor], detections: list[dict(boxes, labels, scores)])` in one pass.
"""
import math
import sys

import torch

from . import detection, fcos, retinanet
from .custom_generalized_transform import CustomGeneralizedRCNNTransform
from ..utils.eval_forward_fasterrcnn import eval_forward_fasterrcnn
from ..utils.eval_forward_retinanet import eval_forward_retinanet
from ..utils.eval_forward_fcos import eval_forward_fcos


def _xavier_init(conv: torch.nn.Module):
    for layer in conv.modules():
        if isinstance(layer, torch.nn.Conv2d):
            torch.nn.init.xavier_uniform_(layer.weight)
            if layer.bias is not None:
                torch.nn.init.constant_(layer.bias, 0.0)


class Detector():
    def __init__(self, name='fasterrcnn_resnet50_fpn', pretrained=True, n_classes=2, size=300, batch_norm_eps=0.001,
                 batch_norm_momentum=0.03, eval_path=None, modality=None, directly_coco=False):
        self.detector = Detector.select_detector(detector_name=name, pretrained=pretrained)
        if not directly_coco:
            self.detector.transform = self.change_generalized_transform(min_size=size, max_size=size, image_mean=[0.0],
                                                                        image_std=[1.0], size_divisible=1,
                                                                        fixed_size=(size, size))
            if 'fasterrcnn' in name:
                in_features = self.detector.roi_heads.box_predictor.cls_score.in_features
                self.detector.roi_heads.box_predictor = detection.FastRCNNPredictor(in_features, n_classes)
                _xavier_init(self.detector.roi_heads)
            elif 'fcos' in name or 'retinanet' in name:
                head = self.detector.head.classification_head
                out_channels = head.conv[0].out_channels
                num_anchors = head.num_anchors
                head.num_classes = n_classes
                cls_logits = torch.nn.Conv2d(out_channels, num_anchors * n_classes, kernel_size=3, stride=1, padding=1)
                torch.nn.init.normal_(cls_logits.weight, std=0.01)
                torch.nn.init.constant_(cls_logits.bias, -math.log((1 - 0.01) / 0.01))
                head.cls_logits = cls_logits
                self.detector.head.invalidate()
            if eval_path is not None and '.bin' in eval_path:
                self.detector.load_state_dict(torch.load(eval_path, map_location="cpu"))
            elif eval_path is not None and '.ckpt' in eval_path:
                eval_path = (eval_path.split('.ckpt')[0] + '.bin').replace('best', 'detector')
                try:
                    self.detector.load_state_dict(torch.load(eval_path, map_location="cpu"))
                except Exception:
                    print("Select model is not compatible with the detector (Requires: .bin dict)")

    def change_generalized_transform(self, min_size=300, max_size=300, image_mean=[0.0], image_std=[1.0], size_divisible=1,
                                     fixed_size=(300, 300)):
        return CustomGeneralizedRCNNTransform(min_size=min_size, max_size=max_size, image_mean=image_mean, image_std=image_std,
                                              size_divisible=size_divisible, fixed_size=fixed_size)

    @staticmethod
    def calculate_loss(detector, outs, targets, train_det=False, model_name='fasterrcnn'):
        if 'fasterrcnn' in model_name:
            losses_det, detections = eval_forward_fasterrcnn(detector, outs, targets, train_det=train_det, model_name=model_name)
        elif 'retinanet' in model_name:
            losses_det, detections = eval_forward_retinanet(detector, outs, targets, train_det=train_det, model_name=model_name)
        elif 'fcos' in model_name:
            losses_det, detections = eval_forward_fcos(detector, outs, targets, train_det=train_det, model_name=model_name)
        else:
            raise ValueError("unknown detector %r" % (model_name,))
        return losses_det, detections

    @staticmethod
    def select_detector(detector_name='fasterrcnn_resnet50_fpn', pretrained=True):
        if detector_name in ('fasterrcnn', 'fasterrcnn_resnet50_fpn'):
            return detection.fasterrcnn_resnet50_fpn(pretrained=pretrained)
        if detector_name in ('retinanet', 'retinanet_resnet50_fpn'):
            return retinanet.retinanet_resnet50_fpn(pretrained=pretrained)
        if detector_name in ('fcos', 'fcos_resnet50_fpn'):
            return fcos.fcos_resnet50_fpn(pretrained=pretrained)
        print("Model Name not found (Using fasterrcnn_resnet50_fpn")
        return detection.fasterrcnn_resnet50_fpn(pretrained=pretrained)
