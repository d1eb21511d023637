"""Operator API, factory 1: `EncoderDecoder(...).encoder_decoder` (reference src/models/encoder_decoder.py:8-45).

Same constructor keywords, same attribute, same head swap (`segmentation_head[-1] = Sigmoid`); the module it builds
runs on libhallucidet_hip.so (hallucidet_amd.segmentation_models.Unet)."""
import torch

from .. import segmentation_models as smp


class EncoderDecoder():
    def __init__(self, name='resnet34', encoder_depth=5, encoder_weights=None, decoder_attention_type=None,
                 in_channels=3, output_channels=3, segmentation_head='sigmoid', dropout=0.2, avg2d_flag=True):
        self.encoder_decoder = smp.Unet(name,
                                        encoder_depth=encoder_depth,
                                        encoder_weights=encoder_weights,
                                        decoder_attention_type=decoder_attention_type,
                                        in_channels=in_channels,
                                        classes=output_channels)
        if segmentation_head == 'sigmoid':
            self.encoder_decoder.segmentation_head[-1] = torch.nn.Sigmoid()
        else:
            # 'relu_bn' / 'avg_dropout_sigmoid' exist in the reference (encoder_decoder.py:32-45) but every script forces
            # Config.EncoderDecoder.decoder_head = 'sigmoid' (config.py:78); they are outside the hot path.
            raise NotImplementedError("hallucidet_amd EncoderDecoder: segmentation_head=%r is not on the hot path" % (segmentation_head,))
