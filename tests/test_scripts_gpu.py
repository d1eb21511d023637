"""The three reference scripts (train_hallucidet.py / eval_hallucidet.py / train_detector.py) end to end on a synthetic
LLVIP tree: flags of Config.argument_parser, data modules, the minimal Trainer driving the hooks, Lightning-layout
checkpoints, the AP@50 lines of eval_hallucidet.py:180-182."""
import os
import sys

import pytest
import torch

from _synth_llvip import make_tree

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_eval_and_detector_scripts(dev, tmp_path, capsys, monkeypatch):
    sys.path.insert(0, ROOT)
    monkeypatch.chdir(tmp_path)
    root = make_tree(tmp_path, n_train=10, n_test=4, hw=(64, 96), extra_objects=False)
    import train_hallucidet, eval_hallucidet, train_detector
    common = ["--dataset", "llvip", "--train", root, "--test", root, "--ext", ".jpg", "--batch", "2", "--num-workers", "0", "--seed", "3"]
    train_hallucidet.main(common + ["--detector", "fasterrcnn", "--epochs", "2", "--precision", "16", "--wandb-name", "t1"])
    out = capsys.readouterr().out
    assert "HalluciDet   on IR  AP@50:" in out and "RGB Detector on RGB AP@50:" in out and "epoch 1 " in out
    ck = os.path.join("lightning_logs", "hallucidet", "t1", "encoder_decoder_pl.ckpt")
    assert os.path.isfile(ck)
    sd = torch.load(ck, map_location="cpu", weights_only=False)["state_dict"]
    assert "encoder_decoder.encoder.conv1.weight" in sd and "detector.rpn.head.conv.weight" in sd
    maps = eval_hallucidet.main(["--dataset", "llvip", "--test", root, "--ext", ".jpg", "--batch", "2", "--num-workers", "0",
                                 "--hallucidet-path", ck, "--precision", "16"])
    out = capsys.readouterr().out
    assert out.count("AP@50") == 3 and set(maps) == {"map_rgb", "map_hall", "map_ir"}
    train_detector.main(common + ["--detector", "fasterrcnn", "--modality", "rgb", "--epochs", "1", "--wandb-name", "t2"])
    out = capsys.readouterr().out
    assert "test:" in out and "map_50" in out
