"""eval_hallucidet.py of the reference (:190-230): load a HalluciDet checkpoint, run the test split, print the three AP@50
lines (:180-182).

    python eval_hallucidet.py --dataset llvip --test <root>/LLVIP --hallucidet-path best.ckpt --detector fasterrcnn --batch 8 --ext .jpg
"""
import torch

from hallucidet_amd.config import Config
from hallucidet_amd.dataloader import MultiModalDataModule
from hallucidet_amd.train_hallucidet import EncoderDecoderLit
from hallucidet_amd.trainer import Trainer
from train_hallucidet import print_ap50


def main(argv=None):
    Config.set_environment()
    args = Config.argument_parser(argv)
    torch.manual_seed(args.seed)
    dataset = args.dataset or "llvip"
    Config.set_detector(args.detector, train_det=False, pretrained=args.directly_coco, dataset=dataset)
    dev = args.device if args.device not in (None, "gpu") else "cuda"
    dm = MultiModalDataModule(dataset, args.test, args.test, args.test, args.test, batch_size=args.batch, num_workers=args.num_workers,
                              ext=args.ext or ".jpg", seed=args.seed)
    kw = dict(batch_size=args.batch, model_name=args.decoder_backbone, detector_name=Config.Detector.name, precision=args.precision, device=dev)
    model = EncoderDecoderLit.load_from_checkpoint(args.hallucidet_path, strict=False, **kw) if args.hallucidet_path else EncoderDecoderLit(**kw)
    if args.detector_path:
        from hallucidet_amd.checkpoint import load_detector
        load_detector(model.detector, args.detector_path)
    model.encoder_decoder.to(dev)
    model.detector.to(dev)
    model.eval()            # Lightning's test loop: BatchNorm on the checkpoint's running statistics, detector in eval mode
    maps = Trainer(device=dev).test(model, dm)
    print_ap50(maps)
    return maps


if __name__ == "__main__":
    main()
