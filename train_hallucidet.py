"""train_hallucidet.py of the reference (:1-60 setup, :447-548 driver) on the MI355X modules: same flags
(Config.argument_parser), LLVIP / FLIR data modules, fit -> save -> test, the three AP@50 lines of eval_hallucidet.py:180-182.

    python train_hallucidet.py --dataset llvip --train <root>/LLVIP --test <root>/LLVIP --detector fasterrcnn \
        --detector-path detector.bin --batch 8 --precision 16 --epochs 10 --ext .jpg
Multi-GPU: python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_hallucidet.py ...
"""
import os

import torch

from hallucidet_amd.config import Config
from hallucidet_amd.dataloader import MultiModalDataModule
from hallucidet_amd.train_hallucidet import EncoderDecoderLit
from hallucidet_amd.trainer import Trainer


def print_ap50(maps):
    g = lambda k: round(float(maps[k]["map_50"]) * 100, 2)
    print("RGB Detector on IR  AP@50: ", g("map_ir"))
    print("RGB Detector on RGB AP@50: ", g("map_rgb"))
    print("HalluciDet   on IR  AP@50: ", g("map_hall"))


def main(argv=None):
    Config.set_environment()
    args = Config.argument_parser(argv)
    torch.manual_seed(args.seed)
    dataset = args.dataset or "llvip"
    Config.set_detector(args.detector, train_det=False, pretrained=args.directly_coco, dataset=dataset)
    Config.set_loss_weights(args)
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    dev = "cuda:%d" % local
    torch.cuda.set_device(local)
    if world > 1:
        # the gradient arena outlives every collective on it: the allocator need not record the RCCL stream on it (0.1 ms of a 10 ms step at
        # world size 1, tools/probe_dist_host.py)
        os.environ.setdefault("TORCH_NCCL_AVOID_RECORD_STREAMS", "1")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    ext = args.ext or ".jpg"
    dm = MultiModalDataModule(dataset, args.train, args.train, args.test, args.test, batch_size=args.batch, num_workers=args.num_workers,
                              ext=ext, seed=args.seed, rank=rank, world_size=world, ablation_flag=args.ablation_flag)
    kw = dict(batch_size=args.batch, model_name=args.decoder_backbone, in_channels=Config.EncoderDecoder.in_channels_encoder,
              output_channels=Config.EncoderDecoder.out_channels_decoder, lr=1e-4 if args.lr is None else args.lr,
              detector_name=Config.Detector.name, train_det=Config.Detector.train_det, fuse_data=args.fuse_data, precision=args.precision, device=dev)
    model = EncoderDecoderLit.load_from_checkpoint(args.pre_train_path, strict=False, **kw) if args.pre_train_path else EncoderDecoderLit(**kw)
    if args.detector_path:
        from hallucidet_amd.checkpoint import load_detector
        load_detector(model.detector, args.detector_path)
    model.prepare()
    out_dir = os.path.join("lightning_logs", args.wandb_project, args.wandb_name)
    tr = Trainer(max_epochs=args.epochs, limit_train_batches=args.limit_train_batches, dirpath=out_dir if rank == 0 else None,
                 monitor="map_hall/map_50", mode="max", device=dev, log=print if rank == 0 else (lambda *a: None))
    tr.fit(model, dm)
    if rank == 0:
        tr.save_checkpoint(model, os.path.join(out_dir, "encoder_decoder_pl.ckpt"))
        print_ap50(tr.test(model, dm))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
