"""train_detector.py of the reference (:22-60 setup, :347-420 driver): fine-tune the detector on one modality.

    python train_detector.py --dataset llvip --train <root>/LLVIP --test <root>/LLVIP --detector fasterrcnn --modality rgb --batch 16 --epochs 200 --ext .jpg
"""
import os

import torch

from hallucidet_amd.config import Config
from hallucidet_amd.dataloader import SingleModalDataModule
from hallucidet_amd.train_detector import DetectorLit
from hallucidet_amd.trainer import Trainer


def main(argv=None):
    Config.set_environment()
    args = Config.argument_parser(argv)
    torch.manual_seed(args.seed)
    dataset = args.dataset or "llvip"
    Config.set_detector(args.detector, train_det=False, pretrained=False, dataset=dataset)
    Config.set_loss_weights(args)
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    dev = "cuda:%d" % local
    torch.cuda.set_device(local)
    if world > 1:
        # the gradient arena outlives every collective on it: the allocator need not record the RCCL stream on it (0.1 ms of a 10 ms step at
        # world size 1, tools/probe_dist_host.py)
        os.environ.setdefault("TORCH_NCCL_AVOID_RECORD_STREAMS", "1")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    dm = SingleModalDataModule(dataset, args.train, args.test, batch_size=args.batch, num_workers=args.num_workers, ext=args.ext or ".jpg",
                               seed=args.seed, rank=rank, world_size=world, modality=args.modality)
    kw = dict(batch_size=args.batch, lr=1e-4 if args.lr is None else args.lr, detector_name=Config.Detector.name, pretrained=args.pretrained,
              modality=args.modality, directly_coco=args.directly_coco, device=dev, precision=args.precision)
    model = DetectorLit.load_from_checkpoint(args.pre_train_path, **kw) if args.pre_train_path else DetectorLit(**kw)
    model.prepare()
    out_dir = os.path.join("lightning_logs", args.wandb_project, args.wandb_name, "_".join([dataset, args.modality, Config.Detector.name]))
    tr = Trainer(max_epochs=args.epochs, limit_train_batches=args.limit_train_batches, dirpath=out_dir if rank == 0 else None, monitor="val_map",
                 mode="max", early_stopping=("val_map", "max", 5), device=dev, log=print if rank == 0 else (lambda *a: None))
    tr.fit(model, dm)
    if rank == 0:
        print("test:", {k: (v.tolist() if torch.is_tensor(v) else v) for k, v in tr.test(model, dm).items()})
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
