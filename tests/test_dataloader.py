"""Input pipeline on a synthetic LLVIP-layout tree: file discovery, VOC-xml rules, uint8 staging, collate,
split determinism, device staging values (== the reference's float/255)."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from hallucidet_amd.dataloader import (DevicePrefetcher, MultiModalDataModule, MultiModalDetectionDataset, SingleModalDetectionDataset,
                                       get_bbox, split_dataset)

from _synth_llvip import make_tree


@pytest.fixture()
def llvip(tmp_path):
    return make_tree(tmp_path)


def test_get_bbox_rules(llvip):
    a = get_bbox(os.path.join(llvip, "visible", "train", "10003") + ".xml", "llvip", True)
    assert a["bboxes"].dtype == np.float64 and a["labels"].shape == (2, 1)
    assert a["bboxes"].tolist() == [[7.0, 5.0, 23.0, 30.0], [28.0, 2.0, 30.0, 10.0]]
    with pytest.raises(Exception, match="Dataset not supported"):
        get_bbox("x.xml", "coco")


def test_multimodal_dataset_items_and_module(llvip):
    ds = MultiModalDetectionDataset("llvip", llvip, llvip, modality="both", ext=".jpg", train=True)
    assert len(ds) == 10 and ds.get_name(0)[0].endswith("visible/train/10000") and ds.get_name(0)[1].endswith("infrared/train/10000")
    rgb, t_rgb, ir, t_ir = ds[2]
    assert rgb.dtype == torch.uint8 and rgb.shape == (3, 32, 40) and ir.dtype == torch.uint8 and ir.shape == (1, 32, 40)
    assert t_rgb["boxes"].dtype == torch.float64 and t_rgb["boxes"].shape == (2, 4) and t_rgb["labels"].tolist() == [1, 1]
    assert torch.equal(t_rgb["boxes"], t_ir["boxes"])                       # LLVIP pairs are aligned: same annotation file
    want = np.asarray(Image.open(os.path.join(llvip, "visible", "train", "10002.jpg")).convert("RGB")).transpose(2, 0, 1)
    assert np.array_equal(rgb.numpy(), want)
    dm = MultiModalDataModule("llvip", llvip, llvip, llvip, llvip, batch_size=2, num_workers=0, ext=".jpg", seed=123)
    tr, va, te = dm.train_dataloader(), dm.val_dataloader(), dm.test_dataloader()
    assert len(tr) == 4 and len(va) == 1 and len(te) == 1                   # 8/2 split, drop_last
    b = next(iter(te))
    assert len(b) == 4 and len(b[0]) == 2 and isinstance(b[1], tuple) and b[0][0].shape == (3, 32, 40)
    a1, a2 = split_dataset(ds, 0.8, seed=123), split_dataset(ds, 0.8, seed=123)
    assert a1[0].indices == a2[0].indices and len(a1[0]) == 8 and len(a1[1]) == 2


def test_single_modal_and_prefetcher_values(llvip):
    ds = SingleModalDetectionDataset("llvip", llvip, modality="ir", ext=".jpg", train=False)
    img, t = ds[0]
    assert len(ds) == 3 and img.shape == (1, 32, 40) and img.dtype == torch.uint8 and t["path_image"].endswith("infrared/test/90000.jpg")
    with pytest.raises(Exception, match="Dataset not supported"):
        SingleModalDetectionDataset("coco", llvip, modality="rgb")
    dm = MultiModalDataModule("llvip", llvip, llvip, llvip, llvip, batch_size=2, num_workers=0, ext=".jpg")
    batches = list(DevicePrefetcher(dm.test_dataloader(), device="cpu"))
    assert len(batches) == 1
    rgb, t_rgb, ir, t_ir = batches[0]
    assert rgb.shape == (2, 3, 32, 40) and rgb.dtype == torch.float32 and ir.shape == (2, 1, 32, 40)
    raw = MultiModalDetectionDataset("llvip", llvip, llvip, modality="both", ext=".jpg", train=False)[0][0]
    # the reference's host value: uint8.astype(float) / 255.0 (float64), later cast to fp32
    want = torch.from_numpy((raw.numpy().astype("float") / 255.0)).float()
    assert torch.equal(rgb[0], want) and float(rgb.max()) <= 1.0
    assert isinstance(t_rgb, list) and t_rgb[0]["labels"].dtype == torch.int64


def test_augmentation_hook_signature_and_empty_fallback(llvip):
    from hallucidet_amd.dataloader.dataloader import DatasetTransform
    ds = MultiModalDetectionDataset("llvip", llvip, llvip, modality="both", ext=".jpg", train=True)
    calls = []

    def flip(image, bboxes, labels, image1, bboxes1, labels1):
        calls.append(image.shape)
        W = image.shape[1]
        fb = lambda b: [[W - x2, y1, W - x1, y2] for x1, y1, x2, y2 in np.asarray(b).tolist()]
        return {"image": image[:, ::-1].copy(), "bboxes": fb(bboxes), "labels": list(np.asarray(labels)), "image1": image1[:, ::-1].copy(),
                "bboxes1": fb(bboxes1), "labels1": list(np.asarray(labels1))}
    rgb, t, ir, _ = DatasetTransform(ds, flip, "multimodal")[1]
    assert calls == [(32, 40, 3)] and rgb.shape == (3, 32, 40) and ir.shape == (1, 32, 40)
    assert torch.equal(rgb, ds[1][0].flip(-1)) and t["boxes"][0].tolist() == [40 - 21.0, 5.0, 40 - 5.0, 30.0]
    drop_all = lambda **k: {"image": k["image"], "bboxes": [], "labels": [], "image1": k["image1"], "bboxes1": [], "labels1": []}
    _, t2, _, _ = DatasetTransform(ds, drop_all, "multimodal")[1]
    assert torch.equal(t2["boxes"], ds[1][1]["boxes"])                      # fell back to the original targets


def test_data_parallel_shards_share_the_split_and_partition_the_epoch(llvip):
    """every rank must build the SAME train/val split and read a disjoint shard of each epoch."""
    dms = [MultiModalDataModule("llvip", llvip, llvip, llvip, llvip, batch_size=2, num_workers=0, ext=".jpg", seed=123, rank=r, world_size=2)
           for r in (0, 1)]
    tr = [dm.train_dataloader() for dm in dms]
    assert tr[0].dataset.subset.indices == tr[1].dataset.subset.indices                       # same split on both ranks
    assert dms[0].val_dataloader().dataset.subset.indices == dms[1].val_dataloader().dataset.subset.indices
    assert len(tr[0]) == len(tr[1]) == 2                                                      # 8 training samples / (2 ranks * batch 2)
    for epoch in range(2):
        seen = [[i for b in tr[r].batch_sampler for i in b] for r in (0, 1)]
        assert len(seen[0]) == len(seen[1]) == 4 and not set(seen[0]) & set(seen[1])
        assert sorted(seen[0] + seen[1]) == list(range(8))                                    # one pass over the set per epoch
        if epoch == 0:
            first = seen
    assert first != seen                                                                      # reshuffled per epoch


def test_kaist_lists_boxes_and_nonempty_training_positions(tmp_path):
    """dataloader.py:102-113 / utils.py:366,387-389: frame lists from <root>/{train,test}-all-20-{rgb,ir}.txt, x/y/w/h boxes,
    training restricted to the frames that keep at least one person box."""
    root = str(tmp_path / "kaist")
    os.makedirs(os.path.join(root, "set00"))
    names = []
    for i, objs in enumerate([[("person", 5, 6, 10, 12)], [], [("cyclist", 1, 1, 9, 9)], [("person", 2, 3, 4, 5), ("person", 0, 0, 1, 2)]]):
        for mod in ("rgb", "ir"):
            stem = os.path.join("set00", "%s_%d" % (mod, i))
            Image.fromarray(np.full((32, 40, 3) if mod == "rgb" else (32, 40), 10 * i, np.uint8)).save(os.path.join(root, stem + ".jpg"))
            with open(os.path.join(root, stem + ".xml"), "w") as f:
                f.write("<annotation>" + "".join("<object><name>%s</name><bndbox><x>%d</x><y>%d</y><w>%d</w><h>%d</h></bndbox></object>" % o
                                                 for o in objs) + "</annotation>")
        names.append(i)
    for mod in ("rgb", "ir"):
        for split in ("train", "test"):
            with open(os.path.join(root, "%s-all-20-%s.txt" % (split, mod)), "w") as f:
                f.write("\n".join(os.path.join("set00", "%s_%d" % (mod, i)) for i in names) + "\n")
    a = get_bbox(os.path.join(root, "set00", "rgb_3.xml"), "kaist", True)
    assert a["bboxes"].tolist() == [[2.0, 3.0, 6.0, 8.0]]                     # w,h added; the 1x2 box (area 2 <= 5) dropped
    ds = MultiModalDetectionDataset("kaist", root, root, modality="both", ext=".jpg", train=True)
    assert ds.indices == [0, 3] and len(ds) == 2
    rgb, t_rgb, ir, t_ir = ds[1]
    assert rgb.shape == (3, 32, 40) and ir.shape == (1, 32, 40) and int(rgb[0, 0, 0]) == 30 and t_ir["boxes"].tolist() == [[2.0, 3.0, 6.0, 8.0]]
    assert len(SingleModalDetectionDataset("kaist", root, modality="ir", ext=".jpg", train=False)) == 4
    with open(os.path.join(root, "train-indices.txt"), "w") as f:
        f.write("3")
    assert SingleModalDetectionDataset("kaist", root, modality="rgb", ext=".jpg", train=True).indices == [3]
