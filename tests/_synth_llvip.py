"""Synthetic LLVIP-layout tree for pipeline / script tests (visible|infrared/{train,test}/*.jpg + Annotations/*.xml)."""
import os

import numpy as np
from PIL import Image

XML = """<annotation><filename>{name}.jpg</filename>{objs}</annotation>"""
OBJ = "<object><name>{cls}</name><bndbox><xmin>{a}</xmin><ymin>{b}</ymin><xmax>{c}</xmax><ymax>{d}</ymax></bndbox></object>"


def make_tree(tmp_path, n_train=10, n_test=3, hw=(32, 40), seed=0, extra_objects=True):
    root = tmp_path / "data" / "LLVIP"
    rng = np.random.RandomState(seed)
    H, W = hw
    for split, n in (("train", n_train), ("test", n_test)):
        for mod in ("visible", "infrared"):
            os.makedirs(root / mod / split, exist_ok=True)
        os.makedirs(root / "Annotations", exist_ok=True)
        for i in range(n):
            name = "%s%04d" % ("1" if split == "train" else "9", i)
            Image.fromarray(rng.randint(0, 256, (H, W, 3), dtype=np.uint8)).save(root / "visible" / split / (name + ".jpg"), quality=95)
            Image.fromarray(rng.randint(0, 256, (H, W), dtype=np.uint8)).save(root / "infrared" / split / (name + ".jpg"), quality=95)
            objs = OBJ.format(cls="person", a=4 + i, b=5, c=20 + i, d=30)
            if extra_objects:
                objs += OBJ.format(cls="person", a=30, b=10, c=28, d=2)          # reversed corners -> re-ordered
                objs += OBJ.format(cls="person", a=1, b=1, c=3, d=3)             # area 4 <= 5 -> dropped
                objs += OBJ.format(cls="car", a=0, b=0, c=30, d=30)              # not a person -> dropped
            (root / "Annotations" / (name + ".xml")).write_text(XML.format(name=name, objs=objs))
    return str(root)
