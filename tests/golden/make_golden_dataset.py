"""Golden vectors for the dataset readers, produced by RUNNING THE REFERENCE'S
readers (dataset/ModelNetDataLoader.py, dataset/ShapeNetDataLoader.py) on the synthetic trees of
dataset_trees.py.  Build container only; writes tests/golden/dataset.npz.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_dataset.py

ModelNet: items of the plain path, of the pre-processed (`process_data`) path and of the
`use_uniform_sample` path (the reader's own numpy FPS; `npoints` lowered from the hard-coded
10000 to 64 after construction so the file stays small; numpy's global generator seeded right
before).  ShapeNet: what the constructor derives from the tree (`datapath`, `classes`); its
`__getitem__` needs a CUDA device (`.cuda()`, ShapeNetDataLoader.py:125) and cannot run here, so
the per-item sampling is pinned through the FPS golden vectors instead."""
import os
import sys
import tempfile
from argparse import Namespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_shim import load_reference, REF_ROOT  # noqa: E402
from dataset_trees import write_modelnet_tree, write_shapenet_tree  # noqa: E402

load_reference()           # aliases the module names ShapeNetDataLoader imports
import importlib  # noqa: E402

mn = importlib.import_module("dataset.ModelNetDataLoader")
sn = importlib.import_module("dataset.ShapeNetDataLoader")

d = {}
with tempfile.TemporaryDirectory() as tmp:
    root = write_modelnet_tree(os.path.join(tmp, "modelnet"))
    for split in ("train", "test"):
        for normals in (False, True):
            args = Namespace(use_uniform_sample=False, use_normals=normals, num_category=40)
            ds = mn.ModelNetDataLoader(root, args, split=split)
            for i in range(len(ds)):
                pts, lab = ds[i]
                d["mn/%s/n%d/%d/points" % (split, normals, i)] = pts
                d["mn/%s/n%d/%d/label" % (split, normals, i)] = np.int64(lab)
    args = Namespace(use_uniform_sample=False, use_normals=True, num_category=40)
    ds = mn.ModelNetDataLoader(root, args, split="train", process_data=True)
    for i in range(len(ds)):
        d["mn/processed/%d/points" % i] = ds[i][0]
    args = Namespace(use_uniform_sample=True, use_normals=True, num_category=40)
    ds = mn.ModelNetDataLoader(root, args, split="train")
    ds.npoints = 64
    np.random.seed(5)
    starts = []
    for i in range(len(ds)):
        st = np.random.get_state()
        starts.append(np.random.randint(0, 300))
        np.random.set_state(st)
        d["mn/uniform/%d/points" % i] = ds[i][0]
    d["mn/uniform/starts"] = np.array(starts)

    root = write_shapenet_tree(os.path.join(tmp, "shapenet"))
    for split in ("train", "trainval", "val", "test"):
        ds = sn.PartNormalDataset(root=root, npoints=64, split=split, normal_channel=True)
        d["sn/%s/datapath" % split] = np.array(["%s|%s" % (c, os.path.relpath(f, root)) for c, f in ds.datapath])
        d["sn/%s/classes" % split] = np.array(["%s=%d" % kv for kv in ds.classes.items()])
    ds = sn.PartNormalDataset(root=root, npoints=64, split="train", class_choice=["Cap"])
    d["sn/choice/datapath"] = np.array(["%s|%s" % (c, os.path.relpath(f, root)) for c, f in ds.datapath])
    d["sn/choice/classes"] = np.array(["%s=%d" % kv for kv in ds.classes.items()])

path = os.path.join(HERE, "dataset.npz")
np.savez_compressed(path, **d)
print("dataset.npz %.1f KiB, %d arrays" % (os.path.getsize(path) / 1024, len(d)))
