"""Writes small synthetic dataset trees in the on-disk formats the reference's readers parse
(dataset/ModelNetDataLoader.py:44-71: modelnet40_normal_resampled-style txt files;
dataset/ShapeNetDataLoader.py:28-84: shapenetcore_partanno_segmentation_benchmark_v0_normal-style
txt + json splits).  Used by make_golden_dataset.py (which runs the reference's readers on them) and
by the tests (which run this project's readers on the very same files)."""
import json
import os

import numpy as np

MODELNET_NAMES = ["airplane", "night_stand", "cup", "chair"]
MODELNET_TRAIN = ["airplane_0001", "night_stand_0002", "cup_0003", "airplane_0004"]
MODELNET_TEST = ["chair_0005", "night_stand_0006"]
MODELNET_POINTS = 300

SHAPENET_CATS = [("Airplane", "02691156"), ("Bag", "02773838"), ("Cap", "02954340")]
# (synset, token, split, number of points, first part label)
SHAPENET_SHAPES = [("02691156", "a1", "train", 150, 0), ("02691156", "a2", "val", 211, 0), ("02691156", "a3", "test", 97, 0),
                   ("02773838", "b1", "train", 180, 4), ("02773838", "b2", "test", 64, 4),
                   ("02954340", "c1", "train", 130, 6), ("02954340", "c2", "train", 201, 6), ("02954340", "c3", "val", 75, 6)]


def _cloud(rng, n):
    xyz = rng.uniform(-1, 1, (n, 3)) * np.array([1.0, 0.6, 0.3]) + rng.uniform(-0.2, 0.2, 3)
    nrm = rng.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return np.concatenate([xyz, nrm], 1)


def write_modelnet_tree(root, seed=7):
    rng = np.random.RandomState(seed)
    os.makedirs(root, exist_ok=True)
    with open(os.path.join(root, "modelnet40_shape_names.txt"), "w") as f:
        f.write("\n".join(MODELNET_NAMES) + "\n")
    for split, ids in (("train", MODELNET_TRAIN), ("test", MODELNET_TEST)):
        with open(os.path.join(root, "modelnet40_%s.txt" % split), "w") as f:
            f.write("\n".join(ids) + "\n")
        for sid in ids:
            name = "_".join(sid.split("_")[:-1])
            os.makedirs(os.path.join(root, name), exist_ok=True)
            np.savetxt(os.path.join(root, name, sid + ".txt"), _cloud(rng, MODELNET_POINTS), fmt="%.6f", delimiter=",")
    return root


def write_shapenet_tree(root, seed=8):
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, "train_test_split"), exist_ok=True)
    with open(os.path.join(root, "synsetoffset2category.txt"), "w") as f:
        for cat, syn in SHAPENET_CATS:
            f.write("%s\t%s\n" % (cat, syn))
    lists = {"train": [], "val": [], "test": []}
    for syn, token, split, n, first in SHAPENET_SHAPES:
        os.makedirs(os.path.join(root, syn), exist_ok=True)
        seg = first + rng.randint(0, 2, (n, 1))
        np.savetxt(os.path.join(root, syn, token + ".txt"), np.concatenate([_cloud(rng, n), seg], 1), fmt="%.6f")
        lists[split].append("shape_data/%s/%s" % (syn, token))
    for split, lst in lists.items():
        with open(os.path.join(root, "train_test_split", "shuffled_%s_file_list.json" % split), "w") as f:
            json.dump(lst, f)
    return root
