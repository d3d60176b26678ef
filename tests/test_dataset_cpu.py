"""Dataset readers on the CPU: file parsing, splits, caches and the oracle's
restatement of the ModelNet reader's host-side FPS, against golden vectors produced by the
reference's own readers on the synthetic trees of tests/golden/dataset_trees.py."""
import os
import pickle
from argparse import Namespace

import numpy as np
import pytest

from dataset_trees import write_modelnet_tree, write_shapenet_tree, MODELNET_TRAIN


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "dataset.npz"))


@pytest.fixture()
def modelnet_root(tmp_path):
    return write_modelnet_tree(str(tmp_path / "modelnet"))


@pytest.fixture()
def shapenet_root(tmp_path):
    return write_shapenet_tree(str(tmp_path / "shapenet"))


def test_oracle_dataset_fps_matches_reference_reader(golden, modelnet_root):
    from oracle import ref_cpu as R
    for i, sid in enumerate(MODELNET_TRAIN):
        name = "_".join(sid.split("_")[:-1])
        pts = np.loadtxt(os.path.join(modelnet_root, name, sid + ".txt"), delimiter=",").astype(np.float32)
        got = R.dataset_farthest_point_sample(pts, 64, int(golden["mn/uniform/starts"][i]))
        got[:, 0:3] = R.dataset_pc_normalize(got[:, 0:3])
        assert np.array_equal(got, golden["mn/uniform/%d/points" % i])


@pytest.mark.parametrize("split,n", [("train", 4), ("test", 2)])
@pytest.mark.parametrize("normals", [False, True])
def test_modelnet_plain_items(golden, modelnet_root, split, n, normals):
    from mpa_amd.dataset.ModelNetDataLoader import ModelNetDataLoader
    ds = ModelNetDataLoader(modelnet_root, Namespace(use_uniform_sample=False, use_normals=normals, num_category=40), split=split)
    assert len(ds) == n
    for i in range(n):
        pts, lab = ds[i]
        assert pts.dtype == np.float32 and np.array_equal(pts, golden["mn/%s/n%d/%d/points" % (split, normals, i)])
        assert int(lab) == int(golden["mn/%s/n%d/%d/label" % (split, normals, i)])
    pts, labs = ds.get_batch([1, 0])
    assert np.array_equal(pts[0], golden["mn/%s/n%d/1/points" % (split, normals)]) and labs.dtype == np.int32


def test_modelnet_processed_cache_roundtrip(golden, modelnet_root):
    from mpa_amd.dataset.ModelNetDataLoader import ModelNetDataLoader
    args = Namespace(use_uniform_sample=False, use_normals=True, num_category=40)
    first = ModelNetDataLoader(modelnet_root, args, split="train", process_data=True)
    assert os.path.exists(first.cache_path) and not os.path.exists(first.save_path)       # .npz, no pickle written
    again = ModelNetDataLoader(modelnet_root, args, split="train", process_data=True)      # served from the cache
    for ds in (first, again):
        for i in range(4):
            assert np.array_equal(ds[i][0], golden["mn/processed/%d/points" % i])


def test_modelnet_pickle_cache_only_on_request(modelnet_root):
    """A `.dat` in the reference's format is read only with args.allow_pickle_cache."""
    from mpa_amd.dataset.ModelNetDataLoader import ModelNetDataLoader
    fake = [np.full((5, 6), float(i), dtype=np.float32) for i in range(4)]
    path = os.path.join(modelnet_root, "modelnet40_train_10000pts.dat")
    with open(path, "wb") as f:
        pickle.dump([fake, [np.array([i], dtype=np.int32) for i in range(4)]], f)
    args = Namespace(use_uniform_sample=False, use_normals=True, num_category=40, allow_pickle_cache=True)
    ds = ModelNetDataLoader(modelnet_root, args, split="train", process_data=True)
    assert ds.list_of_points[2].shape == (5, 6) and int(ds.list_of_labels[3][0]) == 3
    args = Namespace(use_uniform_sample=False, use_normals=True, num_category=40)
    ds = ModelNetDataLoader(modelnet_root, args, split="train", process_data=True)
    assert ds.list_of_points[2].shape == (300, 6)                                          # processed from the txt files


@pytest.mark.parametrize("split", ["train", "trainval", "val", "test"])
def test_shapenet_listing(golden, shapenet_root, split):
    from mpa_amd.dataset.ShapeNetDataLoader import PartNormalDataset
    ds = PartNormalDataset(root=shapenet_root, npoints=64, split=split, normal_channel=True)
    got = ["%s|%s" % (c, os.path.relpath(f, shapenet_root)) for c, f in ds.datapath]
    assert got == list(golden["sn/%s/datapath" % split])
    assert ["%s=%d" % kv for kv in ds.classes.items()] == list(golden["sn/%s/classes" % split])
    assert ds.seg_classes["Airplane"] == [0, 1, 2, 3] and len(ds.seg_classes) == 16


def test_shapenet_class_choice_and_bad_split(golden, shapenet_root):
    from mpa_amd.dataset.ShapeNetDataLoader import PartNormalDataset
    ds = PartNormalDataset(root=shapenet_root, npoints=64, split="train", class_choice=["Cap"])
    assert ["%s|%s" % (c, os.path.relpath(f, shapenet_root)) for c, f in ds.datapath] == list(golden["sn/choice/datapath"])
    assert ["%s=%d" % kv for kv in ds.classes.items()] == list(golden["sn/choice/classes"])
    with pytest.raises(ValueError):
        PartNormalDataset(root=shapenet_root, split="nope")


def test_scanobjectnn_needs_h5py(tmp_path):
    from mpa_amd.dataset.ScanObjectNNDataLoader import ScanObjectNNDataLoader
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="h5py"):
            ScanObjectNNDataLoader(str(tmp_path))
    else:
        with pytest.raises(OSError):
            ScanObjectNNDataLoader(str(tmp_path))
