"""Dataset readers on the device: the sampling the reference does on the host
(ModelNet `use_uniform_sample`) or one cloud at a time on the GPU (ShapeNetPart `__getitem__`) runs
batched on the gfx950 FPS kernel and must select the very same points."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from dataset_trees import write_modelnet_tree, write_shapenet_tree, MODELNET_TRAIN, SHAPENET_SHAPES
from param_fill import unit_cloud

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "dataset.npz"))


@pytest.fixture(scope="module")
def co():
    from oracle import c_oracle
    return c_oracle


def test_fps_of_10000_point_shapes(co):
    """ModelNet's resampled shapes have 10000 points: the 12288-point instantiation of the kernel."""
    from mpa_amd import ops
    xyz = unit_cloud(2, 10000, seed=3)
    start = torch.tensor([9999, 17])
    idx = ops.farthest_point_sample(xyz.cuda(), 1024, start_idx=start)
    assert np.array_equal(idx.cpu().numpy(), co.farthest_point_sample(xyz.numpy(), 1024, start.numpy()))


def test_ragged_fps_equals_per_cloud_calls(co):
    from mpa_amd import ops
    sizes = [97, 300, 64, 211, 1]
    clouds = [unit_cloud(1, n, seed=n)[0] for n in sizes]
    torch.manual_seed(21)
    idx, padded = ops.farthest_point_sample_ragged([c.cuda() for c in clouds], 128)
    torch.manual_seed(21)
    for i, c in enumerate(clouds):
        start = torch.randint(0, c.shape[0], (1,), dtype=torch.long)
        want = co.farthest_point_sample(c[None].numpy(), 128, start.numpy())[0]
        assert np.array_equal(idx[i].cpu().numpy(), want), i          # incl. index 0 repeated once exhausted
        assert int(idx[i].max()) < c.shape[0]
    assert padded.shape == (5, 300, 3)


def test_modelnet_uniform_sampling_matches_reference_reader(golden, tmp_path):
    from mpa_amd.dataset.ModelNetDataLoader import ModelNetDataLoader
    root = write_modelnet_tree(str(tmp_path / "modelnet"))
    ds = ModelNetDataLoader(root, Namespace(use_uniform_sample=True, use_normals=True, num_category=40), split="train")
    ds.npoints = 64                      # as the golden generator did with the reference reader
    np.random.seed(5)
    for i in range(len(ds)):
        assert np.array_equal(ds[i][0], golden["mn/uniform/%d/points" % i]), i
    np.random.seed(5)
    pts, labels = ds.get_batch(list(range(len(ds))))                  # one launch for all four shapes
    for i in range(len(ds)):
        assert np.array_equal(pts[i], golden["mn/uniform/%d/points" % i]), i
    assert labels.tolist() == [0, 1, 2, 0]


def test_modelnet_processed_uniform_cache(tmp_path):
    """process_data + use_uniform_sample at the reader's hard-coded 10000 samples per shape, batched on the
    device, against the oracle's restatement of the reader's numpy loop (first shapes)."""
    from mpa_amd.dataset.ModelNetDataLoader import ModelNetDataLoader
    from oracle import ref_cpu as R
    root = write_modelnet_tree(str(tmp_path / "modelnet"))
    np.random.seed(9)
    starts = [np.random.randint(0, 300) for _ in MODELNET_TRAIN]
    np.random.seed(9)
    ds = ModelNetDataLoader(root, Namespace(use_uniform_sample=True, use_normals=True, num_category=40), split="train",
                            process_data=True, fps_batch=3)
    assert os.path.basename(ds.cache_path) == "modelnet40_train_10000pts_fps.npz" and os.path.exists(ds.cache_path)
    for i in (0, 3):
        sid = MODELNET_TRAIN[i]
        raw = np.loadtxt(os.path.join(root, "_".join(sid.split("_")[:-1]), sid + ".txt"), delimiter=",").astype(np.float32)
        want = R.dataset_farthest_point_sample(raw, 10000, starts[i])
        assert np.array_equal(ds.list_of_points[i], want), i


def test_shapenet_items_and_batches(co, tmp_path):
    from mpa_amd.dataset.ShapeNetDataLoader import PartNormalDataset
    from oracle import ref_cpu as R
    root = write_shapenet_tree(str(tmp_path / "shapenet"))
    ds = PartNormalDataset(root=root, npoints=128, split="trainval", normal_channel=True)
    assert len(ds) == 6
    torch.manual_seed(33)
    items = [ds[i] for i in range(len(ds))]
    torch.manual_seed(33)
    for i, (pts, cls, seg) in enumerate(items):
        cat, fn = ds.datapath[i]
        data = np.loadtxt(fn).astype(np.float32)
        xyzn = data[:, :6].copy()
        xyzn[:, :3] = R.dataset_pc_normalize(xyzn[:, :3])
        start = torch.randint(0, len(data), (1,), dtype=torch.long)
        # the reference samples on the whole xyz|normal row (dataset/ShapeNetDataLoader.py:127-133 hands the
        # [1,N,6] tensor to farthest_point_sample, which sums over all channels)
        idx = co.farthest_point_sample(xyzn[None].copy(), 128, start.numpy())[0]
        assert pts.dtype == np.float32 and np.array_equal(pts, xyzn[idx]), i
        assert seg.dtype == np.float32 and np.array_equal(seg, data[idx, -1]), i
        assert cls.dtype == np.int32 and cls.shape == (1,) and int(cls[0]) == ds.classes[cat]
    # one launch for the six clouds (97..211 points each, 128 samples: some clouds get exhausted)
    fresh = PartNormalDataset(root=root, npoints=128, split="trainval", normal_channel=True)
    torch.manual_seed(33)
    points, cls, seg = fresh.get_batch(list(range(6)))
    assert points.is_cuda and points.shape == (6, 128, 6) and seg.shape == (6, 128) and cls.shape == (6, 1)
    for i, (p, c, s) in enumerate(items):
        assert np.array_equal(points[i].cpu().numpy(), p) and np.array_equal(seg[i].cpu().numpy(), s), i
    assert {t[1] for t in SHAPENET_SHAPES if t[2] in ("train", "val")} == {os.path.basename(f)[:-4] for _, f in ds.datapath}
