"""ModelNet40 / ModelNet10 reader -- mirror of the reference's dataset/ModelNetDataLoader.py
(`modelnet40_normal_resampled` layout: `<root>/modelnet40_shape_names.txt`, `modelnet40_{train,test}.txt`,
`<root>/<shape>/<shape>_NNNN.txt` with 10000 comma-separated `x,y,z,nx,ny,nz` rows; :44-71).

What differs from the reference, deliberately:
  * `use_uniform_sample`: the reference runs a python/numpy FPS loop of `npoints` (10000) iterations per
    shape on the host (:20-41).  Here the shapes are sampled on the device, many shapes per launch
    (`ops.farthest_point_sample_ragged`); the selections are the reference's, index for index, because
    the reference's float64 bookkeeping holds fp32-exact values only (distances are computed in fp32,
    :35, and merely stored in a float64 array) and the start indices are drawn from numpy's global
    generator in the same order (`np.random.randint(0, N)` per shape, :31).
  * the pre-processed cache is an `.npz` next to where the reference writes its pickle (`*.dat`,
    :73-104).  A `.dat` is only read when `args.allow_pickle_cache` is set (unpickling runs code from
    the file); without it the shapes are processed again from the txt files.
"""
import os
import pickle
import warnings

import numpy as np
import torch
from torch.utils.data import Dataset

warnings.filterwarnings('ignore')


def pc_normalize(pc):
    """centre on the centroid, scale the farthest point to radius 1 (:12-17)"""
    centred = pc - pc.mean(axis=0)
    return centred / np.sqrt((centred ** 2).sum(axis=1)).max()


def farthest_point_sample_batch(point_sets, npoint, starts=None, device=None):
    """FPS of several [N_i, D] numpy shapes in one device launch -> list of [npoint, D] arrays (the
    selected rows, in selection order).  `starts`: first index per shape; default: drawn like the
    reference does, `np.random.randint(0, N_i)` per shape in list order."""
    from .. import ops
    device = device or torch.device("cuda")
    if starts is None:
        starts = [np.random.randint(0, p.shape[0]) for p in point_sets]
    clouds = [torch.from_numpy(np.ascontiguousarray(p[:, :3], dtype=np.float32)).to(device) for p in point_sets]
    idx, _ = ops.farthest_point_sample_ragged(clouds, npoint, start_idx=torch.as_tensor(np.asarray(starts), dtype=torch.long))
    idx = idx.cpu().numpy()
    return [p[idx[i]] for i, p in enumerate(point_sets)]


def farthest_point_sample(point, npoint, start=None):
    """Input: point [N, D] numpy; returns the sampled points [npoint, D] (:20-41), sampled on the device."""
    return farthest_point_sample_batch([point], npoint, None if start is None else [start])[0]


class ModelNetDataLoader(Dataset):
    def __init__(self, root, args, split='train', process_data=False, fps_batch=32):
        self.root = root
        self.npoints = 10000
        self.process_data = process_data
        self.uniform = args.use_uniform_sample
        self.use_normals = args.use_normals
        self.num_category = args.num_category
        self.allow_pickle_cache = bool(getattr(args, "allow_pickle_cache", False))

        self.catfile = os.path.join(self.root, 'modelnet%d_shape_names.txt' % (10 if self.num_category == 10 else 40))
        self.cat = [line.rstrip() for line in open(self.catfile)]
        self.classes = dict(zip(self.cat, range(len(self.cat))))

        prefix = 'modelnet10' if self.num_category == 10 else 'modelnet40'
        shape_ids = {s: [line.rstrip() for line in open(os.path.join(self.root, '%s_%s.txt' % (prefix, s)))]
                     for s in ('train', 'test')}
        assert (split == 'train' or split == 'test')
        shape_names = ['_'.join(x.split('_')[0:-1]) for x in shape_ids[split]]
        self.datapath = [(shape_names[i], os.path.join(self.root, shape_names[i], shape_ids[split][i]) + '.txt')
                         for i in range(len(shape_ids[split]))]
        print('The size of %s data is %d' % (split, len(self.datapath)))

        tag = '_fps' if self.uniform else ''
        self.save_path = os.path.join(root, 'modelnet%d_%s_%dpts%s.dat' % (self.num_category, split, self.npoints, tag))
        self.cache_path = self.save_path[:-4] + '.npz'

        if self.process_data:
            if os.path.exists(self.cache_path):
                print('Load processed data from %s...' % self.cache_path)
                with np.load(self.cache_path) as z:
                    offs = z['offsets']
                    self.list_of_points = [z['points'][offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
                    self.list_of_labels = [z['labels'][i:i + 1] for i in range(len(offs) - 1)]
            elif os.path.exists(self.save_path) and self.allow_pickle_cache:
                print('Load processed data from %s...' % self.save_path)
                with open(self.save_path, 'rb') as f:
                    self.list_of_points, self.list_of_labels = pickle.load(f)
            else:
                print('Processing data %s (only running in the first time)...' % self.cache_path)
                self.list_of_points = [None] * len(self.datapath)
                self.list_of_labels = [None] * len(self.datapath)
                for lo in range(0, len(self.datapath), fps_batch):
                    chunk = range(lo, min(lo + fps_batch, len(self.datapath)))
                    sets = [self._load(i) for i in chunk]
                    if self.uniform:
                        sets = farthest_point_sample_batch(sets, self.npoints)
                    else:
                        sets = [p[0:self.npoints, :] for p in sets]
                    for i, p in zip(chunk, sets):
                        self.list_of_points[i] = p
                        self.list_of_labels[i] = np.array([self.classes[self.datapath[i][0]]]).astype(np.int32)
                offs = np.cumsum([0] + [p.shape[0] for p in self.list_of_points])
                np.savez(self.cache_path, points=np.concatenate(self.list_of_points, 0), offsets=offs,
                         labels=np.concatenate(self.list_of_labels))

    def _load(self, index):
        return np.loadtxt(self.datapath[index][1], delimiter=',').astype(np.float32)

    def __len__(self):
        return len(self.datapath)

    def _finish(self, point_set):
        point_set[:, 0:3] = pc_normalize(point_set[:, 0:3])
        if not self.use_normals:
            point_set = point_set[:, 0:3]
        return point_set

    def _get_item(self, index):
        if self.process_data:
            point_set, label = self.list_of_points[index], self.list_of_labels[index]
        else:
            label = np.array([self.classes[self.datapath[index][0]]]).astype(np.int32)
            point_set = self._load(index)
            if self.uniform:
                point_set = farthest_point_sample(point_set, self.npoints)
            else:
                point_set = point_set[0:self.npoints, :]
        return self._finish(point_set), label[0]

    def __getitem__(self, index):
        return self._get_item(index)

    def get_batch(self, indices):
        """Several items with ONE sampling launch for the `use_uniform_sample` path (items and generator
        use equal `[self[i] for i in indices]`) -> (list of point arrays, int32 label array)."""
        if self.process_data or not self.uniform:
            items = [self[i] for i in indices]
            return [p for p, _ in items], np.array([l for _, l in items], dtype=np.int32)
        sets = farthest_point_sample_batch([self._load(i) for i in indices], self.npoints)
        labels = np.array([self.classes[self.datapath[i][0]] for i in indices], dtype=np.int32)
        return [self._finish(p) for p in sets], labels
