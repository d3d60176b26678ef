"""ShapeNetPart reader -- mirror of the reference's dataset/ShapeNetDataLoader.py:27-150
(`shapenetcore_partanno_segmentation_benchmark_v0_normal` layout: `synsetoffset2category.txt`,
`train_test_split/shuffled_{train,val,test}_file_list.json`, `<synset>/<token>.txt` rows
`x y z nx ny nz part`).

The reference samples every item to `npoints` with FPS on the GPU inside `__getitem__`, one cloud per
call, and copies the result back to the host (:121-141) -- from 10-12 DataLoader worker processes.
`__getitem__` here does the same for one item (same values, same generator use); `get_batch(indices)`
is the form meant for training on the MI355X: the items' clouds -- of different sizes -- are sampled in
ONE launch (`ops.farthest_point_sample_ragged`) and points / part labels are gathered on the device and
stay there, equal to stacking the `__getitem__` results of the same indices."""
import json
import os
import warnings

import numpy as np
import torch
from torch.utils.data import Dataset

from .ModelNetDataLoader import pc_normalize  # noqa: F401  (same function in both reference files)

warnings.filterwarnings('ignore')

SEG_CLASSES = {'Earphone': [16, 17, 18], 'Motorbike': [30, 31, 32, 33, 34, 35], 'Rocket': [41, 42, 43],
               'Car': [8, 9, 10, 11], 'Laptop': [28, 29], 'Cap': [6, 7], 'Skateboard': [44, 45, 46],
               'Mug': [36, 37], 'Guitar': [19, 20, 21], 'Bag': [4, 5], 'Lamp': [24, 25, 26, 27],
               'Table': [47, 48, 49], 'Airplane': [0, 1, 2, 3], 'Pistol': [38, 39, 40],
               'Chair': [12, 13, 14, 15], 'Knife': [22, 23]}


class PartNormalDataset(Dataset):
    def __init__(self, root='./data/shapenetcore_partanno_segmentation_benchmark_v0_normal', npoints=2500,
                 split='train', class_choice=None, normal_channel=False, device=None):
        self.npoints = npoints
        self.root = root
        self.catfile = os.path.join(self.root, 'synsetoffset2category.txt')
        self.cat = {}
        self.normal_channel = normal_channel
        self.device = torch.device(device or "cuda")

        with open(self.catfile, 'r') as f:
            for line in f:
                ls = line.strip().split()
                self.cat[ls[0]] = ls[1]
        self.classes_original = dict(zip(self.cat, range(len(self.cat))))
        if class_choice is not None:
            self.cat = {k: v for k, v in self.cat.items() if k in class_choice}

        ids = {}
        for s in ('train', 'val', 'test'):
            with open(os.path.join(self.root, 'train_test_split', 'shuffled_%s_file_list.json' % s), 'r') as f:
                ids[s] = set(str(d.split('/')[2]) for d in json.load(f))
        wanted = {'trainval': ids['train'] | ids['val'], 'train': ids['train'], 'val': ids['val'], 'test': ids['test']}
        if split not in wanted:
            raise ValueError('Unknown split: %s' % split)        # the reference prints and exit(-1)s (:68-70)
        self.meta = {}
        for item in self.cat:
            dir_point = os.path.join(self.root, self.cat[item])
            fns = [fn for fn in sorted(os.listdir(dir_point)) if fn[0:-4] in wanted[split]]
            self.meta[item] = [os.path.join(dir_point, os.path.splitext(os.path.basename(fn))[0] + '.txt') for fn in fns]
        self.datapath = [(item, fn) for item in self.cat for fn in self.meta[item]]
        self.classes = {i: self.classes_original[i] for i in self.cat.keys()}
        self.seg_classes = {k: list(v) for k, v in SEG_CLASSES.items()}
        self.cache = {}  # from index to (point_set, cls, seg) tuple
        self.cache_size = 20000

    def _load(self, index):
        if index in self.cache:
            point_set, cls, seg = self.cache[index]
        else:
            cat, fn = self.datapath[index]
            cls = np.array([self.classes[cat]]).astype(np.int32)
            data = np.loadtxt(fn).astype(np.float32)
            point_set = data[:, 0:6] if self.normal_channel else data[:, 0:3]
            seg = data[:, -1].astype(np.int32)
            if len(self.cache) < self.cache_size:
                self.cache[index] = (point_set, cls, seg)
        # as the reference: normalised in place, so a cached cloud is normalised again on every access (:119)
        point_set[:, 0:3] = pc_normalize(point_set[:, 0:3])
        return point_set, cls, seg

    def get_batch(self, indices, start_idx=None):
        """-> (points [B,npoints,3|6] f32, cls [B,1] int32, seg [B,npoints] f32) on the device."""
        from .. import ops
        loaded = [self._load(i) for i in indices]
        clouds = [torch.from_numpy(p).to(self.device) for p, _, _ in loaded]
        idx, padded = ops.farthest_point_sample_ragged(clouds, self.npoints, start_idx=start_idx)
        points = ops.index_points(padded, idx)
        nmax = padded.shape[1]
        segs = torch.zeros(len(loaded), nmax, 1, dtype=torch.float32, device=self.device)
        for i, (_, _, s) in enumerate(loaded):
            segs[i, :len(s), 0] = torch.from_numpy(s.astype(np.float32)).to(self.device)
            segs[i, len(s):, 0] = float(s[0])
        seg = ops.index_points(segs, idx).squeeze(-1)
        cls = torch.from_numpy(np.stack([c for _, c, _ in loaded])).to(self.device)
        return points, cls, seg

    def __getitem__(self, index):
        points, cls, seg = self.get_batch([index])
        return points[0].cpu().numpy(), cls[0].cpu().numpy(), seg[0].cpu().numpy()

    def __len__(self):
        return len(self.datapath)
