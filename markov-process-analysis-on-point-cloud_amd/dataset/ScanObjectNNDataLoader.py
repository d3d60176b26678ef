"""ScanObjectNN reader with the reference's interface (dataset/ScanObjectNNDataLoader.py:9-31): constructor
`(root, split='training', bg=True)`, items `(cloud [3, N] float32 channel-first, label int64)`.

On-disk contract (the dataset's published layout): `<root>/main_split/` (objects with background points) or
`<root>/main_split_nobg/`, one HDF5 file per split named `<split>_objectdataset_augmentedrot_scale75.h5` holding
`data` [n, 2048, 3] and `label` [n].  h5py is imported on construction, not at package import: the image this
project is built in has no h5py and no ScanObjectNN file, so this reader is untested here ("parity unpinned")."""
import os

from torch.utils.data import Dataset

_SPLITS = ("training", "test")
_H5_SUFFIX = "_objectdataset_augmentedrot_scale75.h5"


def _h5py():
    try:
        import h5py
    except ImportError as e:
        raise ImportError("ScanObjectNNDataLoader reads HDF5 files and needs h5py, which is not installed") from e
    return h5py


class ScanObjectNNDataLoader(Dataset):
    def __init__(self, root, split='training', bg=True):
        h5py = _h5py()
        if split not in _SPLITS:
            raise ValueError("split must be one of %s, got %r" % (_SPLITS, split))
        self.root, self.split, self.bg = root, split, bool(bg)
        self.path = os.path.join(root, "main_split" if bg else "main_split_nobg", split + _H5_SUFFIX)
        with h5py.File(self.path, mode="r") as f:
            # whole arrays in host memory, converted once: clouds stay [n, N, 3] and are transposed per item
            self.data = f["data"][:].astype("float32")
            self.label = f["label"][:].astype("int64")

    def __len__(self):
        return len(self.label)

    def __getitem__(self, index):
        return self.data[index].T, self.label[index]

    def __repr__(self):
        return "ScanObjectNNDataLoader(%s, %d clouds, %s background)" % (self.split, len(self), "with" if self.bg else "without")
