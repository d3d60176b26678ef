"""ScanObjectNN reader -- mirror of the reference's dataset/ScanObjectNNDataLoader.py:9-31
(`<root>/main_split[_nobg]/{training,test}_objectdataset_augmentedrot_scale75.h5`, arrays `data`
[n, 2048, 3] and `label` [n]).  Needs h5py, which is imported when the class is constructed: the
build image of this project does not have it, so there this reader raises ImportError with that
message instead of failing at package import ("parity unpinned": no h5 file or h5py to test with)."""
import warnings

from torch.utils.data import Dataset

warnings.filterwarnings('ignore')


class ScanObjectNNDataLoader(Dataset):
    def __init__(self, root, split='training', bg=True):
        try:
            import h5py
        except ImportError as e:
            raise ImportError("ScanObjectNNDataLoader reads HDF5 files and needs h5py, which is not installed") from e
        self.root = root
        assert (split == 'training' or split == 'test')
        if bg:
            print('Use data with background points')
            dir_name = 'main_split'
        else:
            print('Use data without background points')
            dir_name = 'main_split_nobg'
        h5_name = '{}/{}/{}'.format(self.root, dir_name, split + '_objectdataset_augmentedrot_scale75.h5')
        with h5py.File(h5_name, mode="r") as f:
            self.data = f['data'][:].astype('float32')
            self.label = f['label'][:].astype('int64')
        print('The size of %s data is %d' % (split, self.data.shape[0]))

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, index):
        """channel-first cloud [3, N], label"""
        return self.data[index].T, self.label[index]
