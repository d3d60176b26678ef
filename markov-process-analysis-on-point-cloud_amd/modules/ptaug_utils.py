"""Train-time augmentation on the device -- mirror of the reference's modules/ptaug_utils.py:22-62
(per-cloud random anisotropic scale and shift of the xyz channels of a [B, 3|6, N] batch), the
step immediately before the hot path in tool/train_cls_scanobjectnn.py:244-245.
Plain device-side torch: two tiny elementwise ops per batch; `sample` (the FPS downsample that
precedes it) is ops.sample on the gfx950 FPS kernel."""
import torch

from ..ops import sample  # noqa: F401


def get_aug_args(args):
    if args.dataset == 'ScanObjectNN':
        return {'scale_factor': 0.5, 'shift_factor': 0.3}
    raise Exception('No such dataset')


def scale_point_cloud(batch_data, scale_range=0.2):
    scales = (torch.rand(batch_data.shape[0], 3, 1, device=batch_data.device) * 2. - 1.) * scale_range + 1.
    batch_data *= scales
    return batch_data


def shift_point_cloud(batch_data, shift_range=0.2):
    shifts = (torch.rand(batch_data.shape[0], 3, 1, device=batch_data.device) * 2. - 1.) * shift_range
    batch_data += shifts
    return batch_data


def transform_point_cloud(batch, args, aug_args, train=True, label=None):
    """batch: B x 3/6 x N"""
    if args.aug_scale:
        batch[:, 0:3] = scale_point_cloud(batch[:, 0:3], aug_args['scale_factor'])
    if args.aug_shift:
        batch[:, 0:3] = shift_point_cloud(batch[:, 0:3], shift_range=aug_args['shift_factor'])
    if label is not None:
        return batch, label
    return batch
