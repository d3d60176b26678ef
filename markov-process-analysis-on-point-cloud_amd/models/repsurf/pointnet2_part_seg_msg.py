"""Part-segmentation wiring -- drop-in for the reference's
models/repsurf/pointnet2_part_seg_msg.py:33-180 (get_model, get_loss).  State-dict keys equal
the reference's."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...modules.pointnet2_utils import KeepHighResolutionModulePartSeg, Linear


class get_model(nn.Module):
    def __init__(self, num_classes, normal_channel=False):
        super().__init__()
        self.normal_channel = normal_channel
        self.umb_pool = 'sum'
        self.group_size = 8
        self.return_dist = True
        self.keepHigh = KeepHighResolutionModulePartSeg(3, 64, 128, 256, 512, cuda=True)
        self.conv8 = Linear(896, 512, bn=False)
        self.conv9 = Linear(512, 256, bn=False)
        self.conv10 = Linear(256, 128, bn=False)
        self.conv11 = nn.Linear(128, num_classes)
        self.drop1 = nn.Dropout(0.5)
        self.drop2 = nn.Dropout(0.5)

    def forward(self, xyz, cls_label):
        _, final_points = self.keepHigh(xyz, normal=xyz, label=cls_label)
        x = self.drop1(self.conv8(final_points))
        x = self.conv10(self.conv9(x))
        # logits leave the feature stream in fp32 (on bf16 features the product's fp32 results are stored unrounded)
        return ops.linear(x, self.conv11.weight, self.conv11.bias, out_dtype=torch.float32), xyz


class get_loss(nn.Module):
    """Label-smoothed cross entropy on logits (reference :159-180)."""

    def forward(self, pred, target, trans_feat=None):
        target = target.contiguous().view(-1)
        eps = 0.1
        n_class = pred.size(1)
        one_hot = pred.new_zeros(pred.shape).scatter(1, target.view(-1, 1), 1)
        one_hot = one_hot * (1 - eps) + (1 - one_hot) * eps / (n_class - 1)
        per_point = -(one_hot * F.log_softmax(pred, dim=1)).sum(dim=1)
        n = per_point.numel()
        if n > 4096 and n % 64 == 0:
            # mean in two stages (<= 64 and n/64 elements): keeps torch off its multi-workgroup
            # reduction, whose result is not reliable under HIP-graph replay (see _max_over_points)
            return per_point.view(-1, 64).mean(dim=1).mean()
        return per_point.mean()
