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
        # -(smoothed one-hot * log_softmax(pred, 1)).sum(1).mean() with eps = 0.1, as one op (logits in, fixed-order mean)
        return ops.smooth_loss(pred, target.contiguous().view(-1), 0.1, from_logits=True)
