"""Classification wiring -- drop-in for the reference's models/repsurf/repsurf_ssg_umb.py:35-70
(class Model): KeepHighResolutionModule + a three-layer FC head, log-softmax output.
State-dict keys equal the reference's."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...modules.repsurface_utils import KeepHighResolutionModule, index_points  # noqa: F401


class Model(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.init_nsample = args.num_point
        self.return_dist = args.return_dist
        self.keepHigh = KeepHighResolutionModule(3, 64, 64, 64, 64, cuda=args.cuda_ops)
        self.fc1 = nn.Linear(1024, 512)
        self.bn1 = nn.BatchNorm1d(512)
        self.drop1 = nn.Dropout(0.5)
        self.fc2 = nn.Linear(512, 256)
        self.bn2 = nn.BatchNorm1d(256)
        self.drop2 = nn.Dropout(0.5)
        self.fc3 = nn.Linear(256, args.num_class)
        self.lrelu = nn.LeakyReLU(negative_slope=0.2)

    def forward(self, points):
        center = points[:, :3, :]
        normal = center                       # as the reference (:59): the normal input is dead
        x = self.keepHigh(center, normal)
        x = self.drop1(ops.linear_bn_act(x, self.fc1.weight, self.fc1.bias, self.bn1, 0.2))
        x = self.drop2(ops.linear_bn_act(x, self.fc2.weight, self.fc2.bias, self.bn2, 0.2))
        return ops.log_softmax(ops.linear(x, self.fc3.weight, self.fc3.bias, out_dtype=torch.float32))


class SmoothClsLoss(nn.Module):
    """Label-smoothed NLL on log-probabilities (reference util/utils.py:74-88)."""

    def __init__(self, smoothing_ratio=0.1):
        super().__init__()
        self.smoothing_ratio = smoothing_ratio

    def forward(self, pred, target):
        # -(one_hot * (1 - eps) + (1 - one_hot) * eps / (n_class - 1) * pred).sum(1).mean() as one op
        return ops.smooth_loss(pred, target, self.smoothing_ratio, from_logits=False)
