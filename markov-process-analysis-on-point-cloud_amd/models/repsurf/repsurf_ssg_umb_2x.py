"""RepSurf baseline classifier (umbrella surfaces + three ball-query set abstractions + a global one)
-- drop-in for the reference's models/repsurf/repsurf_ssg_umb_2x.py:11-61.  State-dict keys equal the
reference's (including its spelling `classfier`)."""
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...modules.repsurface_utils import SurfaceAbstractionCD, UmbrellaSurfaceConstructor


class Model(nn.Module):
    def __init__(self, args):
        super().__init__()
        center_channel = 0 if not args.return_center else (6 if args.return_polar else 3)
        repsurf_channel = 10
        self.init_nsample = args.num_point
        self.return_dist = args.return_dist
        self.surface_constructor = UmbrellaSurfaceConstructor(args.group_size + 1, repsurf_channel,
                                                              return_dist=args.return_dist, aggr_type=args.umb_pool,
                                                              cuda=args.cuda_ops)
        sa = dict(pos_channel=center_channel, return_polar=args.return_polar, cuda=args.cuda_ops)
        self.sa1 = SurfaceAbstractionCD(npoint=512, radius=0.1, nsample=24, feat_channel=repsurf_channel,
                                        mlp=[128, 128, 256], group_all=False, **sa)
        self.sa2 = SurfaceAbstractionCD(npoint=128, radius=0.2, nsample=24, feat_channel=256 + repsurf_channel,
                                        mlp=[256, 256, 512], group_all=False, **sa)
        self.sa3 = SurfaceAbstractionCD(npoint=32, radius=0.4, nsample=24, feat_channel=512 + repsurf_channel,
                                        mlp=[512, 512, 1024], group_all=False, **sa)
        self.sa4 = SurfaceAbstractionCD(npoint=None, radius=None, nsample=None, feat_channel=1024 + repsurf_channel,
                                        mlp=[1024, 1024, 2048], group_all=True, **sa)
        self.classfier = nn.Sequential(
            nn.Linear(2048, 512), nn.BatchNorm1d(512), nn.ReLU(True), nn.Dropout(0.4),
            nn.Linear(512, 256), nn.BatchNorm1d(256), nn.ReLU(True), nn.Dropout(0.4),
            nn.Linear(256, args.num_class))

    def forward(self, points):
        center = points[:, :3, :]
        normal = self.surface_constructor(center)
        center, normal, feature = self.sa1(center, normal, None)
        center, normal, feature = self.sa2(center, normal, feature)
        center, normal, feature = self.sa3(center, normal, feature)
        center, normal, feature = self.sa4(center, normal, feature)
        c = self.classfier
        x = feature.reshape(-1, 2048)
        x = c[3](ops.linear_bn_act(x, c[0].weight, c[0].bias, c[1], 0.0))
        x = c[7](ops.linear_bn_act(x, c[4].weight, c[4].bias, c[5], 0.0))
        return F.log_softmax(ops.linear(x, c[8].weight, c[8].bias), -1)
