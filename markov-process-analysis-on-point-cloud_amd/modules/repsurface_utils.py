"""Drop-in for the reference's modules/repsurface_utils.py on MI355X (classification tree):
same names, signatures and state-dict keys; device work in libmpa_hip.so.

Reference locations (Markov_Process_Analysis_on_Point_Cloud/modules/repsurface_utils.py):
  sample_and_group:12  square_distance:129  farthest_point_sample:150  index_points:174
  knn_point:193  SurfaceAbstractionCD:256  Linear:380  LocalMerge:406  LocalTrans:448
  KeepHighResolutionModule:542  resort_points:86  group_by_umbrella:106  SurfaceAbstraction:206
  UmbrellaSurfaceConstructor:321 (shared with pointnet2_utils.py, where they are defined)
(polar_utils / recons_utils: the sibling modules of the same names).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..ops import (farthest_point_sample, index_points, knn_point, query_ball_point, query_knn_point,  # noqa: F401
                   sample, square_distance)
from .pointnet2_utils import (Linear, LocalTrans, UmbrellaSurfaceConstructor, group_by_umbrella,  # noqa: F401
                               centres_and_projections, local_trans_pair, resort_points, stacked_param_groups)


def sample_and_group(npoint, radius, nsample, center, normal, feature, return_normal=True, return_polar=False,
                     cuda=False):
    """reference :12-56 -- FPS, ball query, centre-relative grouping of xyz (| its spherical
    coordinates, `return_polar`) | normal | feature."""
    fps_idx = farthest_point_sample(center, npoint)
    new_center = index_points(center, fps_idx)
    new_normal = index_points(normal, fps_idx)
    idx = query_ball_point(radius, nsample, center, new_center, cuda=cuda)
    group_normal = index_points(normal, idx)
    group_center_norm = index_points(center, idx) - new_center.unsqueeze(2)
    if return_polar:
        from .polar_utils import xyz2sphere
        group_center_norm = torch.cat([group_center_norm, xyz2sphere(group_center_norm)], dim=-1)
    if feature is not None:
        group_feature = index_points(feature, idx)
        parts = [group_center_norm, group_normal, group_feature] if return_normal else [group_center_norm,
                                                                                       group_feature]
    else:
        parts = [group_center_norm, group_normal]
    return new_center, new_normal, torch.cat(parts, dim=-1)


def sample_and_group_all(center, normal, feature, return_normal=True, return_polar=False):
    """reference :58-84 -- the whole cloud as one group around the origin."""
    B, N, C = normal.shape
    new_center = torch.zeros(B, 1, 3, device=center.device)
    group_center = center.view(B, 1, N, 3)
    if return_polar:
        from .polar_utils import xyz2sphere
        group_center = torch.cat([group_center, xyz2sphere(group_center)], dim=-1)
    parts = [group_center, normal.view(B, 1, N, C), feature.view(B, 1, N, -1)] if return_normal else \
        [group_center, feature.view(B, 1, N, -1)]
    return new_center, new_center, torch.cat(parts, dim=-1)


def _conv_bn(x, conv, bn, slope):
    """1x1 Conv2d + BatchNorm2d (+ ReLU when slope == 0.0) over [B,S,G,C] rows on the Linear unit."""
    B, S, G, C = x.shape
    y = ops.linear_bn_act(x.reshape(B, S * G, C), conv.weight.view(conv.out_channels, conv.in_channels), conv.bias,
                          bn, slope)
    return y.view(B, S, G, conv.out_channels)


class SurfaceAbstraction(nn.Module):
    """RepSurf set abstraction without the split first layer (reference :206-254): FPS + ball query (or
    the whole cloud) + shared MLP (1x1 Conv2d + BatchNorm2d + ReLU per layer) + max over the group."""

    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all, return_polar=True, return_normal=True,
                 cuda=False):
        super().__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.return_normal, self.return_polar = return_normal, return_polar
        self.cuda_ops = cuda
        self.group_all = group_all
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv2d(last, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm2d(out_channel))
            last = out_channel

    def forward(self, center, normal, feature):
        normal = normal.permute(0, 2, 1).contiguous()
        center = center.permute(0, 2, 1).contiguous()
        if feature is not None:
            feature = feature.permute(0, 2, 1).contiguous()
        if self.group_all:
            new_center, new_normal, new_feature = sample_and_group_all(center, normal, feature,
                                                                       return_polar=self.return_polar,
                                                                       return_normal=self.return_normal)
        else:
            new_center, new_normal, new_feature = sample_and_group(self.npoint, self.radius, self.nsample, center,
                                                                   normal, feature, return_polar=self.return_polar,
                                                                   return_normal=self.return_normal, cuda=self.cuda_ops)
        new_feature = new_feature.contiguous()
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            new_feature = _conv_bn(new_feature, conv, bn, 0.0)
        new_feature = new_feature.max(dim=2)[0].permute(0, 2, 1)
        return new_center.permute(0, 2, 1), new_normal.permute(0, 2, 1), new_feature


class SurfaceAbstractionCD(nn.Module):
    """RepSurf set abstraction (reference :256-319): FPS + ball query (or the whole cloud, group_all)
    + shared MLP + max over the group.  Grouping on the gfx950 kernels, the 1x1 convolutions with
    their BatchNorm2d + ReLU as the fp32-MFMA Linear unit over the B*S*nsample rows."""

    def __init__(self, npoint, radius, nsample, feat_channel, pos_channel, mlp, group_all, return_normal=True,
                 return_polar=False, cuda=False):
        super().__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.return_normal, self.return_polar = return_normal, return_polar
        self.cuda_ops = cuda
        self.pos_channel = pos_channel
        self.group_all = group_all
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        self.mlp_l0 = nn.Conv2d(pos_channel, mlp[0], 1)
        self.mlp_f0 = nn.Conv2d(feat_channel, mlp[0], 1)
        self.bn_l0 = nn.BatchNorm2d(mlp[0])
        self.bn_f0 = nn.BatchNorm2d(mlp[0])
        last = mlp[0]
        for out_channel in mlp[1:]:
            self.mlp_convs.append(nn.Conv2d(last, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm2d(out_channel))
            last = out_channel

    def forward(self, center, normal, feature):
        normal = normal.permute(0, 2, 1).contiguous()
        center = center.permute(0, 2, 1).contiguous()
        if feature is not None:
            feature = feature.permute(0, 2, 1).contiguous()
        if self.group_all:
            new_center, new_normal, new_feature = sample_and_group_all(center, normal, feature,
                                                                       return_normal=self.return_normal,
                                                                       return_polar=self.return_polar)
        else:
            new_center, new_normal, new_feature = sample_and_group(self.npoint, self.radius, self.nsample, center,
                                                                   normal, feature, return_normal=self.return_normal,
                                                                   return_polar=self.return_polar, cuda=self.cuda_ops)
        # [B,S,G,C] rows (the reference permutes to [B,C,G,S] for Conv2d; same arithmetic per row)
        pc = self.pos_channel
        loc = _conv_bn(new_feature[..., :pc].contiguous(), self.mlp_l0, self.bn_l0, None)
        feat = _conv_bn(new_feature[..., pc:].contiguous(), self.mlp_f0, self.bn_f0, None)
        new_feature = F.relu(loc + feat)
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            new_feature = _conv_bn(new_feature, conv, bn, 0.0)
        new_feature = new_feature.max(dim=2)[0].permute(0, 2, 1)          # max over the group -> [B,C,S]
        return new_center.permute(0, 2, 1), new_normal.permute(0, 2, 1), new_feature


class LocalMerge(nn.Module):
    """State -> state probability-transition block, classification variant (reference :406-446):
    xyz-space and feature-space neighbourhoods, two attention streams, fc2 over 2*out.
    `normal` is returned un-indexed, as in the reference (:446)."""

    def __init__(self, in_channels, out_channels, knn, usetanh=False, residual=False):
        super().__init__()
        self.knn = knn
        self.usetanh = usetanh
        self.residual = residual
        self.fc1 = Linear(out_channels * 2, out_channels, bn=False)
        self.fc2 = Linear(out_channels * 2, out_channels, bn=False)
        self.xyz_Trans = LocalTrans(3, out_channels, knn, usetanh=usetanh, residual=True)
        self.normal_Trans = LocalTrans(10, out_channels, knn, usetanh=usetanh, residual=True)
        self.feature_Trans = LocalTrans(in_channels, out_channels, knn, usetanh=usetanh, residual=residual)
        self.feature_Trans2 = LocalTrans(in_channels, out_channels, knn, usetanh=usetanh, residual=residual)

    def mpa_adjacent_params(self):
        return stacked_param_groups(self.feature_Trans, self.feature_Trans2)

    def forward(self, xyz, base_xyz, normal=None, feature=None, FPS_idx=None, xyz_flag=True, geometry=None):
        # `geometry` (optional, not in the reference signature): this level's precomputed
        # (dist, idx) of knn_point(self.knn, base_xyz, xyz) from ops.geometry_pass
        # `geometry`: a level of ops.GeometryChain -- it issues this state's searches together with the NEXT state's
        # sampling (one launch: the FPS chain hides behind the feature-space search)
        if feature is None:
            dist, idx = knn_point(self.knn, base_xyz, xyz) if geometry is None else geometry.xyz_search()
            merge_features = self.xyz_Trans(features=xyz, idx=idx, pos=base_xyz, FPS_idx=FPS_idx, xyz=True)
        else:
            fs, kvkv = centres_and_projections(self.feature_Trans, self.feature_Trans2, feature, FPS_idx)
            if geometry is None:
                dist, idx = knn_point(self.knn, base_xyz, xyz)
                _, idx_feature = knn_point(self.knn, feature, fs)
            else:
                (dist, idx), idx_feature = geometry.search(self.knn, feature, fs)
            merge_features = self.fc2(local_trans_pair(self.feature_Trans, self.feature_Trans2, feature, idx,
                                                       idx_feature, fs, concat=True, kvkv=kvkv))
        return merge_features, normal, idx, dist


class KeepHighResolutionModule(nn.Module):
    """Classification encoder wiring (reference :542-639): la0 on the full cloud, then five
    FPS halvings (512..32 for N=1024) each followed by a LocalMerge, conv3/conv4, max|avg pool,
    final_class + BN + LeakyReLU -> [B,1024]."""

    def __init__(self, data_C, b1_C, b2_C, b3_C, b4_C, cuda=False):
        super().__init__()
        self.cuda_ops = cuda   # the reference stores this as `self.cuda`, shadowing nn.Module.cuda
        self.drop = nn.Dropout(0.5)
        self.la0 = LocalMerge(32, 64, 8, usetanh=False, residual=True)
        self.la1 = LocalMerge(64, 64, 8, usetanh=False, residual=False)
        self.la2 = LocalMerge(64, 64, 8, usetanh=False, residual=False)
        self.la3 = LocalMerge(64, 128, 8, usetanh=False, residual=True)
        self.la4 = LocalMerge(128, 256, 8, usetanh=False, residual=True)
        self.la5 = LocalMerge(256, 512, 8, usetanh=False, residual=True)
        self.start = Linear(3, 32, bn=False)
        self.conv3 = Linear(512, 512, bn=False)
        self.conv4 = Linear(512, 1024, bn=False)
        self.final = Linear(512, 1024, bn=False)
        self.final_class = nn.Linear(2048, 1024)
        self.bn = nn.BatchNorm1d(1024)
        self.lrelu = nn.LeakyReLU(negative_slope=0.2)

    # the reference hard-codes 512/256/128/64/32 for its 1024-point input
    LEVELS = (512, 256, 128, 64, 32)

    def forward(self, xyz, normal):
        same = normal is xyz                 # (the shipped models pass the coordinates as `normal`: one transposition)
        xyz = xyz.permute(0, 2, 1).contiguous()
        normal = xyz if same else normal.permute(0, 2, 1).contiguous()
        # Every FPS level and every xyz-space kNN depends on the input coordinates only (992 serial FPS
        # iterations on 64 of the 256 CUs): the chain is advanced on demand, state i+1's sampling in the same
        # launch as state i's searches, which keep the rest of the chip busy meanwhile.
        geo = ops.GeometryChain(xyz, self.LEVELS, self.la0.knn)
        feat, normal, _, _ = self.la0(xyz=xyz, base_xyz=xyz, normal=normal, xyz_flag=True, geometry=geo.level(0))
        base = xyz
        for lvl, la in enumerate((self.la1, self.la2, self.la3, self.la4, self.la5), start=1):
            g = geo.level(lvl)
            feat, normal, _, _ = la(xyz=g.xyz, base_xyz=base, normal=normal, feature=feat, FPS_idx=g.fps_idx,
                                    geometry=g)
            base = g.xyz
        final = self.conv4(self.conv3(feat))                       # [B,32,1024]
        fused = ops.pool_max_mean(final)                            # cat((final.max(dim=1)[0], final.mean(dim=1)), 1)
        return ops.linear_bn_act(fused, self.final_class.weight, self.final_class.bias, self.bn, 0.2)
