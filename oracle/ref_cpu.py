"""Plain-PyTorch CPU restatement of the reference's Markov set-abstraction path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this file; the product package never does.  It restates the
reference's *formulation* op for op (python FPS loop, materialised [B,S,N] distance matrix
+ topk, advanced-index gathers, un-fused difference attention, Linear + BatchNorm1d +
LeakyReLU, dense scatter upsample) so that (i) it is the yardstick the HIP path is held
to and (ii) timing it on the GPU box's host cores is a fair stand-in for the reference's
own CPU path ("port" in bench.py's cpu_baseline).

Parity is PINNED: tests/test_oracle_golden.py loads golden vectors produced by importing
the reference itself (tests/golden/make_golden.py) and checks every block below against
them; module/parameter names equal the reference's so state dicts interchange.

Reference locations restated here (paths under Markov_Process_Analysis_on_Point_Cloud/):
  square_distance / knn_point / farthest_point_sample / index_points / query_ball_point
      modules/pointnet2_utils.py:64-134,190-222 (== modules/repsurface_utils.py:129-204)
  upsample                         modules/pointnet2_utils.py:13-50
  Linear                           modules/pointnet2_utils.py:401-425
  LocalTrans                       modules/pointnet2_utils.py:479-574
  LocalMerge (seg / cls)           modules/pointnet2_utils.py:427-477 / modules/repsurface_utils.py:406-446
  Fuse                             modules/pointnet2_utils.py:576-709
  KeepHighResolutionModulePartSeg  modules/pointnet2_utils.py:711-858
  KeepHighResolutionModule         modules/repsurface_utils.py:542-639
  PointNetFeaturePropagation       modules/pointnet2_utils.py:860-912
  cls Model / seg get_model        models/repsurf/repsurf_ssg_umb.py:35-70 / pointnet2_part_seg_msg.py:33-156
  SmoothClsLoss / get_loss         util/utils.py:74-88 / pointnet2_part_seg_msg.py:159-180
  xyz2sphere                       modules/polar_utils.py:10-31
  cal_normal / cal_center / cal_const / check_nan_umb   modules/recons_utils.py:27-57,82-90,108-124,152-176
  group_by_umbrella / UmbrellaSurfaceConstructor        modules/repsurface_utils.py:106-126,321-376
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- L0 ops
def square_distance(src, dst):
    d = -2 * torch.matmul(src, dst.transpose(1, 2))
    d += (src ** 2).sum(-1).unsqueeze(-1)
    d += (dst ** 2).sum(-1).unsqueeze(1)
    return d


def knn_point(nsample, xyz, new_xyz):
    d = square_distance(new_xyz, xyz)
    return torch.topk(d, nsample, dim=-1, largest=False, sorted=True)


def index_points(points, idx):
    B = points.shape[0]
    bshape = [B] + [1] * (idx.dim() - 1)
    bidx = torch.arange(B, dtype=torch.long, device=points.device).view(bshape).expand_as(idx)
    return points[bidx, idx, :]


def farthest_point_sample(xyz, npoint, start_idx=None):
    """start_idx=None draws from the global CPU generator exactly where the reference does."""
    B, N, C = xyz.shape
    out = torch.zeros(B, npoint, dtype=torch.long)
    mind = torch.full((B, N), 1e10)
    far = torch.randint(0, N, (B,), dtype=torch.long) if start_idx is None else start_idx.clone()
    rows = torch.arange(B, dtype=torch.long)
    for i in range(npoint):
        out[:, i] = far
        c = xyz[rows, far, :].view(B, 1, C)
        d = ((xyz - c) ** 2).sum(-1)
        closer = d < mind
        mind[closer] = d[closer]
        far = mind.max(-1)[1]
    return out


def query_ball_point(radius, nsample, xyz, new_xyz):
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    gi = torch.arange(N, dtype=torch.long).view(1, 1, N).repeat(B, S, 1)
    d = square_distance(new_xyz, xyz)
    gi[d > radius ** 2] = N
    gi = gi.sort(dim=-1)[0][:, :, :nsample]
    first = gi[:, :, :1].expand(-1, -1, nsample)
    pad = gi == N
    gi[pad] = first[pad]
    return gi


def upsample(points, knn_idx, scale_ratio=2, dist=None):
    """Dense formulation, as the reference: scatter every coarse row into a
    [B,S,S*r,C] zero tensor, sum over the coarse axis, divide by the number of
    contributors whose channel-0 value is non-zero (0 -> 1)."""
    B, S, C = points.shape
    K = knn_idx.shape[2]
    big = torch.zeros(B, S, S * scale_ratio, C, dtype=points.dtype)
    src = points.unsqueeze(2).expand(B, S, K, C)
    big = big.scatter(2, knn_idx.unsqueeze(-1).expand(B, S, K, C), src)
    total = big.sum(1)
    cnt = torch.count_nonzero(big[..., 0], dim=1).unsqueeze(-1).float()
    cnt = torch.where(cnt == 0, torch.ones_like(cnt), cnt)
    return total / cnt


def three_interpolate(xyz1, xyz2, points2):
    """The interpolation inside PointNetFeaturePropagation (3-NN, inverse-distance)."""
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    if S == 1:
        return points2.repeat(1, N, 1)
    d, idx = square_distance(xyz1, xyz2).sort(dim=-1)
    d, idx = d[:, :, :3], idx[:, :, :3]
    w = 1.0 / (d + 1e-8)
    w = w / w.sum(2, keepdim=True)
    return (index_points(points2, idx) * w.unsqueeze(-1)).sum(2)


# ----------------------------------------------------------------------------- dataset readers (8f-4)
def dataset_pc_normalize(pc):
    """dataset/ModelNetDataLoader.py:12-17 == dataset/ShapeNetDataLoader.py:20-25 (numpy)."""
    import numpy as np
    pc = pc - np.mean(pc, axis=0)
    return pc / np.max(np.sqrt(np.sum(pc ** 2, axis=1)))


def dataset_farthest_point_sample(point, npoint, start):
    """The ModelNet reader's own host-side FPS (dataset/ModelNetDataLoader.py:20-41), numpy, one shape:
    fp32 squared distances kept in a float64 array, strict `<` update, first arg-max; `start` is the
    value the reference draws with np.random.randint(0, N).  Returns the sampled rows [npoint, D]."""
    import numpy as np
    N, D = point.shape
    xyz = point[:, :3]
    chosen = np.zeros((npoint,), dtype=np.int64)
    distance = np.ones((N,)) * 1e10
    farthest = int(start)
    for i in range(npoint):
        chosen[i] = farthest
        dist = np.sum((xyz - xyz[farthest, :]) ** 2, -1)
        closer = dist < distance
        distance[closer] = dist[closer]
        farthest = int(np.argmax(distance, -1))
    return point[chosen]


# ----------------------------------------------------------------------------- L1 blocks
class Linear(nn.Module):
    def __init__(self, in_channels, out_channels, bn=True, act=True):
        super().__init__()
        self.act_flag, self.bn_flag = act, bn
        self.linear = nn.Linear(in_channels, out_channels)
        self.norm1 = nn.LayerNorm(out_channels)
        self.norm2 = nn.BatchNorm1d(out_channels)
        self.act = nn.LeakyReLU(negative_slope=0.2)

    def forward(self, x):
        y = self.linear(x)
        if self.bn_flag:
            y = self.norm1(y)
        else:
            y = self.norm2(y.transpose(1, 2).contiguous()).transpose(1, 2).contiguous()
        return self.act(y) if self.act_flag else y


class LocalTrans(nn.Module):
    def __init__(self, in_c, out_c, patch_num, usetanh=False, residual=False):
        super().__init__()
        self.patchNum, self.residual, self.usetanh, self.out_c = patch_num, residual, usetanh, out_c
        self.q = nn.Linear(in_c, out_c)
        self.k = nn.Linear(in_c, out_c)
        self.v = nn.Linear(in_c, out_c)
        self.conv_res = Linear(in_c, out_c, bn=False)
        self.ffn = Linear(out_c, out_c, bn=False)
        self.tanh = nn.Tanh()

    def forward(self, features, idx, pos, FPS_idx=None, xyz=False):
        center = index_points(features, FPS_idx) if FPS_idx is not None else features
        res = self.conv_res(center) if self.residual else center
        q = self.q(center).unsqueeze(2)
        if xyz:
            rel = index_points(features, idx) - center.unsqueeze(2)
            k, v = self.k(rel), self.v(rel)
        else:
            k = index_points(self.k(features), idx)
            v = index_points(self.v(features), idx)
        e = q - k
        if self.usetanh:
            ctx = torch.matmul(self.tanh(e) / self.patchNum, v).squeeze(-2)
        else:
            a = F.softmax(e / math.sqrt(k.size(-1)), dim=2)
            a = a - a.sum(2, keepdim=True)
            ctx = (a * v).max(2)[0]
        return res + self.ffn(ctx)


class LocalMergeCls(nn.Module):
    """modules/repsurface_utils.py:406-446"""

    def __init__(self, in_channels, out_channels, knn, usetanh=False, residual=False):
        super().__init__()
        self.knn = knn
        self.fc1 = Linear(out_channels * 2, out_channels, bn=False)
        self.fc2 = Linear(out_channels * 2, out_channels, bn=False)
        self.xyz_Trans = LocalTrans(3, out_channels, knn, usetanh, residual=True)
        self.normal_Trans = LocalTrans(10, out_channels, knn, usetanh, residual=True)
        self.feature_Trans = LocalTrans(in_channels, out_channels, knn, usetanh, residual=residual)
        self.feature_Trans2 = LocalTrans(in_channels, out_channels, knn, usetanh, residual=residual)

    def forward(self, xyz, base_xyz, normal=None, feature=None, FPS_idx=None, xyz_flag=True):
        dist, idx = knn_point(self.knn, base_xyz, xyz)
        if feature is None:
            out = self.xyz_Trans(xyz, idx, base_xyz, FPS_idx=FPS_idx, xyz=True)
        else:
            fq = feature if FPS_idx is None else index_points(feature, FPS_idx)
            _, idx_f = knn_point(self.knn, feature, fq)
            a = self.feature_Trans(feature, idx, base_xyz, FPS_idx=FPS_idx)
            b = self.feature_Trans2(feature, idx_f, base_xyz, FPS_idx=FPS_idx)
            out = self.fc2(torch.cat((a, b), dim=2))
        return out, normal, idx, dist


class LocalMergeSeg(nn.Module):
    """modules/pointnet2_utils.py:427-477"""

    def __init__(self, in_channels, out_channels, knn, usetanh=False, residual=False):
        super().__init__()
        self.knn = knn
        self.fc2 = Linear(out_channels * 3, out_channels, bn=False)
        self.xyz_Trans = LocalTrans(3, out_channels, knn, usetanh, residual=True)
        self.normal_Trans = LocalTrans(10, out_channels, knn, usetanh, residual=True)
        self.feature_Trans1 = LocalTrans(in_channels, out_channels, knn, usetanh, residual=residual)
        self.feature_Trans2 = LocalTrans(in_channels, out_channels, knn, usetanh, residual=residual)

    def forward(self, xyz, base_xyz, normal=None, feature=None, FPS_idx=None, xyz_flag=True):
        dist, idx = knn_point(self.knn, base_xyz, xyz)
        if feature is None:
            out = self.xyz_Trans(xyz, idx, base_xyz, FPS_idx=FPS_idx, xyz=True)
        else:
            fq = feature if FPS_idx is None else index_points(feature, FPS_idx)
            _, idx_f = knn_point(self.knn, feature, fq)
            x = self.xyz_Trans(base_xyz, idx, base_xyz, FPS_idx=FPS_idx, xyz=True)
            a = self.feature_Trans1(feature, idx, base_xyz, FPS_idx=FPS_idx)
            b = self.feature_Trans2(feature, idx_f, base_xyz, FPS_idx=FPS_idx)
            out = self.fc2(torch.cat((x, a, b), dim=2))
        if FPS_idx is not None:
            normal = index_points(normal, FPS_idx)
        return out, normal, idx, dist


def _compose(*maps):
    """maps = (FPS_a, FPS_b, ...): index of level-(a+n) points in level-a, i.e.
    FPS_a[FPS_b[...]] as the reference builds with nested index_points."""
    out = maps[-1]
    for m in reversed(maps[:-1]):
        out = torch.gather(m, 1, out)
    return out


class Fuse(nn.Module):
    def __init__(self, c0, c1, c2, c3, c4):
        super().__init__()
        self.knn = 8
        c = (c0, c1, c2, c3, c4)
        for dst in (4, 3, 2, 1, 0):
            for src in range(5):
                if src != dst:
                    setattr(self, "conv%d%d" % (src, dst), Linear(c[src], c[dst], bn=False))
            setattr(self, "conv%d" % dst, Linear(c[dst], c[dst], bn=False))

    def forward(self, num_point, f0=None, f1=None, f2=None, f3=None, f4=None, FPS_0=None, FPS_1=None,
                FPS_2=None, FPS_3=None, knn_0=None, knn_1=None, knn_2=None, knn_3=None, knn_4=None,
                xyz0=None, xyz1=None, xyz2=None, xyz3=None, xyz4=None):
        k = self.knn
        # the reference dispatches on the literal counts 128/256/512/1024/2048 (N=2048 only);
        # matching num_point against the five states' own sizes is the same at N=2048.
        lvl = [f.shape[1] for f in (f0, f1, f2, f3, f4)].index(num_point)
        if lvl == 4:
            t = (self.conv04(index_points(f0, _compose(FPS_0, FPS_1, FPS_2, FPS_3)))
                 , self.conv14(index_points(f1, _compose(FPS_1, FPS_2, FPS_3)))
                 , self.conv24(index_points(f2, _compose(FPS_2, FPS_3)))
                 , self.conv34(index_points(f3, FPS_3)))
            f4 = self.conv4(f4 + t[0] + t[1] + t[2] + t[3]) + f4
        if lvl == 3:
            t = (self.conv03(index_points(f0, _compose(FPS_0, FPS_1, FPS_2)))
                 , self.conv13(index_points(f1, _compose(FPS_1, FPS_2)))
                 , self.conv23(index_points(f2, FPS_2))
                 , self.conv43(upsample(f4, knn_4)))
            f3 = self.conv3(f3 + t[0] + t[1] + t[2] + t[3]) + f3
        if lvl == 2:
            t0 = self.conv02(index_points(f0, _compose(FPS_0, FPS_1)))
            t1 = self.conv12(index_points(f1, FPS_1))
            t2 = self.conv32(upsample(f3, knn_3))
            t3 = self.conv42(upsample(f4, knn_point(k, xyz2, xyz4)[1], scale_ratio=4))
            f2 = self.conv2(f2 + t0 + t1 + t2 + t3) + f2
        if lvl == 1:
            t0 = self.conv01(index_points(f0, FPS_0))
            t1 = self.conv21(upsample(f2, knn_2))
            t2 = self.conv31(upsample(f3, knn_point(k, xyz1, xyz3)[1], scale_ratio=4))
            t3 = self.conv41(upsample(f4, knn_point(k, xyz1, xyz4)[1], scale_ratio=8))
            f1 = self.conv1(f1 + t0 + t1 + t2 + t3) + f1
        if lvl == 0:
            t0 = self.conv10(upsample(f1, knn_1))
            t1 = self.conv20(upsample(f2, knn_point(k, xyz0, xyz2)[1], scale_ratio=4))
            t2 = self.conv30(upsample(f3, knn_point(k, xyz0, xyz3)[1], scale_ratio=8))
            t3 = self.conv40(upsample(f4, knn_point(k, xyz0, xyz4)[1], scale_ratio=16))
            f0 = self.conv0(f0 + t0 + t1 + t2 + t3) + f0
        return f0, f1, f2, f3, f4


class KeepHighResolutionModule(nn.Module):
    """cls encoder, modules/repsurface_utils.py:542-639"""
    LEVELS = (512, 256, 128, 64, 32)

    def __init__(self, data_C, b1_C, b2_C, b3_C, b4_C, cuda=False):
        super().__init__()
        self.cuda_ops = cuda
        self.drop = nn.Dropout(0.5)
        self.la0 = LocalMergeCls(32, 64, 8, residual=True)
        self.la1 = LocalMergeCls(64, 64, 8, residual=False)
        self.la2 = LocalMergeCls(64, 64, 8, residual=False)
        self.la3 = LocalMergeCls(64, 128, 8, residual=True)
        self.la4 = LocalMergeCls(128, 256, 8, residual=True)
        self.la5 = LocalMergeCls(256, 512, 8, residual=True)
        self.start = Linear(3, 32, bn=False)
        self.conv3 = Linear(512, 512, bn=False)
        self.conv4 = Linear(512, 1024, bn=False)
        self.final = Linear(512, 1024, bn=False)
        self.final_class = nn.Linear(2048, 1024)
        self.bn = nn.BatchNorm1d(1024)
        self.lrelu = nn.LeakyReLU(negative_slope=0.2)

    def forward(self, xyz, normal, trace=None):
        xyz = xyz.transpose(1, 2).contiguous()
        feat, _, idx, _ = self.la0(xyz=xyz, base_xyz=xyz)
        if trace is not None:
            trace.update(knn0=idx, f0=feat)
        base = xyz
        for lvl, (S, la) in enumerate(zip(self.LEVELS, (self.la1, self.la2, self.la3, self.la4, self.la5))):
            fps = farthest_point_sample(base, S)
            sub = index_points(base, fps)
            feat, _, idx, _ = la(xyz=sub, base_xyz=base, feature=feat, FPS_idx=fps)
            if trace is not None:
                trace.update({"fps%d" % lvl: fps, "knn%d" % (lvl + 1): idx, "f%d" % (lvl + 1): feat})
            base = sub
        y = self.conv4(self.conv3(feat)).transpose(1, 2).contiguous()
        pooled = torch.cat((y.max(-1)[0], y.mean(-1)), 1)
        return self.lrelu(self.bn(self.final_class(pooled)))


class KeepHighResolutionModulePartSeg(nn.Module):
    """seg encoder-decoder, modules/pointnet2_utils.py:711-858"""

    def __init__(self, data_C, b1_C, b2_C, b3_C, b4_C, cuda=False):
        super().__init__()
        self.neighbour = 16
        self.cuda_ops = cuda
        self.start = Linear(3, 32, bn=False)
        self.la0 = LocalMergeSeg(32, 64, 8, residual=True)
        self.la1 = LocalMergeSeg(64, 64, 8, residual=False)
        self.la2 = LocalMergeSeg(64, 64, 8, residual=False)
        self.la3 = LocalMergeSeg(64, 128, 8, residual=True)
        self.la4 = LocalMergeSeg(128, 256, 8, residual=True)
        self.la4_up = LocalMergeSeg(128, 128, 8, residual=False)
        self.la3_up = LocalMergeSeg(64, 64, 8, residual=False)
        self.la2_up = LocalMergeSeg(64, 64, 8, residual=False)
        self.la1_up = LocalMergeSeg(64, 64, 8, residual=False)
        self.up_conv4 = Linear(256, 128, bn=False)
        self.up_conv3 = Linear(128, 64, bn=False)
        self.up_conv2 = Linear(64, 64, bn=False)
        self.up_conv1 = Linear(64, 64, bn=False)
        self.mlp = Linear(256, 256, bn=False)
        self.conv5 = Linear(64, 256, bn=False)
        self.conv6 = Linear(64, 128, bn=False)
        self.conv7 = Linear(16, 64, bn=False)
        self.conv8 = Linear(64, 256, bn=False)
        for i in range(1, 6):
            setattr(self, "fuse%d" % i, Fuse(64, 64, 64, 128, 256))
        self.lrelu = nn.LeakyReLU(negative_slope=0.2)

    def forward(self, xyz, normal, label):
        x0 = xyz.transpose(1, 2).contiguous()
        nrm = normal.transpose(1, 2).contiguous()
        N = x0.shape[1]
        e0, n0, k0, _ = self.la0(xyz=x0, base_xyz=x0, normal=nrm)
        p0 = farthest_point_sample(x0, N // 2)
        x1 = index_points(x0, p0)
        e1, n1, k1, _ = self.la1(xyz=x1, base_xyz=x0, normal=n0, feature=e0, FPS_idx=p0)
        p1 = farthest_point_sample(x1, N // 4)
        x2 = index_points(x1, p1)
        e2, n2, k2, _ = self.la2(xyz=x2, base_xyz=x1, normal=n1, feature=e1, FPS_idx=p1)
        p2 = farthest_point_sample(x2, N // 8)
        x3 = index_points(x2, p2)
        e3, n3, k3, _ = self.la3(xyz=x3, base_xyz=x2, normal=n2, feature=e2, FPS_idx=p2)
        p3 = farthest_point_sample(x3, N // 16)
        x4 = index_points(x3, p3)
        e4, n4, k4, _ = self.la4(xyz=x4, base_xyz=x3, normal=n3, feature=e3, FPS_idx=p3)

        geo = dict(FPS_0=p0, FPS_1=p1, FPS_2=p2, FPS_3=p3, knn_0=k0, knn_1=k1, knn_2=k2, knn_3=k3, knn_4=k4,
                   xyz0=x0, xyz1=x1, xyz2=x2, xyz3=x3, xyz4=x4)
        d4 = self.mlp(e4)
        d4 = self.fuse1(N // 16, f0=e0, f1=e1, f2=e2, f3=e3, f4=d4, **geo)[4]
        d3 = self.la4_up(xyz=x3, base_xyz=x3, normal=n3, feature=self.up_conv4(upsample(d4, k4)))[0]
        d3 = self.fuse2(N // 8, f0=e0, f1=e1, f2=e2, f3=d3, f4=e4, **geo)[3]
        d2 = self.la3_up(xyz=x2, base_xyz=x2, normal=n2, feature=self.up_conv3(upsample(d3, k3)))[0]
        d2 = self.fuse3(N // 4, f0=e0, f1=e1, f2=d2, f3=e3, f4=e4, **geo)[2]
        d1 = self.la2_up(xyz=x1, base_xyz=x1, normal=n1, feature=self.up_conv2(upsample(d2, k2)))[0]
        d1 = self.fuse4(N // 2, f0=e0, f1=d1, f2=e2, f3=e3, f4=e4, **geo)[1]
        d0 = self.la1_up(xyz=x0, base_xyz=x0, normal=n0, feature=self.up_conv1(upsample(d1, k1)))[0]
        d0 = self.fuse5(N, f0=d0, f1=e1, f2=e2, f3=e3, f4=e4, **geo)[0]

        glob = torch.cat([t.max(1, keepdim=True)[0] for t in (d0, d1, d2, d3, d4)], dim=2).repeat(1, N, 1)
        lab = self.conv7(label).repeat(1, N, 1)
        return x0, torch.cat((self.conv5(d0), glob, lab), 2)


class PointNetFeaturePropagation(nn.Module):
    def __init__(self, in_channel, mlp, act=False):
        super().__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv1d(last, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm1d(out_channel))
            last = out_channel
        self.act = act
        self.conv = Linear(in_channel, out_channel, bn=False, act=act)

    def forward(self, xyz1, xyz2, points1, points2):
        return self.conv(three_interpolate(xyz1, xyz2, points2))


# ----------------------------------------------------------------------------- L2 wiring
class ClsModel(nn.Module):
    """models/repsurf/repsurf_ssg_umb.py:35-70 (class Model)"""

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

    def forward(self, points, trace=None):
        center = points[:, :3, :]
        x = self.keepHigh(center, center, trace=trace)
        x = self.drop1(self.lrelu(self.bn1(self.fc1(x))))
        x = self.drop2(self.lrelu(self.bn2(self.fc2(x))))
        return F.log_softmax(self.fc3(x), -1)


class PartSegModel(nn.Module):
    """models/repsurf/pointnet2_part_seg_msg.py:33-156 (class get_model)"""

    def __init__(self, num_classes, normal_channel=False):
        super().__init__()
        self.normal_channel = normal_channel
        self.keepHigh = KeepHighResolutionModulePartSeg(3, 64, 128, 256, 512, cuda=True)
        self.conv8 = Linear(896, 512, bn=False)
        self.conv9 = Linear(512, 256, bn=False)
        self.conv10 = Linear(256, 128, bn=False)
        self.conv11 = nn.Linear(128, num_classes)
        self.drop1 = nn.Dropout(0.5)
        self.drop2 = nn.Dropout(0.5)

    def forward(self, xyz, cls_label):
        _, feats = self.keepHigh(xyz, normal=xyz, label=cls_label)
        x = self.drop1(self.conv8(feats))
        return self.conv11(self.conv10(self.conv9(x))), xyz


def smooth_cls_loss(pred, target, eps=0.1):
    """util/utils.py:74-88 -- pred are log-probabilities [B, n_class]."""
    n = pred.size(1)
    one_hot = torch.zeros_like(pred).scatter(1, target.view(-1, 1), 1)
    one_hot = one_hot * (1 - eps) + (1 - one_hot) * eps / (n - 1)
    return -(one_hot * pred).sum(dim=1).mean()


def partseg_loss(pred, target, eps=0.1):
    """pointnet2_part_seg_msg.py:159-180 -- pred are logits [M, n_class]."""
    return smooth_cls_loss(F.log_softmax(pred, dim=1), target.contiguous().view(-1), eps)


# ----------------------------------------------------------------------------- umbrella front-end
def xyz2sphere(xyz, normalize=True):
    """(rho, theta, phi) of [..., 3] coordinates; theta is 0 where rho is 0; normalised to [0, 1]."""
    rho = xyz.pow(2).sum(-1, keepdim=True).sqrt().clamp(min=0)
    theta = torch.acos(xyz[..., 2:3] / rho)
    phi = torch.atan2(xyz[..., 1:2], xyz[..., 0:1])
    theta = torch.where(rho == 0, torch.zeros_like(theta), theta)
    if normalize:
        theta = theta / math.pi
        phi = phi / (2 * math.pi) + .5
    return torch.cat([rho, theta, phi], -1)


def group_by_umbrella(xyz, new_xyz, k=9):
    """[B,N',k-1,3 (triangle vertices: centre, p_i, p_i+1), 3]: the k-1 nearest neighbours of every
    point (the nearest -- the point itself -- dropped), relative to the point, sorted by azimuth,
    each paired with its successor (cyclically)."""
    idx = knn_point(k, xyz, new_xyz)[1]
    rel = index_points(xyz, idx)[:, :, 1:] - new_xyz.unsqueeze(-2)
    order = xyz2sphere(rel)[..., 2].argsort(dim=-1)
    srt = torch.gather(rel, 2, order.unsqueeze(-1).expand(-1, -1, -1, 3)).unsqueeze(-2)
    return torch.cat([torch.zeros_like(srt), srt, torch.roll(srt, -1, dims=-3)], dim=-2)


def cal_normal_umb(group_xyz, random_inv=False):
    """Unit normals of the triangles [B,N,G,3,3]; all G normals of a point are flipped by the sign of
    the FIRST triangle's x component; optional per-cloud random flip drawn from the CPU generator."""
    nor = torch.cross(group_xyz[..., 1, :] - group_xyz[..., 0, :], group_xyz[..., 2, :] - group_xyz[..., 0, :], dim=-1)
    unit = nor / torch.norm(nor, dim=-1, keepdim=True)
    unit = unit * ((unit[..., 0:1, 0] > 0).float() * 2. - 1.).unsqueeze(-1)
    if random_inv:
        rnd = (torch.randint(0, 2, (group_xyz.size(0), 1, 1)).float() * 2. - 1.).to(unit.device)
        unit = unit * rnd.unsqueeze(-1)
    return unit


def check_nan_umb(normal, center, pos):
    """Triangles whose normal is NaN (degenerate) take the values of the point's first valid triangle."""
    B, N, G, _ = normal.shape
    bad = torch.isnan(normal).any(-1)
    first = torch.argmax((~bad).int(), dim=-1)
    pick = first.view(B, N, 1, 1)
    out = []
    for t in (normal, center, pos):
        rep = torch.gather(t, 2, pick.expand(-1, -1, 1, t.shape[-1])).expand(-1, -1, G, -1)
        out.append(torch.where(bad.unsqueeze(-1), rep, t))
    return out


class UmbrellaSurfaceConstructor(nn.Module):
    """Umbrella surface features (centre 3 | polar 3 | normal 3 | position 1) of the k-1 triangles
    around every point, three 1x1 convolutions (BN + ReLU after the first two), summed over the
    triangles.  [B,3,N] -> [B,10,N]."""

    def __init__(self, k, in_channel, aggr_type='sum', return_dist=False, random_inv=True, cuda=False):
        super().__init__()
        self.k, self.return_dist, self.random_inv, self.aggr_type = k, return_dist, random_inv, aggr_type
        self.mlps = nn.Sequential(
            nn.Conv2d(in_channel, in_channel, 1, bias=False), nn.BatchNorm2d(in_channel), nn.ReLU(True),
            nn.Conv2d(in_channel, in_channel, 1, bias=True), nn.BatchNorm2d(in_channel), nn.ReLU(True),
            nn.Conv2d(in_channel, in_channel, 1, bias=True))

    def features(self, center):
        tri = group_by_umbrella(center, center, k=self.k)
        normal = cal_normal_umb(tri, random_inv=self.random_inv)
        cen = tri.mean(dim=-2)
        polar = xyz2sphere(cen)
        if self.return_dist:
            pos = (normal * cen).sum(-1, keepdim=True) / torch.sqrt(torch.tensor([3.0]))
            normal, cen, pos = check_nan_umb(normal, cen, pos)
            return torch.cat([cen, polar, normal, pos], dim=-1)
        normal, cen, _ = check_nan_umb(normal, cen, cen[..., :1])
        return torch.cat([cen, polar, normal], dim=-1)

    def forward(self, center):
        f = self.mlps(self.features(center.permute(0, 2, 1)).permute(0, 3, 2, 1))
        if self.aggr_type == 'max':
            return f.max(2)[0]
        return f.mean(2) if self.aggr_type == 'avg' else f.sum(2)


# ----------------------------------------------------------------------------- RepSurf baseline (a14)
def sample_and_group(npoint, radius, nsample, center, normal, feature, return_normal=True, return_polar=False):
    """modules/repsurface_utils.py:12-56."""
    fps_idx = farthest_point_sample(center, npoint)
    new_center, new_normal = index_points(center, fps_idx), index_points(normal, fps_idx)
    idx = query_ball_point(radius, nsample, center, new_center)
    rel = index_points(center, idx) - new_center.unsqueeze(2)
    if return_polar:
        rel = torch.cat([rel, xyz2sphere(rel)], dim=-1)
    parts = [rel, index_points(normal, idx)]
    if feature is not None:
        parts = [rel, index_points(normal, idx), index_points(feature, idx)] if return_normal else \
            [rel, index_points(feature, idx)]
    return new_center, new_normal, torch.cat(parts, dim=-1)


def sample_and_group_all(center, normal, feature, return_normal=True, return_polar=False):
    """modules/repsurface_utils.py:58-84."""
    B, N, C = normal.shape
    zero = torch.zeros(B, 1, 3)
    gc = center.view(B, 1, N, 3)
    if return_polar:
        gc = torch.cat([gc, xyz2sphere(gc)], dim=-1)
    parts = [gc, normal.view(B, 1, N, C), feature.view(B, 1, N, -1)] if return_normal else [gc, feature.view(B, 1, N, -1)]
    return zero, zero, torch.cat(parts, dim=-1)


class SurfaceAbstractionCD(nn.Module):
    """modules/repsurface_utils.py:256-319."""

    def __init__(self, npoint, radius, nsample, feat_channel, pos_channel, mlp, group_all, return_normal=True,
                 return_polar=False, cuda=False):
        super().__init__()
        self.npoint, self.radius, self.nsample, self.group_all = npoint, radius, nsample, group_all
        self.return_normal, self.return_polar, self.pos_channel = return_normal, return_polar, pos_channel
        self.mlp_convs, self.mlp_bns = nn.ModuleList(), nn.ModuleList()
        self.mlp_l0, self.mlp_f0 = nn.Conv2d(pos_channel, mlp[0], 1), nn.Conv2d(feat_channel, mlp[0], 1)
        self.bn_l0, self.bn_f0 = nn.BatchNorm2d(mlp[0]), nn.BatchNorm2d(mlp[0])
        last = mlp[0]
        for oc in mlp[1:]:
            self.mlp_convs.append(nn.Conv2d(last, oc, 1))
            self.mlp_bns.append(nn.BatchNorm2d(oc))
            last = oc

    def forward(self, center, normal, feature):
        normal, center = normal.permute(0, 2, 1), center.permute(0, 2, 1)
        feature = feature.permute(0, 2, 1) if feature is not None else None
        if self.group_all:
            nc, nn_, nf = sample_and_group_all(center, normal, feature, self.return_normal, self.return_polar)
        else:
            nc, nn_, nf = sample_and_group(self.npoint, self.radius, self.nsample, center, normal, feature,
                                           self.return_normal, self.return_polar)
        nf = nf.permute(0, 3, 2, 1)
        x = F.relu(self.bn_l0(self.mlp_l0(nf[:, :self.pos_channel])) + self.bn_f0(self.mlp_f0(nf[:, self.pos_channel:])))
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            x = F.relu(bn(conv(x)))
        return nc.permute(0, 2, 1), nn_.permute(0, 2, 1), x.max(2)[0]


class RepSurf2xModel(nn.Module):
    """models/repsurf/repsurf_ssg_umb_2x.py:11-61."""

    def __init__(self, args):
        super().__init__()
        cc = 0 if not args.return_center else (6 if args.return_polar else 3)
        self.surface_constructor = UmbrellaSurfaceConstructor(args.group_size + 1, 10, return_dist=args.return_dist,
                                                              aggr_type=args.umb_pool)
        kw = dict(pos_channel=cc, return_polar=args.return_polar)
        self.sa1 = SurfaceAbstractionCD(512, 0.1, 24, 10, mlp=[128, 128, 256], group_all=False, **kw)
        self.sa2 = SurfaceAbstractionCD(128, 0.2, 24, 256 + 10, mlp=[256, 256, 512], group_all=False, **kw)
        self.sa3 = SurfaceAbstractionCD(32, 0.4, 24, 512 + 10, mlp=[512, 512, 1024], group_all=False, **kw)
        self.sa4 = SurfaceAbstractionCD(None, None, None, 1024 + 10, mlp=[1024, 1024, 2048], group_all=True, **kw)
        self.classfier = nn.Sequential(nn.Linear(2048, 512), nn.BatchNorm1d(512), nn.ReLU(True), nn.Dropout(0.4),
                                       nn.Linear(512, 256), nn.BatchNorm1d(256), nn.ReLU(True), nn.Dropout(0.4),
                                       nn.Linear(256, args.num_class))

    def forward(self, points):
        center = points[:, :3, :]
        normal = self.surface_constructor(center)
        center, normal, feature = self.sa1(center, normal, None)
        center, normal, feature = self.sa2(center, normal, feature)
        center, normal, feature = self.sa3(center, normal, feature)
        center, normal, feature = self.sa4(center, normal, feature)
        return F.log_softmax(self.classfier(feature.view(-1, 2048)), -1)
