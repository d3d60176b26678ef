"""Drop-in for the reference's modules/pointnet2_utils.py on MI355X: the same operator names,
nn.Module classes, constructor signatures and state-dict keys, with the device work done by
the gfx950 kernels of libmpa_hip.so (see ../ops.py).

Reference locations (Markov_Process_Analysis_on_Point_Cloud/modules/pointnet2_utils.py):
  upsample:13  mod_index:53  index_points:64  farthest_point_sample:84  query_ball_point:112
  sample_and_group:137  sample_and_group_all:168  square_distance:190  knn_point:211  knn_point2:224
  random_sample:253  convert_polar:263  resort_points:289  group_by_umbrella:309
  UmbrellaSurfaceConstructor:333  Linear:401  LocalMerge:427  LocalTrans:479  Fuse:576
  KeepHighResolutionModulePartSeg:711  PointNetFeaturePropagation:860
Every public name of the reference file exists here, so the reference's own model files import over
this module unchanged (tests/test_cabi_cpu.py runs that recipe).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..ops import (farthest_point_sample, index_points, knn_point, query_ball_point, query_knn_point,  # noqa: F401
                   sample, square_distance, three_interpolate, three_nn, upsample)


def mod_index(bse_xyz, mod_idx, xyz):
    """reference :53-61 -- rows mod_idx[b] of bse_xyz [B,N,D] replaced by xyz [B,M,D]."""
    out = bse_xyz.clone()
    bidx = torch.arange(mod_idx.shape[0], device=bse_xyz.device).unsqueeze(-1).expand_as(mod_idx)
    out[bidx.reshape(-1), mod_idx.reshape(-1), :] = xyz.reshape(-1, xyz.shape[-1]).to(out.dtype)
    return out


def sample_and_group_all(xyz, points):
    """reference :168-186 -- the whole cloud as one group around the origin."""
    B, N, C = xyz.shape
    new_xyz = torch.zeros(B, 1, C, device=xyz.device)
    grouped_xyz = xyz.view(B, 1, N, C)
    if points is not None:
        return new_xyz, torch.cat([grouped_xyz, points.view(B, 1, N, -1)], dim=-1)
    return new_xyz, grouped_xyz


def knn_point2(nsample, xyz, new_xyz):
    """reference :224-251 (unused by the models): top-k of the distance matrix with exact zeros
    replaced by 10 + N(0,1) noise and the diagonal zeroed.  The noise comes from the device generator,
    so results are random where zeros occur (parity unpinned by construction)."""
    sq = square_distance(new_xyz, xyz)
    B, N, _ = sq.shape
    noise = torch.randn(sq.shape, device=sq.device)
    sq = torch.where(sq == 0, 10 + noise, sq)
    off_diag = 1.0 - torch.eye(N, sq.shape[2], device=sq.device).unsqueeze(0)
    return torch.topk(sq * off_diag, nsample, dim=-1, largest=False, sorted=True)


def random_sample(xyz, sample_num):
    """reference :253-261 -- one random permutation (CPU generator) shared by the whole batch."""
    B, N, _ = xyz.shape
    perm = torch.randperm(N)
    idx = perm[:sample_num].to(xyz.device)
    return xyz[:, idx, :], idx.unsqueeze(0).expand(B, sample_num)


def convert_polar(neighbours, center):
    """reference :263-287 -- axis-wise polar angles of [B,3,S,K] neighbour offsets (keeps the
    reference's r_yz = sqrt(y^2 + y^2))."""
    rel = (neighbours - center).permute(0, 2, 3, 1).contiguous()
    x, y, z = rel[..., 0], rel[..., 1], rel[..., 2]
    r_xy, r_zx, r_yz = torch.sqrt(x ** 2 + y ** 2), torch.sqrt(z ** 2 + x ** 2), torch.sqrt(y ** 2 + y ** 2)
    u = lambda t: t.unsqueeze(-3).contiguous()
    z_beta, z_alpha = u(torch.atan2(z, r_xy)), u(torch.atan2(y, x))
    y_beta, y_alpha = u(torch.atan2(y, r_zx)), u(torch.atan2(x, z))
    x_beta, x_alpha = u(torch.atan2(x, r_yz)), u(torch.atan2(z, y))
    return x_alpha, x_beta, y_alpha, y_beta, z_alpha, z_beta


def resort_points(points, idx):
    """reference :289-307 -- points [B,N,G,C] re-ordered along G by idx [B,N,G]."""
    return torch.gather(points, 2, idx.unsqueeze(-1).expand(-1, -1, -1, points.shape[-1]))


def group_by_umbrella(xyz, new_xyz, k=9, cuda=False):
    """reference :309-331 (== repsurface_utils.py:106-126).  [B,N',k-1,3 (centre, p_i, p_i+1),3]: the
    k-1 nearest neighbours of each point of new_xyz (the nearest dropped), relative to it, sorted by
    azimuth, paired cyclically."""
    from .polar_utils import xyz2sphere
    idx = query_knn_point(k, xyz, new_xyz)
    rel = index_points(xyz, idx)[:, :, 1:] - new_xyz.unsqueeze(-2)
    order = xyz2sphere(rel)[..., 2].argsort(dim=-1, stable=True)
    srt = resort_points(rel, order).unsqueeze(-2)
    return torch.cat([torch.zeros_like(srt), srt, torch.roll(srt, -1, dims=-3)], dim=-2)


class UmbrellaSurfaceConstructor(nn.Module):
    """Umbrella-based surface abstraction (reference :333-399 == repsurface_utils.py:321-376): per
    point the k-1 triangles around it -> (centre | polar | normal | position) -> three 1x1 convolutions
    (BatchNorm + ReLU after the first two) -> sum / mean / max over the triangles.  [B,3,N] ->
    [B,in_channel,N].  The triangle features come from one fused kernel (ops.umbrella_features), the
    convolutions run as the MFMA Linear unit over the B*N*(k-1) rows; parameter names equal the
    reference's."""

    def __init__(self, k, in_channel, aggr_type='sum', return_dist=False, random_inv=True, cuda=False):
        super().__init__()
        self.k = k
        self.return_dist = return_dist
        self.random_inv = random_inv
        self.aggr_type = aggr_type
        self.cuda_ops = cuda      # the reference stores this as `self.cuda`, shadowing nn.Module.cuda
        self.mlps = nn.Sequential(
            nn.Conv2d(in_channel, in_channel, 1, bias=False), nn.BatchNorm2d(in_channel), nn.ReLU(True),
            nn.Conv2d(in_channel, in_channel, 1, bias=True), nn.BatchNorm2d(in_channel), nn.ReLU(True),
            nn.Conv2d(in_channel, in_channel, 1, bias=True))

    def forward(self, center):
        center = center.permute(0, 2, 1).contiguous()
        B, N, _ = center.shape
        sign = None
        if self.random_inv:       # per-cloud flip drawn from the CPU generator, as the reference does
            sign = torch.randint(0, 2, (B, 1, 1)).float().view(B) * 2. - 1.
        f = ops.umbrella_features(center, self.k, cloud_sign=sign, return_dist=self.return_dist)   # [B,N,G,CH]
        G, CH = f.shape[2], f.shape[3]
        m = self.mlps
        x = f.view(B, N * G, CH)
        x = ops.linear_bn_act(x, m[0].weight.view(CH, CH), None, m[1], 0.0)
        x = ops.linear_bn_act(x, m[3].weight.view(CH, CH), m[3].bias, m[4], 0.0)
        x = ops.linear(x, m[6].weight.view(CH, CH), m[6].bias).view(B, N, G, CH)
        if self.aggr_type == 'max':
            x = x.max(dim=2)[0]
        elif self.aggr_type == 'avg':
            x = x.mean(dim=2)
        else:
            x = x.sum(dim=2)
        return x.permute(0, 2, 1)


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False):
    """reference :137-165 -- FPS, ball query, centre-relative grouping."""
    B, N, C = xyz.shape
    fps_idx = farthest_point_sample(xyz, npoint)
    new_xyz = index_points(xyz, fps_idx)
    idx = query_ball_point(radius, nsample, xyz, new_xyz)
    grouped_xyz = index_points(xyz, idx)
    grouped_xyz_norm = grouped_xyz - new_xyz.view(B, npoint, 1, C)
    if points is not None:
        new_points = torch.cat([grouped_xyz_norm, index_points(points, idx)], dim=-1)
    else:
        new_points = grouped_xyz_norm
    if returnfps:
        return new_xyz, new_points, grouped_xyz, fps_idx
    return new_xyz, new_points


class Linear(nn.Module):
    """The transition / pointwise MLP unit (reference :401-425): nn.Linear, then LayerNorm
    (`bn=True`, never used by the models) or BatchNorm1d over the B*S rows (`bn=False`), then
    LeakyReLU(0.2) if `act`.  Input must be [B,S,C]."""

    def __init__(self, in_channels, out_channels, bn=True, act=True):
        super().__init__()
        self.act_flag = act
        self.bn_flag = bn
        self.linear = nn.Linear(in_channels, out_channels)
        self.norm1 = nn.LayerNorm(out_channels)
        self.norm2 = nn.BatchNorm1d(out_channels)
        self.act = nn.LeakyReLU(negative_slope=0.2)

    def forward(self, input):
        if input.dim() != 3:
            raise ValueError("Linear expects a [B,S,C] input (as the reference's BatchNorm1d permute does)")
        return self.fused(input, None)

    def fused(self, input, residual):
        """Linear unit with `residual` (or None) added after the activation in the same pass."""
        if self.bn_flag:
            out = self.norm1(self.linear(input))
            out = self.act(out) if self.act_flag else out
            return out if residual is None else residual + out
        return ops.linear_bn_act(input, self.linear.weight, self.linear.bias, self.norm2,
                                 0.2 if self.act_flag else None, residual=residual)


class LocalTrans(nn.Module):
    """Difference-wise attention between a point-set state and its K-neighbourhoods in the
    previous state (reference :479-574)."""

    def __init__(self, in_c, out_c, patch_num, usetanh=False, residual=False):
        super().__init__()
        self.patchNum = patch_num
        self.residual = residual
        self.usetanh = usetanh
        self.out_c = out_c
        self.q = nn.Linear(in_c, out_c)
        self.k = nn.Linear(in_c, out_c)
        self.v = nn.Linear(in_c, out_c)
        self.conv_res = Linear(in_c, out_c, bn=False)
        self.ffn = Linear(out_c, out_c, bn=False)
        self.tanh = nn.Tanh()

    def mpa_adjacent_params(self):
        """Parameters the flat buffers should keep back to back (distributed.GradReducer): the key and
        value projections are applied as one stacked weight (ops.linear_kv)."""
        return ((self.k.weight, self.v.weight), (self.k.bias, self.v.bias))

    def forward(self, features, idx, pos, FPS_idx=None, xyz=False, center=None):
        # `center` (optional, not in the reference signature): index_points(features, FPS_idx) when
        # the caller already has it (LocalMerge gathers it once for both feature streams and the kNN)
        if center is None:
            center = index_points(features, FPS_idx) if FPS_idx is not None else features
        if self.usetanh:
            return self.finish(self._tanh_context(features, idx, center, xyz), center)
        if xyz:
            context = ops.diffattn_xyz(features, center, idx, self.q.weight, self.q.bias, self.k.weight,
                                       self.k.bias, self.v.weight, self.v.bias)
        else:
            # a shift of q (or of every k_j) leaves softmax_j(q - k_j) unchanged: dL/dbq = dL/dbk = 0
            q = ops.linear(center, self.q.weight, self.q.bias, bias_grad_is_zero=True)
            kv = ops.linear_kv(features, self.k, self.v)
            context = ops.diffattn(q, kv, idx)
        return self.finish(context, center)

    def _tanh_context(self, features, idx, center, xyz):
        """`usetanh=True` (reference :535-537, :560-562; never enabled by the models): attention =
        tanh(q - k) / patchNum, context = matmul(attention, value).squeeze(-2) exactly as written there --
        a [K,C] x [K,C] product, so like the reference it only runs when C == K."""
        q = ops.linear(center, self.q.weight, self.q.bias).unsqueeze(-2)
        if xyz:
            rel = index_points(features, idx) - center.unsqueeze(-2)
            key = ops.linear(rel, self.k.weight, self.k.bias)
            val = ops.linear(rel, self.v.weight, self.v.bias)
        else:
            key = index_points(ops.linear(features, self.k.weight, self.k.bias), idx)
            val = index_points(ops.linear(features, self.v.weight, self.v.bias), idx)
        attention = torch.tanh(q - key) / self.patchNum
        return torch.matmul(attention, val).squeeze(-2)

    def finish(self, context, center):
        """out = res + Linear_ffn(context), res = conv_res(centre) if `residual` else the centre."""
        residual = self.conv_res(center) if self.residual else center
        return self.ffn.fused(context, residual)


def stacked_param_groups(t1, t2):
    """The parameter groups local_trans_pair reads as stacked weights (kept back to back in the flat
    parameter buffers by distributed.GradReducer / optim.FlatAdam)."""
    return ((t1.k.weight, t1.v.weight, t2.k.weight, t2.v.weight), (t1.k.bias, t1.v.bias, t2.k.bias, t2.v.bias),
            (t1.q.weight, t2.q.weight), (t1.q.bias, t2.q.bias))


def local_trans_pair(t1, t2, features, idx1, idx2, center, concat=False, kvkv=None):
    """t1(features, idx1), t2(features, idx2) for two feature-branch LocalTrans blocks that share
    their base rows and centres (LocalMerge's two feature streams): the four key/value projections
    are one GEMM over the base rows, the two query projections one GEMM over the centres, and the
    attention backward hands each stacked projection a single gradient."""
    if t1.usetanh or t2.usetanh:
        outs = t1(features, idx1, None, center=center), t2(features, idx2, None, center=center)
        return torch.cat(outs, 2) if concat else outs
    # the centres have three readers (query projections, both residuals): one alias each, so that their gradients
    # meet in one summing launch (ops.fanout) instead of two adds
    cen_q, cen_1, cen_2 = ops.fanout(center, 3)
    qq = ops.linear_stack(cen_q, (t1.q, t2.q), (True, True))
    if kvkv is None:        # (LocalMerge hands the projections over when it computed them together with the centres)
        kvkv = ops.linear_stack(features, (t1.k, t1.v, t2.k, t2.v), (True, False, True, False))
    c1, c2 = ops.diffattn_pair(qq, kvkv, idx1, idx2)
    return finish_group((t1, t2), (c1, c2), (cen_1, cen_2), concat=concat)


def centres_and_projections(t1, t2, feature, FPS_idx):
    """(fs, kvkv): the sampled centres index_points(feature, FPS_idx) and the stacked key | value projections of two
    feature streams, from one autograd node (ops.gather_and_stack: the features receive one gradient instead of two
    that autograd has to add).  (fs, None) when that form does not apply."""
    if FPS_idx is None or t1.usetanh or t2.usetanh:
        return (feature if FPS_idx is None else index_points(feature, FPS_idx)), None
    return ops.gather_and_stack(feature, FPS_idx, (t1.k, t1.v, t2.k, t2.v), (True, False, True, False))


def _unit_group(units, xs, residuals=None, mode="each"):
    """Independent `Linear` units (BatchNorm form) through ops.linear_bn_act_group: their forward products are one
    launch and their input-gradient products another.  mode "each": [unit(x)]; "concat": the outputs side by side
    (torch.cat(..., -1) written in place); "chain": residuals[0] + sum of the units."""
    if any(u.bn_flag for u in units):                          # LayerNorm form (never built by the models)
        if mode == "chain":
            acc = residuals[0]
            for u, x in zip(units, xs):
                acc = acc + u(x)
            return acc
        outs = [u.fused(x, None if residuals is None else residuals[i]) for i, (u, x) in enumerate(zip(units, xs))]
        return torch.cat(outs, -1) if mode == "concat" else outs
    return ops.linear_bn_act_group(list(xs), [u.linear for u in units], [u.norm2 for u in units],
                                   [0.2 if u.act_flag else None for u in units], residuals=residuals, mode=mode)


def finish_group(trans, contexts, centers, concat=False):
    """[t.finish(context, centre)] for parallel LocalTrans streams (concat=True: their concatenation along the
    channels, as LocalMerge feeds it to fc2): the conv_res units of the streams that have one run as a group, then
    the ffn units (each with its stream's residual)."""
    res = list(centers)
    by_dtype = {}
    for i, t in enumerate(trans):
        if t.residual:
            by_dtype.setdefault(centers[i].dtype, []).append(i)          # (the xyz stream's centres stay fp32)
    for members in by_dtype.values():
        outs = _unit_group([trans[i].conv_res for i in members], [centers[i] for i in members])
        for i, o in zip(members, outs):
            res[i] = o
    return _unit_group([t.ffn for t in trans], list(contexts), residuals=res, mode="concat" if concat else "each")


class LocalMerge(nn.Module):
    """State -> state probability-transition block, part-seg variant (reference :427-477):
    xyz-space and feature-space neighbourhoods, three attention streams, fc2 over 3*out."""

    def __init__(self, in_channels, out_channels, knn, usetanh=False, residual=False):
        super().__init__()
        self.knn = knn
        self.usetanh = usetanh
        self.residual = residual
        self.fc2 = Linear(out_channels * 3, out_channels, bn=False)
        self.xyz_Trans = LocalTrans(3, out_channels, knn, usetanh=usetanh, residual=True)
        self.normal_Trans = LocalTrans(10, out_channels, knn, usetanh=usetanh, residual=True)
        self.feature_Trans1 = LocalTrans(in_channels, out_channels, knn, usetanh=usetanh, residual=residual)
        self.feature_Trans2 = LocalTrans(in_channels, out_channels, knn, usetanh=usetanh, residual=residual)

    def mpa_adjacent_params(self):
        return stacked_param_groups(self.feature_Trans1, self.feature_Trans2)

    def forward(self, xyz, base_xyz, normal=None, feature=None, FPS_idx=None, xyz_flag=True, geometry=None):
        # `geometry` (optional, not in the reference signature): this level's precomputed
        # (dist, idx) of knn_point(self.knn, base_xyz, xyz) from ops.geometry_pass
        # `geometry`: a level of ops.GeometryChain -- it issues this state's searches together with the NEXT state's
        # sampling (one launch: the FPS chain hides behind the feature-space search)
        if feature is None:
            dist, idx = knn_point(self.knn, base_xyz, xyz) if geometry is None else geometry.xyz_search()
            merge_features = self.xyz_Trans(features=xyz, idx=idx, pos=base_xyz, FPS_idx=FPS_idx, xyz=True)
        else:
            fs, kvkv = centres_and_projections(self.feature_Trans1, self.feature_Trans2, feature, FPS_idx)
            if geometry is None:
                dist, idx = knn_point(self.knn, base_xyz, xyz)
                _, idx_feature = knn_point(self.knn, feature, fs)
            else:
                (dist, idx), idx_feature = geometry.search(self.knn, feature, fs)
            merge_features = self.fc2(self._three_streams(base_xyz, feature, idx, idx_feature, FPS_idx, fs, kvkv))
        if FPS_idx is not None:
            normal = index_points(normal, FPS_idx)
        return merge_features, normal, idx, dist

    def _three_streams(self, base_xyz, feature, idx, idx_feature, FPS_idx, fs, kvkv=None):
        """torch.cat((xyz_Trans(base_xyz, idx), feature_Trans1(feature, idx), feature_Trans2(feature, idx_feature)), 2):
        the attention contexts per stream as before, their closing conv_res / ffn units as groups whose outputs land
        side by side in one tensor."""
        tx, t1, t2 = self.xyz_Trans, self.feature_Trans1, self.feature_Trans2
        if tx.usetanh or t1.usetanh or t2.usetanh:
            xyz_f = tx(features=base_xyz, idx=idx, pos=base_xyz, FPS_idx=FPS_idx, xyz=True)
            f1, f2 = local_trans_pair(t1, t2, feature, idx, idx_feature, fs, kvkv=kvkv)
            return torch.cat((xyz_f, f1, f2), dim=2)
        cx = index_points(base_xyz, FPS_idx) if FPS_idx is not None else base_xyz
        ctx_x = ops.diffattn_xyz(base_xyz, cx, idx, tx.q.weight, tx.q.bias, tx.k.weight, tx.k.bias, tx.v.weight,
                                 tx.v.bias)
        fs_q, fs_1, fs_2 = ops.fanout(fs, 3)           # (three readers of the centres: see local_trans_pair)
        qq = ops.linear_stack(fs_q, (t1.q, t2.q), (True, True))
        if kvkv is None:
            kvkv = ops.linear_stack(feature, (t1.k, t1.v, t2.k, t2.v), (True, False, True, False))
        c1, c2 = ops.diffattn_pair(qq, kvkv, idx, idx_feature)
        return finish_group((tx, t1, t2), (ctx_x, c1, c2), (cx, fs_1, fs_2), concat=True)


def _compose(*maps):
    """FPS_a[FPS_b[...]]: index of a coarser state's points in a finer state."""
    out = maps[-1]
    for m in reversed(maps[:-1]):
        out = torch.gather(m, 1, out)
    return out


class Fuse(nn.Module):
    """Cross-state fusion (reference :576-709): bring every other state's features to the target
    state (composed FPS gathers downwards, `upsample` upwards), one Linear each, sum, Linear,
    residual.  The reference selects the target by the literal counts 128..2048; here the target
    is the state whose size equals `num_point`, which is identical at N=2048 and also works for
    other cloud sizes."""

    def __init__(self, c0, c1, c2, c3, c4):
        super().__init__()
        self.knn = 8
        c = (c0, c1, c2, c3, c4)
        for dst in (4, 3, 2, 1, 0):
            for src in range(5):
                if src != dst:
                    setattr(self, "conv%d%d" % (src, dst), Linear(c[src], c[dst], bn=False))
            setattr(self, "conv%d" % dst, Linear(c[dst], c[dst], bn=False))

    def forward(self, num_point, f0=None, f1=None, f2=None, f3=None, f4=None, FPS_0=None, FPS_1=None, FPS_2=None,
                FPS_3=None, knn_0=None, knn_1=None, knn_2=None, knn_3=None, knn_4=None, xyz0=None, xyz1=None,
                xyz2=None, xyz3=None, xyz4=None):
        f = [f0, f1, f2, f3, f4]
        fps = [FPS_0, FPS_1, FPS_2, FPS_3]
        knn = [knn_0, knn_1, knn_2, knn_3, knn_4]
        xyz = [xyz0, xyz1, xyz2, xyz3, xyz4]
        dst = [t.shape[1] for t in f].index(num_point)
        convs, ts = [], []
        for src in range(5):
            if src == dst:
                continue
            convs.append(getattr(self, "conv%d%d" % (src, dst)))
            if src < dst:      # finer -> coarser: gather through the composed FPS maps
                t = index_points(f[src], _compose(*fps[src:dst]))
            elif src == dst + 1:   # adjacent coarser state: reuse the encoder's kNN
                t = upsample(f[src], knn[src])
            else:              # non-adjacent: fresh xyz kNN of the coarse state in the target state
                ratio = f[dst].shape[1] // f[src].shape[1]
                t = upsample(f[src], knn_point(self.knn, xyz[dst], xyz[src])[1], scale_ratio=ratio)
            ts.append(t)
        # acc = f[dst] + conv_0(t_0) + conv_1(t_1) + ... (same order of additions as the reference's expression):
        # the four products are one grouped launch, every unit's normalise pass adds the running sum as its residual
        acc = _unit_group(convs, ts, residuals=[f[dst]] + [None] * (len(convs) - 1), mode="chain")
        f[dst] = getattr(self, "conv%d" % dst).fused(acc, f[dst])
        return tuple(f)


def _max_over_points(t):
    """t.max(dim=1, keepdim=True)[0] for [B,N,C] (reference :846-850) on this library's kernel (ops.max_over_points:
    same values, gradient to the first maximum).  torch's single-stage form picks a multi-workgroup reduction whose
    result was wrong from the second replay of a captured HIP graph on (garbage arg-max indices -> the backward's
    scatter_ faulted; regression test test_max_over_points_under_graph_replay keeps the torch-only evidence)."""
    return ops.max_over_points(t)


class KeepHighResolutionModulePartSeg(nn.Module):
    """Part-seg encoder-decoder wiring (reference :711-858)."""

    def __init__(self, data_C, b1_C, b2_C, b3_C, b4_C, cuda=False):
        super().__init__()
        self.neighbour = 16
        self.cuda_ops = cuda   # the reference stores this as `self.cuda`, shadowing nn.Module.cuda
        self.start = Linear(3, 32, bn=False)
        self.la0 = LocalMerge(32, 64, 8, usetanh=False, residual=True)
        self.la1 = LocalMerge(64, 64, 8, usetanh=False, residual=False)
        self.la2 = LocalMerge(64, 64, 8, usetanh=False, residual=False)
        self.la3 = LocalMerge(64, 128, 8, usetanh=False, residual=True)
        self.la4 = LocalMerge(128, 256, 8, usetanh=False, residual=True)
        self.la4_up = LocalMerge(128, 128, 8, usetanh=False, residual=False)
        self.la3_up = LocalMerge(64, 64, 8, usetanh=False, residual=False)
        self.la2_up = LocalMerge(64, 64, 8, usetanh=False, residual=False)
        self.la1_up = LocalMerge(64, 64, 8, usetanh=False, residual=False)
        self.up_conv4 = Linear(256, 128, bn=False)
        self.up_conv3 = Linear(128, 64, bn=False)
        self.up_conv2 = Linear(64, 64, bn=False)
        self.up_conv1 = Linear(64, 64, bn=False)
        self.mlp = Linear(256, 256, bn=False)
        self.conv5 = Linear(64, 256, bn=False)
        self.conv6 = Linear(64, 128, bn=False)
        self.conv7 = Linear(16, 64, bn=False)
        self.conv8 = Linear(64, 256, bn=False)
        self.fuse1 = Fuse(64, 64, 64, 128, 256)
        self.fuse2 = Fuse(64, 64, 64, 128, 256)
        self.fuse3 = Fuse(64, 64, 64, 128, 256)
        self.fuse4 = Fuse(64, 64, 64, 128, 256)
        self.fuse5 = Fuse(64, 64, 64, 128, 256)
        self.lrelu = nn.LeakyReLU(negative_slope=0.2)

    def forward(self, xyz, normal, label):
        x0 = xyz.permute(0, 2, 1).contiguous()
        nrm = x0 if normal is xyz else normal.permute(0, 2, 1).contiguous()   # (the models pass xyz as `normal`)
        N = x0.shape[1]
        # encoder: four FPS halvings, one LocalMerge per state.  The sampling chain and the xyz-space
        # kNNs depend on the coordinates only: ops.GeometryChain advances them on demand, state i's searches
        # sharing a launch with state i+1's sampling (FPS start indices drawn in the reference's order)
        geo_ = ops.GeometryChain(x0, (N // 2, N // 4, N // 8, N // 16), self.la0.knn)
        # every encoder state has five readers (the next LocalMerge -- the head's mlp for e4 -- and four of the five
        # Fuse calls): one alias per reader (ops.fanout), so that each state's gradient is summed in one launch
        e0, n0, k0, d0_ = self.la0(xyz=x0, base_xyz=x0, normal=nrm, xyz_flag=True, geometry=geo_.level(0))
        e0 = ops.fanout(e0, 5)
        g1 = geo_.level(1)
        p0, x1 = g1.fps_idx, g1.xyz
        e1, n1, k1, _ = self.la1(xyz=x1, base_xyz=x0, normal=n0, feature=e0[0], FPS_idx=p0, xyz_flag=True, geometry=g1)
        e1 = ops.fanout(e1, 5)
        g2 = geo_.level(2)
        p1, x2 = g2.fps_idx, g2.xyz
        e2, n2, k2, _ = self.la2(xyz=x2, base_xyz=x1, normal=n1, feature=e1[0], FPS_idx=p1, xyz_flag=False, geometry=g2)
        e2 = ops.fanout(e2, 5)
        g3 = geo_.level(3)
        p2, x3 = g3.fps_idx, g3.xyz
        e3, n3, k3, _ = self.la3(xyz=x3, base_xyz=x2, normal=n2, feature=e2[0], FPS_idx=p2, xyz_flag=True, geometry=g3)
        e3 = ops.fanout(e3, 5)
        g4 = geo_.level(4)
        p3, x4 = g4.fps_idx, g4.xyz
        e4, n4, k4, _ = self.la4(xyz=x4, base_xyz=x3, normal=n3, feature=e3[0], FPS_idx=p3, xyz_flag=False, geometry=g4)
        e4 = ops.fanout(e4, 5)

        geo = dict(FPS_0=p0, FPS_1=p1, FPS_2=p2, FPS_3=p3, knn_0=k0, knn_1=k1, knn_2=k2, knn_3=k3, knn_4=k4,
                   xyz0=x0, xyz1=x1, xyz2=x2, xyz3=x3, xyz4=x4)
        # decoder: upsample -> Linear -> self-state LocalMerge -> cross-state Fuse, coarse to fine
        d4 = self.mlp(e4[0])
        d4 = self.fuse1(N // 16, f0=e0[1], f1=e1[1], f2=e2[1], f3=e3[1], f4=d4, **geo)[4]
        d3 = self.la4_up(xyz=x3, base_xyz=x3, normal=n3, feature=self.up_conv4(upsample(d4, k4)))[0]
        d3 = self.fuse2(N // 8, f0=e0[2], f1=e1[2], f2=e2[2], f3=d3, f4=e4[1], **geo)[3]
        d2 = self.la3_up(xyz=x2, base_xyz=x2, normal=n2, feature=self.up_conv3(upsample(d3, k3)))[0]
        d2 = self.fuse3(N // 4, f0=e0[3], f1=e1[3], f2=d2, f3=e3[2], f4=e4[2], **geo)[2]
        d1 = self.la2_up(xyz=x1, base_xyz=x1, normal=n1, feature=self.up_conv2(upsample(d2, k2)))[0]
        d1 = self.fuse4(N // 2, f0=e0[4], f1=d1, f2=e2[3], f3=e3[3], f4=e4[3], **geo)[1]
        d0 = self.la1_up(xyz=x0, base_xyz=x0, normal=n0, feature=self.up_conv1(upsample(d1, k1)))[0]
        d0 = self.fuse5(N, f0=d0, f1=e1[4], f2=e2[4], f3=e3[4], f4=e4[4], **geo)[0]

        # per-cloud rows (five global maxima | label embedding) next to every point's conv5 features; the
        # broadcast + concatenation is one op whose backward sums the per-cloud columns with this library's
        # kernel (torch's expand() backward is a multi-workgroup reduction: see ops._CatBroadcast)
        glob = torch.cat([_max_over_points(t) for t in (d0, d1, d2, d3, d4)], dim=2)
        lab = self.conv7(label.to(d0.dtype))                             # (the one-hot joins the feature stream's dtype)
        final = ops.cat_broadcast(self.conv5(d0), torch.cat((glob, lab), 2))
        return x0, final


class PointNetFeaturePropagation(nn.Module):
    """3-NN inverse-distance interpolation + Linear (reference :860-912)."""

    def __init__(self, in_channel, mlp, act=False):
        super().__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last_channel = in_channel
        for out_channel in mlp:
            self.mlp_convs.append(nn.Conv1d(last_channel, out_channel, 1))
            self.mlp_bns.append(nn.BatchNorm1d(out_channel))
            last_channel = out_channel
        self.act = act
        self.conv = Linear(in_channel, out_channel, bn=False, act=self.act)

    def forward(self, xyz1, xyz2, points1, points2):
        return self.conv(three_interpolate(xyz1, xyz2, points2))
