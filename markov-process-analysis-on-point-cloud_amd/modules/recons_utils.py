"""Triangle reconstruction helpers of the RepSurf front-end -- mirror of the reference's
modules/recons_utils.py (cal_normal :27-57, cal_center :82-90, cal_const :108-124, check_nan :127-149,
check_nan_umb :152-176, knn_recons :19-25).  Plain tensor code on whatever device the inputs live;
UmbrellaSurfaceConstructor's hot path is the fused kernel behind ops.umbrella_features."""
import torch

from ..ops import index_points, query_knn_point


def knn_recons(k, center, context, cuda=False):
    return index_points(context, query_knn_point(k, context, center))


def cal_normal(group_xyz, random_inv=False, is_group=False):
    """Unit normals of triangles [...,3 (vertices),3].  Sign: x component positive -- per triangle, or
    (is_group) all triangles of a point by the FIRST triangle's x; optional per-cloud random flip drawn
    from the CPU generator as in the reference."""
    nor = torch.cross(group_xyz[..., 1, :] - group_xyz[..., 0, :], group_xyz[..., 2, :] - group_xyz[..., 0, :], dim=-1)
    unit = nor / torch.norm(nor, dim=-1, keepdim=True)
    if is_group:
        sign = (unit[..., 0:1, 0] > 0).float() * 2. - 1.
    else:
        sign = (unit[..., 0] > 0).float() * 2. - 1.
    unit = unit * sign.unsqueeze(-1)
    if random_inv:
        rnd = (torch.randint(0, 2, (group_xyz.size(0), 1, 1)).float() * 2. - 1.).to(unit.device)
        unit = unit * (rnd.unsqueeze(-1) if is_group else rnd)
    return unit


def cal_center(group_xyz):
    return group_xyz.mean(dim=-2)


def cal_const(normal, center, is_normalize=True):
    const = (normal * center).sum(-1, keepdim=True)
    return const / (3.0 ** 0.5) if is_normalize else const


def _replace_nan(first_of, bad, tensors):
    out = []
    for t in tensors:
        rep = first_of(t)
        out.append(torch.where(bad.unsqueeze(-1), rep, t))
    return out


def check_nan(normal, center, pos=None):
    """[B,N,*]: points whose normal is NaN take the cloud's first valid point's values."""
    B, N, _ = normal.shape
    bad = torch.isnan(normal).any(-1)
    first = torch.argmax((~bad).int(), dim=-1)
    ar = torch.arange(B, device=normal.device)
    ts = [normal, center] + ([pos] if pos is not None else [])
    res = _replace_nan(lambda t: t[ar, first].unsqueeze(1).expand(-1, N, -1), bad, ts)
    return tuple(res)


def check_nan_umb(normal, center, pos=None):
    """[B,N,G,*]: triangles whose normal is NaN take the point's first valid triangle's values."""
    B, N, G, _ = normal.shape
    bad = torch.isnan(normal).any(-1)
    pick = torch.argmax((~bad).int(), dim=-1).view(B, N, 1, 1)
    ts = [normal, center] + ([pos] if pos is not None else [])
    res = _replace_nan(lambda t: torch.gather(t, 2, pick.expand(-1, -1, 1, t.shape[-1])).expand(-1, -1, G, -1), bad, ts)
    return tuple(res)
