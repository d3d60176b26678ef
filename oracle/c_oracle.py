"""ctypes wrapper over oracle/libmpa_oracle.so (the C restatement in mpa_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of mpa_oracle.c.  Importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the product package.
All functions take and return numpy arrays (float32 / int64, C-contiguous).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmpa_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    src = os.path.join(_HERE, "mpa_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(_i64p)


def square_distance(src, dst):
    """reference square_distance(src [B,S,C], dst [B,N,C]) -> [B,S,N]"""
    src, ps = _f(src)
    dst, pd = _f(dst)
    B, S, C = src.shape
    N = dst.shape[1]
    out = np.empty((B, S, N), np.float32)
    lib().orc_square_distance(ps, pd, B, S, N, C, out.ctypes.data_as(_f32p))
    return out


def knn_point(nsample, xyz, new_xyz):
    """reference knn_point(nsample, xyz=base [B,N,C], new_xyz=query [B,S,C]) -> (dist, idx)"""
    base, pb = _f(xyz)
    query, pq = _f(new_xyz)
    B, N, C = base.shape
    S = query.shape[1]
    dist = np.empty((B, S, nsample), np.float32)
    idx = np.empty((B, S, nsample), np.int64)
    lib().orc_knn(pb, pq, B, N, S, C, nsample, dist.ctypes.data_as(_f32p), idx.ctypes.data_as(_i64p))
    return dist, idx


def farthest_point_sample(xyz, npoint, start_idx):
    xyz, px = _f(xyz)
    start, pst = _i(start_idx)
    B, N, C = xyz.shape
    out = np.empty((B, npoint), np.int64)
    lib().orc_fps(px, B, N, C, npoint, pst, out.ctypes.data_as(_i64p))
    return out


def query_ball_point(radius, nsample, xyz, new_xyz):
    base, pb = _f(xyz)
    query, pq = _f(new_xyz)
    B, N, C = base.shape
    S = query.shape[1]
    out = np.empty((B, S, nsample), np.int64)
    r2 = np.float32(radius ** 2)
    lib().orc_ball_query(pb, pq, B, N, S, C, ctypes.c_float(float(r2)), nsample, out.ctypes.data_as(_i64p))
    return out


def three_nn(xyz1, xyz2):
    """3 nearest of xyz2 (base [B,S,C]) for every xyz1 (query [B,N,C]) -> (dist, idx) [B,N,3]"""
    q, pq = _f(xyz1)
    b, pb = _f(xyz2)
    B, Nq, C = q.shape
    Nb = b.shape[1]
    dist = np.empty((B, Nq, 3), np.float32)
    idx = np.empty((B, Nq, 3), np.int64)
    lib().orc_three_nn(pq, pb, B, Nq, Nb, C, dist.ctypes.data_as(_f32p), idx.ctypes.data_as(_i64p))
    return dist, idx
