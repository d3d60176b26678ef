"""Device ops of the Markov set-abstraction path: thin torch wrappers over the C ABI of
libmpa_hip.so.  torch owns memory, streams and autograd plumbing; all arithmetic of the path
runs in the hand-written gfx950 kernels.  There is no CPU path here: every op raises on
non-CUDA tensors (the CPU oracle under oracle/ is test infrastructure and is never imported).

Function names and argument order follow the reference's modules/pointnet2_utils.py
(== modules/repsurface_utils.py); legacy keyword arguments used by the reference's callers
(`cuda=`, `is_group=`) are accepted and ignored.
"""
import ctypes
import os

import torch

from ._lib import lib, check, BnUnit, GemmProblem, GemmTnProblem, GemmTnProblemBf16, GeoRider

_vp = ctypes.c_void_p


def _p(t):
    return _vp(t.data_ptr()) if t is not None else _vp(0)


def _stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


# Optional per-kernel timing (bench.py's roofline leg): HIP events are recorded on the stream
# the kernel is launched on (torch's current stream) around the named entry points only.
_TIMERS = None


def enable_kernel_timing(names):
    global _TIMERS
    _TIMERS = {n: [] for n in names}


def disable_kernel_timing():
    global _TIMERS
    _TIMERS = None


def kernel_timing_results():
    """-> {name: {"launches", "ms", "algo_bytes", "algo_flops", "algo_units"}} totals (synchronises).  algo_units: the
    kernel's own unit of work where it has one (kNN: query-base distance evaluations)."""
    torch.cuda.synchronize()
    out = {}
    for name, recs in (_TIMERS or {}).items():
        out[name] = {"launches": len(recs), "ms": sum(r[0].elapsed_time(r[1]) for r in recs),
                     "algo_bytes": sum(r[2] for r in recs), "algo_flops": sum(r[3] for r in recs),
                     "algo_units": sum(r[4] for r in recs)}
    return out


def _launch(name, *args, algo_bytes=0, algo_flops=0, algo_units=0, tag=None, variant=None, timer=None, pre=None):
    # variant: which device kernel the entry point will pick ("mpa_gemm_f32/shortk"), so that the
    # roofline leg can price HBM-bound and MFMA-bound launches of one entry point separately.
    # timer: account the launch under this name instead of its own; pre = (name, args): a helper launch that belongs
    # to this one (issued first, inside the same event bracket).
    fn = getattr(lib, name)
    recs = None
    if _TIMERS is not None:
        key = timer or name
        recs = _TIMERS.get(key + "/" + variant) if variant else None
        if recs is None:
            recs = _TIMERS.get(key)
    if recs is None:
        if pre is not None:
            check(getattr(lib, pre[0])(*pre[1]), pre[0])
        check(fn(*args), name)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if pre is not None:
        check(getattr(lib, pre[0])(*pre[1]), pre[0])
    check(fn(*args), name)
    e1.record()
    recs.append((e0, e1, algo_bytes, algo_flops, algo_units))
    if _TAGS is not None:
        _TAGS.append((name, tag, e0, e1))


_TAGS = None      # development: per-launch (name, tag, events) when set to a list


# FPS start indices: None = draw from the CPU generator per call (reference behaviour); a hook
# (B, N, device) -> int64 device tensor lets a captured HIP graph replay with fresh draws.
_FPS_START_HOOK = None


def set_fps_start_hook(fn):
    global _FPS_START_HOOK
    _FPS_START_HOOK = fn


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mpa_amd ops need CUDA (ROCm) tensors; there is no CPU fallback in the product path")


def _f32(t):
    if t.dtype != torch.float32:
        raise TypeError("expected float32, got %s" % t.dtype)
    return t.contiguous()


def _f32_pair(base, query):
    """The fp32 rows the searches read, of `base` and `query` (bf16 features: their exact fp32 values).  A state searched
    in itself (the decoder's blocks) is converted once."""
    b = _f32(base.detach().float())
    return b, (b if query is base else _f32(query.detach().float()))


def _i64(t):
    return t.to(torch.int64).contiguous()


# ---- feature precision -------------------------------------------------------------------------
# Features (activations and their gradients) are stored fp32 (default: the reference's precision, the
# parity path) or bf16 (BASELINE configs 3 and 5: "features / GEMM operands bf16, fp32 accumulate;
# coordinates, distances, indices, BatchNorm statistics and optimizer state fp32").  Every op
# dispatches on the dtype of the feature tensor it is handed; FEATURE_DTYPE only decides what the ops
# that CREATE features from fp32 inputs emit (the xyz branch of LocalTrans, on raw coordinates).
FEATURE_DTYPE = torch.float32
_FEATURE_DTYPES = (torch.float32, torch.bfloat16)


def set_feature_dtype(dtype):
    """torch.float32 (reference precision) or torch.bfloat16 (bf16 feature path).  Returns the old value."""
    global FEATURE_DTYPE
    if dtype not in _FEATURE_DTYPES:
        raise TypeError("feature dtype must be float32 or bfloat16, got %s" % dtype)
    old, FEATURE_DTYPE = FEATURE_DTYPE, dtype
    return old


class feature_dtype:
    """with ops.feature_dtype(torch.bfloat16): ...   (context-manager form of set_feature_dtype)"""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        self.old = set_feature_dtype(self.dtype)
        return self

    def __exit__(self, *a):
        set_feature_dtype(self.old)
        return False


def _feat(t):
    """a feature tensor: fp32 or bf16, made contiguous"""
    if t.dtype not in _FEATURE_DTYPES:
        raise TypeError("expected a float32 or bfloat16 feature tensor, got %s" % t.dtype)
    return t.contiguous()


def _sfx(t):
    """entry-point suffix for a feature tensor's storage type"""
    return "bf16" if t.dtype == torch.bfloat16 else "f32"


# ------------------------------------------------------------------------------- sampling
def _fps_start(B, N, device, start_idx=None):
    """First index of every cloud: from the global CPU generator exactly as in the reference
    (torch.randint on CPU, then moved), or from the graph-safe feeder when one is installed."""
    if start_idx is None and _FPS_START_HOOK is not None:
        fed = _FPS_START_HOOK(B, N, device)           # runtime.FpsStartFeeder (None outside a fed pass)
        if fed is not None:
            return fed
    if start_idx is None:
        start_idx = torch.randint(0, N, (B,), dtype=torch.long)
    if not start_idx.is_cuda and (int(start_idx.min()) < 0 or int(start_idx.max()) >= N):
        raise ValueError("farthest_point_sample: start_idx out of range")
    return start_idx.to(device=device, dtype=torch.int64).contiguous()


def fps_and_knn_xyz(fps_in, npoint, k, knn_base, knn_query, start_idx=None):
    """farthest_point_sample(fps_in, npoint, return_xyz=True) and knn_point(k, knn_base, knn_query) on
    coordinates, as ONE launch (the sampling keeps one workgroup per cloud busy, the search uses the
    rest of the chip).  -> (fps_idx, fps_xyz, dist, idx)."""
    _dev(fps_in, knn_base, knn_query)
    fps_in, base, query = _f32(fps_in.detach()), _f32(knn_base.detach()), _f32(knn_query.detach())
    B, fN, C = fps_in.shape
    N, S = base.shape[1], query.shape[1]
    if C != 3 or base.shape[2] != 3 or query.shape[2] != 3:
        raise ValueError("fps_and_knn_xyz works on xyz coordinates (C == 3)")
    start = _fps_start(B, fN, fps_in.device, start_idx)
    fidx = torch.empty(B, npoint, dtype=torch.int64, device=fps_in.device)
    fxyz = torch.empty(B, npoint, 3, dtype=torch.float32, device=fps_in.device)
    dist = torch.empty(B, S, k, dtype=torch.float32, device=base.device)
    idx = torch.empty(B, S, k, dtype=torch.int64, device=base.device)
    _launch("mpa_fps_knn_xyz_f32", _p(fps_in), B, fN, npoint, _p(start), _p(fidx), _p(fxyz), _p(base), _p(query), N, S, k,
            _p(dist), _p(idx), _stream(), algo_units=npoint)          # units: serial FPS iterations of the launch
    _memo_put(base, query, k, dist, idx)
    return fidx, fxyz, dist, idx


def farthest_point_sample(xyz, npoint, cuda=False, start_idx=None, return_xyz=False):
    """reference: modules/pointnet2_utils.py:84-109.  xyz [B,N,C] -> int64 [B,npoint] (C == 3: the
    register-resident kernel of the models' path; any other C: the generic kernel).
    The first index of every cloud comes from the global CPU generator exactly as in the
    reference (torch.randint on CPU, then moved), so equal seeds give equal samples."""
    _dev(xyz)
    xyz = _f32(xyz)
    B, N, C = xyz.shape
    start = _fps_start(B, N, xyz.device, start_idx)
    out = torch.empty(B, npoint, dtype=torch.int64, device=xyz.device)
    if C != 3 or N > 12288:
        # rows of any width (the reference sums the squared differences over all C channels, :103-104), and clouds
        # beyond the register-resident kernel's 12,288 points (the same arithmetic, rows re-read from L2)
        _launch("mpa_fps_generic_f32", _p(xyz), B, N, C, npoint, _p(start), _p(out), _stream())
        return (out, index_points(xyz, out)) if return_xyz else out
    oxyz = torch.empty(B, npoint, 3, dtype=torch.float32, device=xyz.device) if return_xyz else None
    _launch("mpa_fps_f32", _p(xyz), B, N, npoint, _p(start), _p(out), _p(oxyz), _stream(), algo_units=npoint)
    return (out, oxyz) if return_xyz else out


def farthest_point_sample_ragged(clouds, npoint, start_idx=None):
    """FPS of clouds with DIFFERENT point counts in one launch: `clouds` is a list of [N_i, >=3] device
    tensors -> (idx int64 [B, npoint], padded [B, Nmax, C]); distances run over all C channels, as in the
    reference.  Every cloud is padded to the longest one
    with copies of its point 0: a copy always carries point 0's running distance and a higher index,
    so it never wins the arg-max -- row i equals farthest_point_sample(clouds[i][None], npoint) of the
    reference, called one cloud at a time in list order (dataset/ShapeNetDataLoader.py:127-133),
    including the repeated index 0 once a cloud is exhausted and the order the start indices are drawn
    from the global CPU generator (one torch.randint(0, N_i, (1,)) per cloud)."""
    _dev(*clouds)
    B = len(clouds)
    nmax = max(int(c.shape[0]) for c in clouds)
    C = int(clouds[0].shape[1])
    padded = torch.empty(B, nmax, C, dtype=torch.float32, device=clouds[0].device)
    for i, c in enumerate(clouds):
        n = int(c.shape[0])
        padded[i, :n] = c
        padded[i, n:] = c[0]
    if start_idx is None:
        start_idx = torch.cat([torch.randint(0, int(c.shape[0]), (1,), dtype=torch.long) for c in clouds])
    start_idx = torch.as_tensor(start_idx, dtype=torch.long)
    for i, c in enumerate(clouds):
        if not 0 <= int(start_idx[i]) < int(c.shape[0]):
            raise ValueError("farthest_point_sample_ragged: start_idx[%d] out of range" % i)
    # the reference hands the WHOLE [1,N,C] row set to farthest_point_sample (xyz | normals when the reader's
    # normal_channel is on), whose distance sums over all C channels (modules/pointnet2_utils.py:103-104)
    idx = farthest_point_sample(padded, npoint, start_idx=start_idx)
    return idx, padded


def sample(nsample, feature, cuda=False):
    """Legacy helper the reference's train loop calls but never defines
    (tool/train_cls_scanobjectnn.py:244): FPS-downsample a channel-first batch
    [B,C,N] -> [B,C,nsample] using the first three channels as coordinates."""
    pts = feature.transpose(1, 2).contiguous()
    idx = farthest_point_sample(pts[:, :, :3].contiguous(), nsample)
    return index_points(pts, idx).transpose(1, 2).contiguous()


# ------------------------------------------------------------------------------- distances
def square_distance(src, dst):
    """reference: modules/pointnet2_utils.py:190-209.  [B,S,C],[B,N,C] -> [B,S,N] (bit-exact)."""
    _dev(src, dst)
    src, dst = _f32(src), _f32(dst)
    B, S, C = src.shape
    N = dst.shape[1]
    out = torch.empty(B, S, N, dtype=torch.float32, device=src.device)
    _launch("mpa_square_distance_f32", _p(src), _p(dst), B, S, N, C, _p(out), _stream())
    return out


# Coordinate searches repeat inside one forward pass: the part-seg decoder's la1_up searches (x0, x0) again, which the
# encoder's la0 has already done (101-137 us at B=32, N=2048).  The last few results are remembered, keyed by the
# coordinate TENSORS themselves (the entries hold them, so their addresses cannot be recycled) and their versions (an
# in-place edit misses) and used only on the stream that produced them (stream order makes them ready); results carry no
# gradient and callers never write into them.
_XYZ_KNN_MEMO = []
_XYZ_KNN_MEMO_SIZE = 32         # a part-seg pass issues 16 distinct coordinate searches before it repeats one


def _memo_key(base, query, k):
    return (base.data_ptr(), query.data_ptr(), base._version, query._version, tuple(base.shape), tuple(query.shape), int(k))


def _memo_get(base, query, k):
    key = _memo_key(base, query, k)
    cur = torch.cuda.current_stream(base.device)
    for e in _XYZ_KNN_MEMO:
        if e[0] == key and e[5] == cur:
            return e[3], e[4]
    return None


def _memo_put(base, query, k, dist, idx):
    _XYZ_KNN_MEMO.append((_memo_key(base, query, k), base, query, dist, idx, torch.cuda.current_stream(base.device)))
    del _XYZ_KNN_MEMO[:-_XYZ_KNN_MEMO_SIZE]


def clear_knn_memo():
    """Forget the remembered coordinate searches.  The memo is scoped to ONE forward pass (GeometryChain and
    GraphedTrainStep clear it when a pass starts): its key is (address, torch version counter, shape), and a
    coordinate buffer rewritten through a raw-pointer launch or `.data` keeps both -- callers must not rely on the
    memo for persistent buffers that change between passes."""
    del _XYZ_KNN_MEMO[:]


def knn_point(nsample, xyz, new_xyz):
    """reference: modules/pointnet2_utils.py:211-222.  Returns (dist [B,S,k], idx int64 [B,S,k]),
    ascending.  Indices are not differentiable; dist carries no gradient (the models never use it)."""
    _dev(xyz, new_xyz)
    # bf16 features: the search runs on their exact fp32 values (distances and indices are always fp32 /
    # int64 work: the result is the reference's arithmetic applied to the rounded features)
    base, query = _f32_pair(xyz, new_xyz)
    B, N, C = base.shape
    S = query.shape[1]
    if C == 3:
        hit = _memo_get(base, query, nsample)
        if hit is not None:
            return hit
    dist = torch.empty(B, S, nsample, dtype=torch.float32, device=base.device)
    idx = torch.empty(B, S, nsample, dtype=torch.int64, device=base.device)
    if C in (32, 64) and nsample <= 8 and (S + 31) // 32 * B >= 1024 and base.data_ptr() % 16 == 0 and query.data_ptr() % 16 == 0:
        # the searches of the fine states (>= 1024 workgroups of 32 queries): base-row norms once per search (one small
        # launch) instead of once per workgroup, pass and tile, which also frees the registers for two query groups per
        # workgroup (every staged base tile feeds two MFMA chains): 538 -> 415 us at B=32, S=N=2048, C=64.  Smaller
        # searches gain less than the extra launch costs and keep the self-contained entry point.
        norms = torch.empty(B, (N + 31) // 32 * 32, dtype=torch.float32, device=base.device)
        _launch("mpa_knn_norms_f32", _p(base), _p(norms), _p(query), B, N, S, C, nsample, _p(dist), _p(idx), _stream(),
                algo_bytes=B * (4 * C * (S + N) + 12 * S * nsample), algo_flops=2 * B * S * N * C, algo_units=B * S * N,
                timer="mpa_knn_f32", pre=("mpa_row_norms_f32", (_p(base), B, N, C, _p(norms), _stream())))
        return dist, idx
    _launch("mpa_knn_f32", _p(base), _p(query), B, N, S, C, nsample, _p(dist), _p(idx), _stream(),
            algo_bytes=B * (4 * C * (S + N) + 12 * S * nsample), algo_flops=2 * B * S * N * C, algo_units=B * S * N)
    if C == 3:
        _memo_put(base, query, nsample, dist, idx)
    return dist, idx


def query_knn_point(k, xyz, new_xyz, cuda=False):
    """Legacy name used by modules/repsurface_utils.py:111 and recons_utils.py (defined nowhere
    in the reference): indices only."""
    return knn_point(k, xyz, new_xyz)[1]


def query_ball_point(radius, nsample, xyz, new_xyz, cuda=False):
    """reference: modules/pointnet2_utils.py:112-134."""
    _dev(xyz, new_xyz)
    base, query = _f32(xyz.detach()), _f32(new_xyz.detach())
    B, N, C = base.shape
    S = query.shape[1]
    idx = torch.empty(B, S, nsample, dtype=torch.int64, device=base.device)
    r2 = float(torch.tensor(radius ** 2, dtype=torch.float32))   # the compare happens in fp32
    _launch("mpa_ball_query_f32", _p(base), _p(query), B, N, S, C, r2, nsample, _p(idx), _stream())
    return idx


def three_nn(xyz1, xyz2):
    """3 nearest xyz2 rows for every xyz1 row, as PointNetFeaturePropagation sorts them
    (modules/pointnet2_utils.py:899-901) -> (dist [B,N,3], idx [B,N,3])."""
    return knn_point(3, xyz2, xyz1)


# ------------------------------------------------------------------------------- geometry pass
class _GeoLevel:
    """One point-set state of a forward pass: its coordinates, the FPS indices that selected it from the state before,
    and its coordinate search (dist, idx) in that state.  Levels of a GeometryChain compute the search -- and the NEXT
    state's sampling -- on demand (xyz_search / search); levels of a geometry_pass carry them precomputed."""
    __slots__ = ("xyz", "fps_idx", "dist", "idx", "event", "chain", "i")

    def __init__(self):
        self.xyz = self.fps_idx = self.dist = self.idx = self.event = self.chain = None
        self.i = 0

    def xyz_search(self):
        """(dist, idx) of knn_point(k, state before, this state) [level 0: itself in itself]."""
        if self.idx is None:
            self.chain._advance(self, None, None, None)
        return self.dist, self.idx

    def search(self, k, feature, query):
        """((dist, idx) as xyz_search, idx_feature = knn_point(k, feature, query)[1]); a chain level issues the next
        state's sampling, this coordinate search (if still due) and the feature search as one launch."""
        if self.chain is None:
            return (self.dist, self.idx), knn_point(k, feature, query)[1]
        idx_f = self.chain._advance(self, k, feature, query)
        return (self.dist, self.idx), idx_f


class GeometryPass:
    """All sampling levels and xyz-space neighbourhoods of one forward, produced on a side stream.
    level(i) makes the current stream wait for level i's event and returns its tensors."""

    def __init__(self, levels, side):
        self.levels, self.side = levels, side

    def level(self, i):
        g = self.levels[i]
        if g.event is not None:
            torch.cuda.current_stream().wait_event(g.event)
            g.event = None
        return g


# The next state's FPS rides in the launch of this state's searches (mpa_fps_knn_feat_f32).  Off: the chain issues the
# same work as separate launches (bench.py times kernels that way: one entry point per event bracket).
FUSE_FPS_FEATURE_SEARCH = os.environ.get("MPA_NO_FPS_FEATURE_FUSION") is None


def set_fps_feature_fusion(on):
    global FUSE_FPS_FEATURE_SEARCH
    old, FUSE_FPS_FEATURE_SEARCH = FUSE_FPS_FEATURE_SEARCH, bool(on)
    return old


def geo_level(fps_src, npoint, start, fps_idx_out, fps_xyz_out, k_xyz, xyz_base, xyz_query, extra, k_feat, feat_base,
              feat_query):
    """One state's launch in the cross-step pipeline (GeometryPipeline): farthest_point_sample(fps_src, npoint) INTO the
    given buffers (another batch's coordinates), knn_point(k_xyz, xyz_base, xyz_query), an optional second coordinate
    search extra = (base, query, k, dist_out, idx_out) into given buffers, and knn_point(k_feat, feat_base, feat_query) --
    as one launch where the shapes allow (mpa_geo_level_f32 / mpa_coarse_level_f32), else as the separate entry points.
    -> ((dist, idx), (dist_f, idx_f)); every result equals the separate calls' bit for bit."""
    _dev(fps_src, xyz_base, feat_base)
    fb, fq = _f32_pair(feat_base, feat_query)
    B, N, C = fb.shape
    S = fq.shape[1]
    fin = _f32(fps_src.detach())
    fN = fin.shape[1]
    dev = fb.device
    xb, xq = _f32(xyz_base.detach()), _f32(xyz_query.detach())
    xN, xS = xb.shape[1], xq.shape[1]
    coarse = extra is None and _coarse_ok(fN, xN, N, C, k_xyz, k_feat, fb, fq) and fN <= 256
    regular = (FUSE_FPS_FEATURE_SEARCH and C in (64, 128) and k_feat <= 8 and k_xyz <= 8 and 128 < fN <= 4096
               and fb.data_ptr() % 16 == 0 and fq.data_ptr() % 16 == 0 and (extra is None or extra[2] <= 8))
    if not (coarse or regular):
        idx, x = farthest_point_sample(fin, npoint, start_idx=start, return_xyz=True)
        fps_idx_out.copy_(idx)
        fps_xyz_out.copy_(x)
        if extra is not None:
            d, i = knn_point(extra[2], extra[0], extra[1])
            extra[3].copy_(d)
            extra[4].copy_(i)
        return knn_point(k_xyz, xb, xq), knn_point(k_feat, fb, fq)
    dx = torch.empty(B, xS, k_xyz, dtype=torch.float32, device=dev)
    ix = torch.empty(B, xS, k_xyz, dtype=torch.int64, device=dev)
    df = torch.empty(B, S, k_feat, dtype=torch.float32, device=dev)
    jf = torch.empty(B, S, k_feat, dtype=torch.int64, device=dev)
    if coarse:
        _launch("mpa_coarse_level_f32", _p(fin), B, fN, int(npoint), _p(start), _p(fps_idx_out), _p(fps_xyz_out), _p(xb), _p(xq),
                xN, xS, int(k_xyz), _p(dx), _p(ix), _p(fb), _p(fq), N, S, C, k_feat, _p(df), _p(jf), _stream(),
                algo_units=int(npoint))
    else:
        norms = None
        if C == 64 and (S + 31) // 32 * B >= 1024 and not PIPELINE_ONE_QUERY_GROUP:
            norms = torch.empty(B, (N + 31) // 32 * 32, dtype=torch.float32, device=dev)
            _launch("mpa_row_norms_f32", _p(fb), B, N, C, _p(norms), _stream())
        zb = zq = zd = zi = None
        zN = zS = zK = 0
        if extra is not None:
            zb, zq = _f32(extra[0].detach()), _f32(extra[1].detach())
            zN, zS, zK, zd, zi = zb.shape[1], zq.shape[1], int(extra[2]), extra[3], extra[4]
        _launch("mpa_geo_level_f32", _p(fin), B, fN, int(npoint), _p(start), _p(fps_idx_out), _p(fps_xyz_out), _p(xb), _p(xq),
                xN, xS, int(k_xyz), _p(dx), _p(ix), _p(zb), _p(zq), zN, zS, zK, _p(zd), _p(zi), _p(fb), _p(norms), _p(fq), N,
                S, C, k_feat, _p(df), _p(jf), _stream(), algo_units=int(npoint))
    _memo_put(xb, xq, k_xyz, dx, ix)
    return (dx, ix), (df, jf)


# development switch: one 32-query group per workgroup for the state-1 feature search that carries the next batch's
# level-1 sampling (more, shorter workgroups around the 64 sampling ones)
PIPELINE_ONE_QUERY_GROUP = os.environ.get("MPA_PIPELINE_QG1") == "1"
PIPELINE_KNN0_LEVEL = int(os.environ.get("MPA_PIPELINE_KNN0", "1"))     # which state's launch carries the next batch's state-0 search


def _coarse_ok(fN, xN, N, C, k_xyz, k_feat, fb, fq):
    """shapes of mpa_coarse_level_f32 (a coarse state's sampling + searches as one launch of small workgroups)"""
    npad = (N + 31) // 32 * 32
    lds = 4 * ((npad + 32) * (C + 4) + npad + 32 * (npad + 1))      # base + query rows, norms, 32 x N distances
    return (FUSE_FPS_FEATURE_SEARCH and C in (32, 64, 128, 256) and N <= 256 and lds <= 160 * 1024
            and (fN is None or fN <= 256) and (xN is None or xN <= 256) and k_feat <= min(8, N)
            and (k_xyz is None or k_xyz <= min(8, xN)) and fb.data_ptr() % 16 == 0 and fq.data_ptr() % 16 == 0)


def _coarse_level(fps_in, npoint, start_idx, k_xyz, xyz_base, xyz_query, k_feat, fb, fq):
    B, N, C = fb.shape
    S = fq.shape[1]
    dev = fb.device
    fin = start = fidx = fxyz = None
    fN = 0
    if fps_in is not None:
        fin = _f32(fps_in.detach())
        fN = fin.shape[1]
        start = _fps_start(B, fN, dev, start_idx)
        fidx = torch.empty(B, npoint, dtype=torch.int64, device=dev)
        fxyz = torch.empty(B, npoint, 3, dtype=torch.float32, device=dev)
    xb = xq = dx = ix = None
    xN = xS = 0
    if xyz_base is not None:
        xb, xq = _f32(xyz_base.detach()), _f32(xyz_query.detach())
        xN, xS = xb.shape[1], xq.shape[1]
        dx = torch.empty(B, xS, k_xyz, dtype=torch.float32, device=dev)
        ix = torch.empty(B, xS, k_xyz, dtype=torch.int64, device=dev)
    df = torch.empty(B, S, k_feat, dtype=torch.float32, device=dev)
    jf = torch.empty(B, S, k_feat, dtype=torch.int64, device=dev)
    _launch("mpa_coarse_level_f32", _p(fin), B, fN, int(npoint or 0), _p(start), _p(fidx), _p(fxyz), _p(xb), _p(xq), xN, xS,
            int(k_xyz or 0), _p(dx), _p(ix), _p(fb), _p(fq), N, S, C, k_feat, _p(df), _p(jf), _stream(),
            algo_units=int(npoint or 0))
    if xb is not None:
        _memo_put(xb, xq, k_xyz, dx, ix)
    return fidx, fxyz, (None if xb is None else (dx, ix)), (df, jf)


def knn_xyz_and_feature(k_xyz, xyz_base, xyz_query, k_feat, feat_base, feat_query):
    """knn_point(k_xyz, xyz_base, xyz_query) and knn_point(k_feat, feat_base, feat_query) as ONE launch where the
    shapes allow (the fused kernel of fps_knn_fused without sampling workgroups), else as two.
    -> ((dist, idx), (dist_f, idx_f)), bit-identical to the separate calls."""
    _dev(xyz_base, feat_base, feat_query)
    fb, fq = _f32_pair(feat_base, feat_query)
    B, N, C = fb.shape
    S = fq.shape[1]
    if _coarse_ok(None, xyz_base.shape[1], N, C, k_xyz, k_feat, fb, fq):
        _, _, rx, rf = _coarse_level(None, None, None, k_xyz, xyz_base, xyz_query, k_feat, fb, fq)
        return rx, rf
    ok = (FUSE_FPS_FEATURE_SEARCH and C in (64, 128) and k_feat <= 8 and k_xyz <= 8 and fb.data_ptr() % 16 == 0
          and fq.data_ptr() % 16 == 0)
    if not ok:
        return knn_point(k_xyz, xyz_base, xyz_query), knn_point(k_feat, fb, fq)
    dev = fb.device
    xb, xq = _f32(xyz_base.detach()), _f32(xyz_query.detach())
    xN, xS = xb.shape[1], xq.shape[1]
    dx = torch.empty(B, xS, k_xyz, dtype=torch.float32, device=dev)
    ix = torch.empty(B, xS, k_xyz, dtype=torch.int64, device=dev)
    df = torch.empty(B, S, k_feat, dtype=torch.float32, device=dev)
    jf = torch.empty(B, S, k_feat, dtype=torch.int64, device=dev)
    norms = None
    if C == 64 and (S + 31) // 32 * B >= 1024:
        norms = torch.empty(B, (N + 31) // 32 * 32, dtype=torch.float32, device=dev)
        _launch("mpa_row_norms_f32", _p(fb), B, N, C, _p(norms), _stream())
    _launch("mpa_fps_knn_feat_f32", None, B, 0, 0, None, None, None, _p(xb), _p(xq), xN, xS, int(k_xyz), _p(dx), _p(ix),
            _p(fb), _p(norms), _p(fq), N, S, C, k_feat, _p(df), _p(jf), _stream(),
            algo_bytes=B * (4 * C * (S + N) + 12 * S * k_feat), algo_flops=2 * B * S * N * C, algo_units=B * S * N,
            timer="mpa_knn_f32")
    _memo_put(xb, xq, k_xyz, dx, ix)
    return (dx, ix), (df, jf)


def fps_knn_fused(fps_in, npoint, k_xyz, xyz_base, xyz_query, k_feat, feat_base, feat_query, start_idx=None):
    """farthest_point_sample(fps_in, npoint, return_xyz=True), knn_point(k_xyz, xyz_base, xyz_query) (skipped when
    xyz_base is None) and knn_point(k_feat, feat_base, feat_query) as ONE launch where the shapes allow (feature rows of
    64 / 128 floats, k <= 8, 129..2048 points to sample from), otherwise as the separate launches.
    -> (fps_idx, fps_xyz, (dist, idx) | None, (dist_f, idx_f)); every result equals the separate calls' bit for bit."""
    _dev(fps_in, feat_base, feat_query)
    fb, fq = _f32_pair(feat_base, feat_query)
    B, N, C = fb.shape
    S = fq.shape[1]
    fN = fps_in.shape[1]
    if fps_in.shape[2] == 3 and _coarse_ok(fN, None if xyz_base is None else xyz_base.shape[1], N, C,
                                           None if xyz_base is None else k_xyz, k_feat, fb, fq):
        return _coarse_level(fps_in, npoint, start_idx, k_xyz, xyz_base, xyz_query, k_feat, fb, fq)
    ok = (FUSE_FPS_FEATURE_SEARCH and C in (64, 128) and k_feat <= 8 and 128 < fN <= 4096 and fps_in.shape[2] == 3
          and fb.data_ptr() % 16 == 0 and fq.data_ptr() % 16 == 0 and (xyz_base is None or k_xyz <= 8))
    if not ok:
        if xyz_base is not None:
            fidx, fxyz, dx, ix = fps_and_knn_xyz(fps_in, npoint, k_xyz, xyz_base, xyz_query, start_idx=start_idx)
            res_x = (dx, ix)
        else:
            fidx, fxyz = farthest_point_sample(fps_in, npoint, start_idx=start_idx, return_xyz=True)
            res_x = None
        return fidx, fxyz, res_x, knn_point(k_feat, fb, fq)
    fin = _f32(fps_in.detach())
    start = _fps_start(B, fN, fin.device, start_idx)
    dev = fin.device
    fidx = torch.empty(B, npoint, dtype=torch.int64, device=dev)
    fxyz = torch.empty(B, npoint, 3, dtype=torch.float32, device=dev)
    xb = xq = dx = ix = None
    xN = xS = 0
    if xyz_base is not None:
        xb, xq = _f32(xyz_base.detach()), _f32(xyz_query.detach())
        xN, xS = xb.shape[1], xq.shape[1]
        dx = torch.empty(B, xS, k_xyz, dtype=torch.float32, device=dev)
        ix = torch.empty(B, xS, k_xyz, dtype=torch.int64, device=dev)
    df = torch.empty(B, S, k_feat, dtype=torch.float32, device=dev)
    jf = torch.empty(B, S, k_feat, dtype=torch.int64, device=dev)
    norms = None
    if C == 64 and (S + 31) // 32 * B >= 1024:                   # the fine states: two query groups per workgroup
        norms = torch.empty(B, (N + 31) // 32 * 32, dtype=torch.float32, device=dev)
        _launch("mpa_row_norms_f32", _p(fb), B, N, C, _p(norms), _stream())
    _launch("mpa_fps_knn_feat_f32", _p(fin), B, fN, npoint, _p(start), _p(fidx), _p(fxyz), _p(xb), _p(xq), xN, xS,
            int(k_xyz or 0), _p(dx), _p(ix), _p(fb), _p(norms), _p(fq), N, S, C, k_feat, _p(df), _p(jf), _stream(),
            algo_units=npoint)
    if xb is not None:
        _memo_put(xb, xq, k_xyz, dx, ix)
    return fidx, fxyz, (None if xb is None else (dx, ix)), (df, jf)


class GeometryChain:
    """The sampling chain xyz -> npoints[0] -> npoints[1] ... of one forward pass, advanced on demand: level i's
    coordinate search and state i+1's sampling are issued when la_i asks for its neighbourhoods, in the same launch as
    la_i's feature-space search (level 0, which has no features: FPS + coordinate search).  FPS start indices are drawn
    (or fed) in the reference's order (state 1, 2, ...)."""

    def __init__(self, xyz, npoints, k):
        _dev(xyz)
        clear_knn_memo()                 # a new forward pass: searches remembered from an earlier one never match
        self.npoints, self.k = tuple(npoints), k
        g = _GeoLevel()
        g.xyz, g.chain, g.i = xyz, self, 0
        self.levels = [g] + [None] * len(self.npoints)
        self.pipe = None
        pf = _PREFETCH
        if pf is not None and pf.enabled:
            if pf.spec is None:
                pf.spec = (tuple(xyz.shape), self.npoints, k)          # discovery pass: the chain runs as usual
            elif pf.spec == (tuple(xyz.shape), self.npoints, k) and pf.ready:
                pf.attach(self)                                        # every level comes from the previous step's riders

    def level(self, i):
        g = self.levels[i]
        if g is None:
            raise RuntimeError("GeometryChain: state %d exists once state %d has been searched (LocalMerge order)" % (i, i - 1))
        return g

    def _advance(self, g, k_feat, feature, query):
        """Issue whatever of (next state's FPS, this state's coordinate search, feature search) is still due."""
        i, L = g.i, len(self.npoints)
        base = self.levels[i - 1].xyz if i > 0 else g.xyz
        if self.pipe is not None and i >= 1 and feature is not None and g.idx is None:
            with torch.no_grad():
                (g.dist, g.idx), (_, idx_f) = self.pipe.level_launch(self, i, base, g.xyz, k_feat, feature, query)
            return idx_f
        need_xyz = g.idx is None
        need_fps = i < L and self.levels[i + 1] is None
        idx_f = None
        with torch.no_grad():
            if need_fps and feature is not None:
                fidx, fxyz, rx, (_, idx_f) = fps_knn_fused(g.xyz, self.npoints[i], self.k, base if need_xyz else None,
                                                           g.xyz if need_xyz else None, k_feat, feature, query)
                if rx is not None:
                    g.dist, g.idx = rx
            elif need_fps:
                if need_xyz:
                    fidx, fxyz, g.dist, g.idx = fps_and_knn_xyz(g.xyz, self.npoints[i], self.k, base, g.xyz)
                else:
                    fidx, fxyz = farthest_point_sample(g.xyz, self.npoints[i], return_xyz=True)
            elif need_xyz and feature is not None:
                (g.dist, g.idx), (_, idx_f) = knn_xyz_and_feature(self.k, base, g.xyz, k_feat, feature, query)
            else:
                if need_xyz:
                    g.dist, g.idx = knn_point(self.k, base, g.xyz)
                if feature is not None:
                    idx_f = knn_point(k_feat, feature, query)[1]
            if need_fps:
                n = _GeoLevel()
                n.xyz, n.fps_idx, n.chain, n.i = fxyz, fidx, self, i + 1
                self.levels[i + 1] = n
        return idx_f


# ---- cross-step geometry: the NEXT batch's sampling chain rides in this step's weight-gradient launches -------------
_PREFETCH = None


def set_geometry_prefetch(pf):
    """Install (or remove, None) the GeometryPrefetch that GeometryChain consults.  Returns the previous one."""
    global _PREFETCH
    old, _PREFETCH = _PREFETCH, pf
    return old


class GeometryPrefetch:
    """Persistent geometry of ONE model configuration (clouds [B,N,3], sampling levels npoints, K), filled a step ahead.

    The sampling chain (farthest_point_sample level after level, reference modules/repsurface_utils.py:581-619) and the
    coordinate searches depend on the input coordinates only.  Under HIP-graph replay nothing overlaps the chain's
    serial iterations (0.39 ms of a 3.7 ms classification step on 64 of 256 CUs), so the chain of batch t+1 is computed
    DURING step t: two geometry riders (csrc/geo_rider.h) travel in the two grouped weight-gradient launches that
    close the backward pass --
        rider 0: level 1's sampling from the next batch's coordinates + the level-0 search (state 0 in itself),
        rider 1: levels 2..L as one chained workgroup per cloud + the level-1 search (state 1 in state 0) --
    and write into the buffers below, which step t+1's forward pass reads (GeometryChain.attach).  The buffers are
    overwritten by the LAST launches of a step, after every reader of that step (stream order), so one set suffices.
    Searches of levels >= 2 stay in the forward pass, sharing a launch with the level's feature-space search.
    FPS start indices keep the reference's draw order (one torch.randint per level, level 1 first): they are drawn a
    step early, through the same hook (runtime.FpsStartFeeder) as the in-pass chain."""

    def __init__(self):
        self.spec = None             # ((B, N, 3), npoints, k) recorded by the first GeometryChain of a discovery pass
        self.enabled = True
        # searches carried by the riders (development switch MPA_RIDER_SEARCH=0: sampling only, levels 0 / 1 are
        # searched inside the forward pass like the deeper levels)
        self.with_search = os.environ.get("MPA_RIDER_SEARCH", "1") != "0"
        self.ready = False           # buffers hold the geometry of the batch the next forward pass will see
        self.next_xyz = None         # [B, N, 3] coordinates of the next batch (static: captured graphs read it)
        self.fps_idx, self.fps_xyz, self.knn = [], [], []
        self.starts = None           # start-index views of the pass being recorded (level order)

    def allocate(self, device):
        (B, N, _), npoints, k = self.spec
        self.next_xyz = torch.zeros(B, N, 3, dtype=torch.float32, device=device)
        self.fps_idx = [torch.zeros(B, s, dtype=torch.int64, device=device) for s in npoints]
        self.fps_xyz = [torch.zeros(B, s, 3, dtype=torch.float32, device=device) for s in npoints]
        sizes = [N] + list(npoints)
        self.knn = [(torch.zeros(B, sizes[i], k, dtype=torch.float32, device=device),
                     torch.zeros(B, sizes[i], k, dtype=torch.int64, device=device)) for i in range(min(2, len(sizes)))]

    def supported(self):
        (B, N, C), npoints, k = self.spec
        return (C == 3 and N <= 4096 and 1 <= len(npoints) <= 5 and k <= 8 and
                all(b <= a for a, b in zip((N,) + tuple(npoints), npoints)))

    def _draw_starts(self, device):
        (B, N, _), npoints, _ = self.spec
        return [_fps_start(B, n, device) for n in (N,) + tuple(npoints[:-1])]

    def attach(self, chain):
        """Hand the buffered geometry to a forward pass's chain and draw (or get fed) the start indices the riders of
        THIS pass will use for the next batch."""
        (B, N, _), npoints, k = self.spec
        g0 = chain.levels[0]
        if self.with_search:
            g0.dist, g0.idx = self.knn[0]
        for i in range(len(npoints)):
            g = _GeoLevel()
            g.xyz, g.fps_idx, g.chain, g.i = self.fps_xyz[i], self.fps_idx[i], chain, i + 1
            if i == 0 and len(self.knn) > 1 and self.with_search:
                g.dist, g.idx = self.knn[1]
            chain.levels[i + 1] = g
        self.starts = self._draw_starts(g0.xyz.device)

    def riders(self):
        """(GeoRider * 2): what the step's closing launches carry (next_xyz -> buffers), None outside a recorded pass."""
        if self.starts is None:
            return None
        arr = self._riders(self.next_xyz, self.starts)
        self.starts = None
        # the carried form's two counters per rider (work queue, finished sampling workgroups): pre-zeroed words of the
        # pass's arena (inside a captured graph: cleared by the graph's first node at every replay)
        q = _zeros_acc(32, self.next_xyz.device)
        for i in range(2):
            arr[i].queue = q.data_ptr() + 64 * i
        self._keep = self._keep + (q,)
        return arr

    def _riders(self, xyz, starts):
        (B, N, _), npoints, k = self.spec
        arr = (GeoRider * 2)()
        r0, r1 = arr[0], arr[1]
        r0.src, r0.B, r0.N, r0.nlev = xyz.data_ptr(), B, N, 1
        r0.S[0], r0.start[0], r0.idx[0], r0.xyz[0] = npoints[0], starts[0].data_ptr(), self.fps_idx[0].data_ptr(), self.fps_xyz[0].data_ptr()
        if self.with_search:
            r0.base, r0.query, r0.sN, r0.sS, r0.sK = xyz.data_ptr(), xyz.data_ptr(), N, N, k
            r0.dist, r0.kidx = self.knn[0][0].data_ptr(), self.knn[0][1].data_ptr()
        r1.src, r1.B, r1.N, r1.nlev = self.fps_xyz[0].data_ptr(), B, npoints[0], len(npoints) - 1
        for j in range(1, len(npoints)):
            r1.S[j - 1], r1.start[j - 1] = npoints[j], starts[j].data_ptr()
            r1.idx[j - 1], r1.xyz[j - 1] = self.fps_idx[j].data_ptr(), self.fps_xyz[j].data_ptr()
        if self.with_search:
            r1.base, r1.query, r1.sN, r1.sS, r1.sK = xyz.data_ptr(), self.fps_xyz[0].data_ptr(), N, npoints[0], k
            r1.dist, r1.kidx = self.knn[1][0].data_ptr(), self.knn[1][1].data_ptr()
        self._keep = (xyz, starts)       # (the launches are asynchronous: keep their operands alive)
        return arr

    def compute_now(self, xyz):
        """The same two riders as launches of their own, for `xyz` [B,N,3]: the first batch of a run (and any batch
        that was not announced a step ahead)."""
        xyz = _f32(xyz.detach())
        arr = self._riders(xyz, self._draw_starts(xyz.device))
        for i in range(2):
            _launch("mpa_geo_rider_f32", ctypes.byref(arr[i]), _stream())
        self.ready = True


class GeometryPipeline:
    """Cross-step geometry carried by the SEARCH launches (round 3, second form; the first, GeometryPrefetch above, rides in
    the weight-gradient launches and is slower: MFMA workgroups on the same CU slow the sampling waves 2.3x, search
    workgroups do not -- the in-pass launches have always mixed the two).

    In a pass over batch t, the launch of state i's searches (i = 1..L) carries, as its first B workgroups, level i of the
    sampling chain of batch t+1 -- level 1 from the next batch's coordinates, level i from level i-1's result, which the
    launch of state i-1 has just produced -- and the launch of state 1 also carries batch t+1's state-0 search.  So
    sampling never sits alone on the critical path: today's first launch (level 1 + state-0 search, 193 us at batch 64 of
    1024 points: 183 us of sampling hiding a 64 us search) disappears, and every level rides beside searches of the
    CURRENT batch, which exist anyway.  Two sets of buffers: `nxt` is written during the pass, `cur` = a copy of it taken
    by the next pass's first node (one ~10 MB device copy), read by that pass's forward and backward.  Start indices are
    drawn a step ahead in the reference's order; results are bit-identical to the in-pass chain
    (tests/test_gpu_prefetch.py)."""

    def __init__(self):
        self.spec = None
        self.enabled = True
        self.ready = False           # `nxt` holds the geometry of the batch the next forward pass will see
        self.next_xyz = None
        self.cur = self.nxt = None
        self.starts = None

    class _Set:
        pass

    def supported(self):
        (B, N, C), npoints, k = self.spec
        return (C == 3 and 128 < N <= 4096 and 1 <= len(npoints) <= 8 and k <= 8 and
                all(b <= a for a, b in zip((N,) + tuple(npoints), npoints)))

    def allocate(self, device):
        (B, N, _), npoints, k = self.spec
        self.next_xyz = torch.zeros(B, N, 3, dtype=torch.float32, device=device)
        parts = []                                        # (name, index, shape, dtype)
        for i, s_ in enumerate(npoints):
            parts.append(("fps_idx", i, (B, s_), torch.int64))
            parts.append(("fps_xyz", i, (B, s_, 3), torch.float32))
        parts.append(("knn0_dist", 0, (B, N, k), torch.float32))
        parts.append(("knn0_idx", 0, (B, N, k), torch.int64))
        offs, off = [], 0
        for _, _, shape, dt in parts:
            offs.append(off)
            n = 1
            for d in shape:
                n *= d
            off += (n * (8 if dt == torch.int64 else 4) + 255) // 256 * 256
        sets = []
        for _ in range(2):
            flat = torch.zeros(off, dtype=torch.uint8, device=device)
            st = GeometryPipeline._Set()
            st.flat, st.fps_idx, st.fps_xyz = flat, [None] * len(npoints), [None] * len(npoints)
            for (name, i, shape, dt), o in zip(parts, offs):
                n = 1
                for d in shape:
                    n *= d
                v = flat[o:o + n * (8 if dt == torch.int64 else 4)].view(dt).view(*shape)
                if name == "fps_idx":
                    st.fps_idx[i] = v
                elif name == "fps_xyz":
                    st.fps_xyz[i] = v
                else:
                    setattr(st, name, v)
            sets.append(st)
        self.cur, self.nxt = sets

    def _draw_starts(self, device):
        (B, N, _), npoints, _ = self.spec
        return [_fps_start(B, n, device) for n in (N,) + tuple(npoints[:-1])]

    def attach(self, chain):
        """First thing of a forward pass: what the previous pass computed becomes this pass's geometry, the start indices
        of the levels this pass will compute for the next batch are drawn (or fed)."""
        (B, N, _), npoints, k = self.spec
        self.cur.flat.copy_(self.nxt.flat)
        g0 = chain.levels[0]
        g0.dist, g0.idx = self.cur.knn0_dist, self.cur.knn0_idx
        # a decoder that searches the input cloud in itself again (part-seg, S3DIS, completion) finds this result
        xb = _f32(g0.xyz.detach().float())
        _memo_put(xb, xb, k, g0.dist, g0.idx)
        for i in range(len(npoints)):
            g = _GeoLevel()
            g.xyz, g.fps_idx, g.chain, g.i = self.cur.fps_xyz[i], self.cur.fps_idx[i], chain, i + 1
            chain.levels[i + 1] = g
        chain.pipe = self
        self.starts = self._draw_starts(g0.xyz.device)

    def level_launch(self, chain, i, base, xyz_i, k_feat, feature, query):
        """State i's searches for the current batch + level i of the next batch's chain (+ its state-0 search at i = 1)."""
        (B, N, _), npoints, k = self.spec
        src = self.next_xyz if i == 1 else self.nxt.fps_xyz[i - 2]
        # the next batch's state-0 search needs its coordinates only, so any launch could carry it: state 1's measured
        # best (3.511 ms per cls-fp32 step against 3.535 in state 2's launch, 3.560 for the in-pass chain)
        carrier = 2 if (len(npoints) >= 2 and PIPELINE_KNN0_LEVEL == 2) else 1
        extra = (self.next_xyz, self.next_xyz, k, self.nxt.knn0_dist, self.nxt.knn0_idx) if i == carrier else None
        return geo_level(src, npoints[i - 1], self.starts[i - 1], self.nxt.fps_idx[i - 1], self.nxt.fps_xyz[i - 1], chain.k,
                         base, xyz_i, extra, k_feat, feature, query)

    def riders(self):
        return None                     # nothing rides in the weight-gradient launches in this form

    def compute_now(self, xyz):
        """The chain and the state-0 search of `xyz` [B,N,3] into `nxt` with the stand-alone entry points: the first batch
        of a run, and any batch that was not announced a step ahead."""
        (B, N, _), npoints, k = self.spec
        xyz = _f32(xyz.detach())
        starts = self._draw_starts(xyz.device)
        cur = xyz
        with torch.no_grad():
            for i, s_ in enumerate(npoints):
                idx, x = farthest_point_sample(cur, s_, start_idx=starts[i], return_xyz=True)
                self.nxt.fps_idx[i].copy_(idx)
                self.nxt.fps_xyz[i].copy_(x)
                cur = self.nxt.fps_xyz[i]
            clear_knn_memo()
            d, j = knn_point(k, xyz, xyz)
            self.nxt.knn0_dist.copy_(d)
            self.nxt.knn0_idx.copy_(j)
            clear_knn_memo()
        self.ready = True


_GEO_STREAMS = {}
GEOMETRY_SIDE_STREAM = os.environ.get("MPA_GEO_SIDE", "0") == "1"   # side stream for the FPS/kNN chain


def geometry_pass(xyz, npoints, k):
    """FPS chain xyz -> npoints[0] -> npoints[1] ... and, per level, knn_point(k, base, sampled)
    (level 0: knn_point(k, xyz, xyz)).  Depends on the coordinates only, so it is
    issued on a side stream and overlaps the feature path; FPS start indices are drawn (or fed)
    in the reference's order."""
    _dev(xyz)
    cur = torch.cuda.current_stream(xyz.device)
    use_side = GEOMETRY_SIDE_STREAM
    side = None
    if use_side:
        side = _GEO_STREAMS.get(xyz.device.index)
        if side is None:
            side = _GEO_STREAMS[xyz.device.index] = torch.cuda.Stream(xyz.device)
        side.wait_stream(cur)
    levels = []
    ctx = torch.cuda.stream(side) if use_side else _NullCtx()
    with ctx, torch.no_grad():
        # level i: kNN(base = state i-1 (state 0 for i = 0), query = state i).  It shares a launch with
        # the sampling of state i+1 (both need state i only); FPS start indices are drawn in the
        # reference's order (state 1, 2, ...).
        L = len(npoints)
        base, cur_xyz, cur_fps = xyz, xyz, None
        for i in range(L + 1):
            g = _GeoLevel()
            g.xyz, g.fps_idx = cur_xyz, cur_fps
            if i < L:
                nxt_fps, nxt_xyz, g.dist, g.idx = fps_and_knn_xyz(cur_xyz, npoints[i], k, base, cur_xyz)
            else:
                g.dist, g.idx = knn_point(k, base, cur_xyz)
            g.event = None
            if use_side:
                g.event = torch.cuda.Event()
                g.event.record(side)
                for t in (g.xyz, g.fps_idx, g.dist, g.idx):
                    if t is not None and t is not xyz:
                        t.record_stream(cur)
            levels.append(g)
            base = cur_xyz
            if i < L:
                cur_xyz, cur_fps = nxt_xyz, nxt_fps
    return GeometryPass(levels, side)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


# ------------------------------------------------------------------------------- umbrella surfaces
def umbrella_features(xyz, k=9, cloud_sign=None, return_dist=True):
    """RepSurf umbrella surface features of every point (reference modules/repsurface_utils.py:106-126 +
    the feature part of UmbrellaSurfaceConstructor.forward :350-364): xyz [B,N,3] -> [B,N,k-1,10|9]
    = triangle centre | its spherical coordinates | unit normal | (plane constant).  cloud_sign [B]
    of +-1: the per-cloud random inversion of the normals (None: none).  Coordinates only: no gradient."""
    _dev(xyz)
    with torch.no_grad():
        xyz = _f32(xyz.detach())
        B, N, C = xyz.shape
        if C != 3:
            raise ValueError("umbrella_features works on xyz coordinates (C == 3)")
        idx = knn_point(k, xyz, xyz)[1]
        sign = None if cloud_sign is None else cloud_sign.to(device=xyz.device, dtype=torch.float32).contiguous()
        out = torch.empty(B, N, k - 1, 10 if return_dist else 9, dtype=torch.float32, device=xyz.device)
        _launch("mpa_umbrella_features_f32", _p(xyz), _p(idx), B, N, k, _p(sign), int(bool(return_dist)), _p(out),
                _stream())
    return out


# ------------------------------------------------------------------------------- gathers
class _IndexPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx):
        B, N, C = points.shape
        flat = idx.reshape(B, -1)
        M = flat.shape[1]
        out = torch.empty(B, M, C, dtype=points.dtype, device=points.device)
        _launch("mpa_gather_fwd_" + _sfx(points), _p(points), _p(flat), B, N, M, C, _p(out), _stream())
        ctx.save_for_backward(flat)
        ctx.shape = (B, N, M, C)
        ctx.unique = idx.dim() == 2
        return out.view(*idx.shape, C)

    @staticmethod
    def backward(ctx, grad):
        (flat,) = ctx.saved_tensors
        B, N, M, C = ctx.shape
        grad = grad.contiguous()
        if grad.dtype == torch.bfloat16 and C % 2 == 0 and M <= N and ctx.unique:
            # [B,S] index maps of at most N rows (the FPS maps: every row listed once): scattered straight into a bf16
            # destination (no fp32 staging buffer, no cast).  Neighbour lists [B,S,K] repeat rows: their sums stay fp32
            # (a compare-and-swap that rounds to bf16 after every add would depend on the order of the adds)
            gp = torch.zeros(B, N, C, dtype=torch.bfloat16, device=grad.device)
            _launch("mpa_gather_bwd_into_bf16", _p(grad), _p(flat), B, N, M, C, _p(gp), _stream())
            return gp, None
        # rows listed several times are summed in fp32 (float atomics on whole rows), whatever the storage type
        gp = torch.zeros(B, N, C, dtype=torch.float32, device=grad.device)
        _launch("mpa_gather_bwd_" + _sfx(grad), _p(grad), _p(flat), B, N, M, C, _p(gp), _stream())
        return (gp if grad.dtype == torch.float32 else gp.to(grad.dtype)), None


def index_points(points, idx, cuda=False, is_group=False):
    """reference: modules/pointnet2_utils.py:64-81.  points [B,N,C], idx [B,S] or [B,S,K]."""
    _dev(points, idx)
    if points.dtype not in _FEATURE_DTYPES:
        # index-map composition etc. (integer payloads): plain device indexing
        B = points.shape[0]
        bidx = torch.arange(B, device=points.device).view([B] + [1] * (idx.dim() - 1)).expand_as(idx)
        return points[bidx, idx, :]
    return _IndexPoints.apply(points.contiguous(), _i64(idx))


# ------------------------------------------------------------------------------- attention
def _attn_ws_bytes(t, B, N, S, K, C):
    fn = lib.mpa_diffattn_bwd_workspace_bytes_bf16 if t.dtype == torch.bfloat16 else lib.mpa_diffattn_bwd_workspace_bytes
    return int(fn(B, N, S, K, C))


class _DiffAttn(torch.autograd.Function):
    """ctx = max_j (softmax_j((q-k_j)/sqrt(C)) - sum) * v_j with k|v = the two halves of kv."""

    @staticmethod
    def forward(ctx, q, kv, idx):
        B, S, C = q.shape
        N = kv.shape[1]
        K = idx.shape[2]
        es = q.element_size()
        out = torch.empty_like(q)
        argk = torch.empty(B, S, C, dtype=torch.uint8, device=q.device)
        _launch("mpa_diffattn_fwd_" + _sfx(q), _p(q), C, _p(kv), _vp(kv.data_ptr() + es * C), 2 * C, _p(idx), B, N, S, K, C,
                _p(out), _p(argk), _stream(), algo_bytes=B * S * (es * (2 * C + 2 * K * C) + 8 * K + C))
        ctx.save_for_backward(q, kv, idx, argk)
        return out

    @staticmethod
    def backward(ctx, grad):
        q, kv, idx, argk = ctx.saved_tensors
        B, S, C = q.shape
        N = kv.shape[1]
        K = idx.shape[2]
        es = q.element_size()
        grad = grad.contiguous()
        gq = torch.empty_like(q)
        gkv = torch.empty_like(kv)          # fully written by the kernels
        # per-slot gradients + inverted neighbour table: the atomic-free backward's scratch
        need = _attn_ws_bytes(q, B, N, S, K, C)
        ws = torch.empty(need, dtype=torch.uint8, device=q.device) if need else None
        _launch("mpa_diffattn_bwd_" + _sfx(q), _p(q), C, _p(kv), _vp(kv.data_ptr() + es * C), 2 * C, _p(idx), _p(argk),
                _p(grad), B, N, S, K, C, _p(gq), _p(gkv), _vp(gkv.data_ptr() + es * C), 2 * C, _p(ws),
                0 if ws is None else ws.numel(), _stream(),
                algo_bytes=B * S * (es * (3 * C + 2 * K * C + C) + 8 * K + C))
        return gq, gkv, None


def diffattn(q, kv, idx):
    """Difference-wise attention core of LocalTrans' feature branch
    (modules/pointnet2_utils.py:558-569).  q [B,S,C]; kv [B,N,2C] = projected keys | values;
    idx [B,S,K] -> ctx [B,S,C]."""
    _dev(q, kv, idx)
    return _DiffAttn.apply(_feat(q), _feat(kv), _i64(idx))


class _DiffAttnPair(torch.autograd.Function):
    """Two difference-wise attentions over the same base rows whose projections were computed
    stacked: qq [B,S,2C] = q1|q2, kvkv [B,N,4C] = k1|v1|k2|v2 (LocalMerge's two feature streams).
    Each stream reads its column blocks in place; backward writes both streams' gradients into one
    [B,S,2C] / [B,N,4C] pair, so the stacked projections get ONE gradient each (no additions)."""

    @staticmethod
    def forward(ctx, qq, kvkv, idx1, idx2):
        B, S, C2 = qq.shape
        C = C2 // 2
        N = kvkv.shape[1]
        K = idx1.shape[2]
        es = qq.element_size()
        outs, argks = [], []
        for s_, idx in enumerate((idx1, idx2)):
            out = torch.empty(B, S, C, dtype=qq.dtype, device=qq.device)
            argk = torch.empty(B, S, C, dtype=torch.uint8, device=qq.device)
            _launch("mpa_diffattn_fwd_" + _sfx(qq), _vp(qq.data_ptr() + es * C * s_), 2 * C,
                    _vp(kvkv.data_ptr() + 2 * es * C * s_), _vp(kvkv.data_ptr() + 2 * es * C * s_ + es * C), 4 * C,
                    _p(idx), B, N, S, K, C, _p(out), _p(argk), _stream(),
                    algo_bytes=B * S * (es * (2 * C + 2 * K * C) + 8 * K + C))
            outs.append(out)
            argks.append(argk)
        ctx.save_for_backward(qq, kvkv, idx1, idx2, *argks)
        return outs[0], outs[1]

    @staticmethod
    def backward(ctx, g1, g2):
        qq, kvkv, idx1, idx2, a1, a2 = ctx.saved_tensors
        B, S, C2 = qq.shape
        C = C2 // 2
        N = kvkv.shape[1]
        K = idx1.shape[2]
        es = qq.element_size()
        gqq = torch.empty_like(qq)
        gkvkv = torch.empty_like(kvkv)          # every column block fully written by its stream
        need = _attn_ws_bytes(qq, B, N, S, K, C)
        for s_, (idx, argk, g) in enumerate(((idx1, a1, g1), (idx2, a2, g2))):
            ws = torch.empty(need, dtype=torch.uint8, device=qq.device) if need else None
            _launch("mpa_diffattn_bwd_" + _sfx(qq), _vp(qq.data_ptr() + es * C * s_), 2 * C,
                    _vp(kvkv.data_ptr() + 2 * es * C * s_), _vp(kvkv.data_ptr() + 2 * es * C * s_ + es * C), 4 * C,
                    _p(idx), _p(argk), _p(g.contiguous()), B, N, S, K, C, _vp(gqq.data_ptr() + es * C * s_),
                    _vp(gkvkv.data_ptr() + 2 * es * C * s_), _vp(gkvkv.data_ptr() + 2 * es * C * s_ + es * C), 4 * C,
                    _p(ws), 0 if ws is None else ws.numel(), _stream(),
                    algo_bytes=B * S * (es * (3 * C + 2 * K * C + C) + 8 * K + C))
        return gqq, gkvkv, None, None


def diffattn_pair(qq, kvkv, idx1, idx2):
    """(ctx1, ctx2) of two difference-wise attentions on stacked projections (see _DiffAttnPair)."""
    _dev(qq, kvkv, idx1, idx2)
    return _DiffAttnPair.apply(_feat(qq), _feat(kvkv), _i64(idx1), _i64(idx2))


class _DiffAttnXYZ(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, center, idx, Wq, bq, Wk, bk, Wv, bv, out_dtype):
        B, N, _ = xyz.shape
        S, K = idx.shape[1], idx.shape[2]
        C = Wq.shape[0]
        out = torch.empty(B, S, C, dtype=out_dtype, device=xyz.device)
        argk = torch.empty(B, S, C, dtype=torch.uint8, device=xyz.device)
        params = [t.contiguous() for t in (Wq, bq, Wk, bk, Wv, bv)]
        _launch("mpa_diffattn_xyz_fwd_" + _sfx(out), _p(xyz), _p(center), _p(idx), *[_p(t) for t in params], B, N, S, K, C,
                                           _p(out), _p(argk), _stream())
        ctx.save_for_backward(xyz, center, idx, argk, *params)
        ctx.direct = tuple(_direct(t) for t in (Wq, bq, Wk, bk, Wv, bv))
        return out

    @staticmethod
    def backward(ctx, grad):
        xyz, center, idx, argk, Wq, bq, Wk, bk, Wv, bv = ctx.saved_tensors
        B, N, _ = xyz.shape
        S, K = idx.shape[1], idx.shape[2]
        C = Wq.shape[0]
        grad = grad.contiguous()
        # the kernel accumulates with atomics: into the cleared flat gradients when installed
        # (GradReducer direct mode), else into fresh zero tensors
        gs = [d if d is not None else torch.zeros_like(t) for d, t in zip(ctx.direct, (Wq, bq, Wk, bk, Wv, bv))]
        _launch("mpa_diffattn_xyz_bwd_" + _sfx(grad), _p(xyz), _p(center), _p(idx), _p(Wq), _p(bq), _p(Wk), _p(bk), _p(Wv),
                _p(bv), _p(argk), _p(grad), B, N, S, K, C, *[_p(g) for g in gs], _stream())
        return (None, None, None) + tuple(None if d is not None else g for d, g in zip(ctx.direct, gs)) + (None,)


def diffattn_xyz(xyz, center, idx, Wq, bq, Wk, bk, Wv, bv):
    """LocalTrans' xyz branch (modules/pointnet2_utils.py:518-544) on raw coordinates:
    xyz [B,N,3] base, center [B,S,3], idx [B,S,K]; W* [C,3], b* [C] -> ctx [B,S,C].
    Coordinates are network inputs and receive no gradient.  The context is emitted in FEATURE_DTYPE: this is
    where the feature stream of a model starts."""
    _dev(xyz, center, idx, Wq)
    return _DiffAttnXYZ.apply(_f32(xyz.detach()), _f32(center.detach()), _i64(idx), Wq, bq, Wk, bk, Wv, bv,
                              FEATURE_DTYPE)


# ------------------------------------------------------------------------------- decoder
class _UpsampleMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, knn_idx, n_fine):
        B, S, C = points.shape
        K = knn_idx.shape[2]
        out = torch.empty(B, n_fine, C, dtype=points.dtype, device=points.device)
        cnt = torch.empty(B, n_fine, dtype=torch.float32, device=points.device)
        nbytes = int(lib.mpa_upsample_workspace_bytes(B, S, K, n_fine))      # 0: atomic scatter path (fp32 only)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=points.device)
        _launch("mpa_upsample_mean_fwd_" + _sfx(points), _p(points), _p(knn_idx), B, S, K, n_fine, C, _p(out), _p(cnt),
                _p(ws) if nbytes else None, nbytes, _stream())
        ctx.save_for_backward(knn_idx, cnt)
        ctx.shape = (B, S, K, n_fine, C)
        return out

    @staticmethod
    def backward(ctx, grad):
        knn_idx, cnt = ctx.saved_tensors
        B, S, K, n_fine, C = ctx.shape
        grad = grad.contiguous()
        gp = torch.empty(B, S, C, dtype=grad.dtype, device=grad.device)
        _launch("mpa_upsample_mean_bwd_" + _sfx(grad), _p(grad), _p(knn_idx), _p(cnt), B, S, K, n_fine, C, _p(gp), _stream())
        return gp, None, None


def upsample(points, knn_idx, scale_ratio=2, dist=None):
    """reference: modules/pointnet2_utils.py:13-50 (coarse->fine transition).  points [B,S,C],
    knn_idx [B,S,K] (values < S*scale_ratio) -> [B,S*scale_ratio,C].  `dist` is unused, as in
    the reference.  Note: the divisor depends on which channel-0 values are exactly zero; like
    the reference, that dependence is not differentiated."""
    _dev(points, knn_idx)
    n_fine = points.shape[1] * scale_ratio
    return _UpsampleMean.apply(_feat(points), _i64(knn_idx), n_fine)


class _ThreeInterp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points2, idx, dist):
        B, Nb, C = points2.shape
        Nq = idx.shape[1]
        out = torch.empty(B, Nq, C, dtype=points2.dtype, device=points2.device)
        _launch("mpa_three_interp_fwd_" + _sfx(points2), _p(points2), _p(idx), _p(dist), B, Nq, Nb, C, _p(out), _stream())
        ctx.save_for_backward(idx, dist)
        ctx.shape = (B, Nq, Nb, C)
        return out

    @staticmethod
    def backward(ctx, grad):
        idx, dist = ctx.saved_tensors
        B, Nq, Nb, C = ctx.shape
        grad = grad.contiguous()
        # coarse rows are listed by several fine points: summed in fp32 (float atomics), rounded once for bf16 features
        gp = torch.zeros(B, Nb, C, dtype=torch.float32, device=grad.device)
        _launch("mpa_three_interp_bwd_" + _sfx(grad), _p(grad), _p(idx), _p(dist), B, Nq, Nb, C, _p(gp), _stream())
        return (gp if grad.dtype == torch.float32 else gp.to(grad.dtype)), None, None


def three_interpolate(xyz1, xyz2, points2):
    """Inverse-distance 3-NN interpolation of PointNetFeaturePropagation
    (modules/pointnet2_utils.py:896-906): xyz1 [B,N,3] fine, xyz2 [B,S,3] coarse, points2 [B,S,D] (fp32 or bf16
    features: the interpolated rows stay in the feature stream's storage type)."""
    _dev(xyz1, xyz2, points2)
    if xyz2.shape[1] == 1:
        return points2.repeat(1, xyz1.shape[1], 1)
    dist, idx = three_nn(xyz1, xyz2)
    return _ThreeInterp.apply(_feat(points2), idx, dist)


class _MaxOverPoints(torch.autograd.Function):
    """x [B,N,C] -> max over the points [B,1,C]; the gradient goes to the first row attaining the maximum."""

    @staticmethod
    def forward(ctx, x):
        B, N, C = x.shape
        out = torch.empty(B, 1, C, dtype=x.dtype, device=x.device)
        arg = torch.empty(B, C, dtype=torch.int32, device=x.device)
        _launch("mpa_max_points_fwd_" + _sfx(x), _p(x), B, N, C, _p(out), _p(arg), _stream())
        ctx.save_for_backward(arg)
        ctx.shape = (B, N, C)
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        B, N, C = ctx.shape
        g = g.contiguous()
        gx = torch.empty(B, N, C, dtype=g.dtype, device=g.device)
        _launch("mpa_max_points_bwd_" + _sfx(g), _p(g), _p(arg), B, N, C, _p(gx), _stream())
        return gx


def max_over_points(x):
    """x.max(dim=1, keepdim=True)[0] for a [B,N,C] state (reference modules/pointnet2_utils.py:846-850): one launch
    forward, one backward (zero fill and scatter fused).  torch's reduction mis-replays under HIP-graph capture here
     and costs 2 x 20 us staged; ties go to the lowest row."""
    _dev(x)
    return _MaxOverPoints.apply(_feat(x))


# ------------------------------------------------------------------------------- heads: log-softmax, losses, pooling
class _LogSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        M, C = x.shape
        y = torch.empty_like(x)
        _launch("mpa_log_softmax_fwd_f32", _p(x), M, C, _p(y), _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        M, C = y.shape
        gx = torch.empty_like(y)
        _launch("mpa_log_softmax_bwd_f32", _p(y), _p(g.contiguous()), M, C, _p(gx), _stream())
        return gx


def log_softmax(x):
    """F.log_softmax(x, -1) for fp32 logits [..., C] (models/repsurf/repsurf_ssg_umb.py:67): one launch each way."""
    _dev(x)
    lead = x.shape[:-1]
    return _LogSoftmax.apply(_f32(x).reshape(-1, x.shape[-1])).view(*lead, x.shape[-1])


class _SmoothLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target, eps, from_logits):
        M, C = x.shape
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        lse = torch.empty(M, dtype=torch.float32, device=x.device) if from_logits else None
        part = torch.empty(int(lib.mpa_smooth_loss_workspace_floats(M)), dtype=torch.float32, device=x.device)
        _launch("mpa_smooth_loss_fwd_f32", _p(x), _p(target), M, C, float(eps), int(from_logits), _p(lse), _p(part), _p(loss),
                _stream())
        ctx.save_for_backward(x, target, lse)
        ctx.cfg = (float(eps), int(from_logits))
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        x, target, lse = ctx.saved_tensors
        eps, from_logits = ctx.cfg
        M, C = x.shape
        gx = torch.empty_like(x)
        _launch("mpa_smooth_loss_bwd_f32", _p(x), _p(target), _p(lse), _p(g.contiguous().view(1)), M, C, eps, from_logits,
                _p(gx), _stream())
        return gx, None, None, None


def smooth_loss(x, target, eps=0.1, from_logits=False):
    """Label-smoothed loss over rows x [M,C] with int64 targets [M]: the mean of -sum_c w_c lp_c, w = 1-eps at the target
    and eps/(C-1) elsewhere.  from_logits=False: x are log-probabilities (reference util/utils.py:74-88); True: x are
    logits and lp = log_softmax(x) (models/repsurf/pointnet2_part_seg_msg.py:159-180).  Two launches forward (rows, then
    a fixed-order sum: bit-reproducible), one backward -- the reference's formulation is ~15 elementwise launches."""
    _dev(x, target)
    return _SmoothLoss.apply(_f32(x), _i64(target).view(-1), eps, bool(from_logits))


class _Fanout(torch.autograd.Function):
    """x -> n aliases of x, one per consumer; backward: the n gradients summed in ONE launch (mpa_add_n_*)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n = n
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g for g in grads if g is not None]
        if len(gs) == 1:
            return gs[0], None
        first = gs[0]
        C = first.shape[-1] if first.dim() else 0
        rows = first.numel() // C if C else 0
        lds = [_row_stride(g, C) for g in gs]
        ok = (first.dtype in (torch.float32, torch.bfloat16) and C and C % 4 == 0 and len(gs) <= 8
              and all(g.dtype == first.dtype and g.shape == first.shape for g in gs))
        if not ok:
            acc = gs[0]
            for g in gs[1:]:
                acc = acc + g
            return acc, None
        # (a contribution that is a column block of a wider tensor -- one stream of a concatenated gradient -- is read
        # in place through its row stride)
        gs = [g if ld else g.contiguous() for g, ld in zip(gs, lds)]
        out = torch.empty(first.shape, dtype=first.dtype, device=first.device)
        ptrs = (ctypes.c_void_p * len(gs))(*[g.data_ptr() for g in gs])
        strides = (ctypes.c_longlong * len(gs))(*[ld or C for ld in lds])
        align = 16 if first.dtype == torch.float32 else 8
        if any(g.data_ptr() % align or (ld or C) % 4 for g, ld in zip(gs, lds)):
            gs = [g.contiguous() for g in gs]
            ptrs = (ctypes.c_void_p * len(gs))(*[g.data_ptr() for g in gs])
            strides = (ctypes.c_longlong * len(gs))(*[C] * len(gs))
        _launch("mpa_add_n_" + _sfx(out), ptrs, strides, len(gs), rows, C, _p(out), _stream())
        return out, None


def _row_stride(t, C):
    """Row stride (elements) if `t` [..., C] is a set of equally spaced dense rows (a contiguous tensor or a column block
    of one), else 0."""
    if t.dim() == 0 or t.stride(-1) != 1 and t.shape[-1] != 1:
        return 0
    if t.dim() == 1:
        return C
    ld = t.stride(-2)
    if ld < C:
        return 0
    expect = ld
    for d in range(t.dim() - 2, -1, -1):
        if t.shape[d] != 1 and t.stride(d) != expect:
            return 0
        expect *= t.shape[d]
    return ld


def fanout(x, n):
    """n aliases of `x` for n consumers (modules that read one tensor several times: a LocalTrans pair's centres feed the
    query projection and both residuals, the part-seg encoder's states feed four Fuse calls and the next LocalMerge):
    the consumers' gradients meet in one summing launch instead of autograd's n - 1 pairwise adds.  Values are
    untouched (views); without a gradient to route it is `(x,) * n`."""
    if n <= 1 or not (torch.is_grad_enabled() and x.requires_grad) or not x.is_cuda:
        return (x,) * n
    return _Fanout.apply(x, n)


class _PoolMaxMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, P, C = x.shape
        out = torch.empty(B, 2 * C, dtype=torch.float32, device=x.device)
        arg = torch.empty(B, C, dtype=torch.int32, device=x.device)
        _launch("mpa_pool_max_mean_fwd_f32", _p(x), B, P, C, _p(out), _p(arg), _stream())
        ctx.save_for_backward(arg)
        ctx.dims = (B, P, C)
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        B, P, C = ctx.dims
        gx = torch.empty(B, P, C, dtype=torch.float32, device=g.device)
        _launch("mpa_pool_max_mean_bwd_f32", _p(g.contiguous()), _p(arg), B, P, C, _p(gx), _stream())
        return gx


def pool_max_mean(x):
    """torch.cat((x.max(dim=1)[0], x.mean(dim=1)), 1) for x [B,P,C] (the classification head's pooling over the
    points of the last state, reference modules/repsurface_utils.py:629-633): one launch each way; fp32 (bf16 rows
    are promoted: the result feeds the per-cloud layers, which run in fp32 anyway)."""
    _dev(x)
    return _PoolMaxMean.apply(_f32(_feat(x).float()))


class _CatBroadcast(torch.autograd.Function):
    """cat((a [B,N,Ca], rows [B,1,Cr] broadcast over N), 2): per-point features next to per-cloud rows (the
    part-seg head: conv5(points) | global maxima | label embedding, reference modules/pointnet2_utils.py:846-856).
    Backward hands `a` its column block of the gradient in place and sums the broadcast columns per cloud with
    mpa_group_col_sum -- not with torch's reduction, which autograd would pick for expand() and which returns
    wrong values from the second replay of a captured HIP graph."""

    @staticmethod
    def forward(ctx, a, rows):
        B, N, Ca = a.shape
        Cr = rows.shape[2]
        out = torch.empty(B, N, Ca + Cr, dtype=a.dtype, device=a.device)
        out[:, :, :Ca] = a
        out[:, :, Ca:] = rows
        ctx.dims = (B, N, Ca, Cr)
        return out

    @staticmethod
    def backward(ctx, g):
        B, N, Ca, Cr = ctx.dims
        g = g.contiguous()
        grows = torch.empty(B, 1, Cr, dtype=torch.float32, device=g.device)
        _launch("mpa_group_col_sum_" + _sfx(g), _vp(g.data_ptr() + g.element_size() * Ca), B, N, Cr, Ca + Cr, _p(grows),
                _stream())
        return g[:, :, :Ca], grows.to(g.dtype)


def cat_broadcast(a, rows):
    """torch.cat((a, rows.expand(-1, N, -1)), 2) for a [B,N,Ca] and per-cloud rows [B,1,Cr] (see _CatBroadcast)."""
    _dev(a, rows)
    return _CatBroadcast.apply(_feat(a), _feat(rows).to(a.dtype))


# ------------------------------------------------------------------------------- transition MLP
def _workspace(device, nbytes):
    """Split-K scratch for one call: an ordinary temporary from torch's allocator.  Under HIP-graph capture it
    comes from the graph's private pool and is recycled in stream order like every other temporary of the
    step.  (Round 1 kept ONE persistent buffer per (device, stream) and replaced it when a call needed more:
    replacing it in the middle of a capture freed memory whose address earlier nodes of the same graph had
    baked in -- when that buffer came from an older, already released graph pool the replay wrote into
    unmapped pages: "Memory access fault ... write access to a read-only page")"""
    return torch.empty(max((nbytes + 3) // 4, 4), dtype=torch.float32, device=device)


def _gemm(A, lda, tA, Bm, ldb, tB, bias, C, ldc, M, N, K, accumulate=0, tile_stats=None, a_col_sum=None, stats_acc=0):
    # stats_acc = R > 0: tile_stats is the pre-zeroed accumulate form [R][3][N] (see mpa_gemm_f32)
    if A.dtype == torch.bfloat16:
        return _gemm_bf16(A, lda, tA, Bm, ldb, tB, bias, C, ldc, M, N, K, accumulate, tile_stats, a_col_sum, stats_acc)
    ws, ws_bytes = None, 0
    ntiles = ((M + 63) // 64) * ((N + 63) // 64)
    if ntiles < 256 and K >= 512 and ldc == N:          # split-K partial tiles (see mpa_gemm_f32)
        splits = min((512 + ntiles - 1) // ntiles, K // 256)
        if splits > 1:
            ws = _workspace(C.device, splits * M * N * 4)
            ws_bytes = ws.numel() * 4
    # (mirrors the dispatch of mpa_gemm_f32: whole 64x64 tiles with K = 64 or 128 go to gemm_shortk_kernel)
    shortk = (not tA and not accumulate and a_col_sum is None and M % 64 == 0 and N % 64 == 0 and K in (64, 128)
              and lda % 4 == 0 and ldb % 4 == 0 and A.data_ptr() % 16 == 0 and Bm.data_ptr() % 16 == 0)
    _launch("mpa_gemm_f32", _p(A), lda, tA, _p(Bm), ldb, tB, _p(bias), _p(C), ldc, M, N, K, accumulate,
            _p(tile_stats), stats_acc, _p(a_col_sum), _p(ws), ws_bytes, _stream(), algo_bytes=4 * (M * K + N * K + M * N),
            algo_flops=2 * M * N * K, tag=(M, N, K, tA, tB, tile_stats is not None),
            variant="shortk" if shortk else "tiled")


def _gemm_bf16(A, lda, tA, Bm, ldb, tB, bias, C, ldc, M, N, K, accumulate, tile_stats, a_col_sum, stats_acc=0):
    """The same products on bf16 features (mpa_gemm_bf16 / mpa_gemm_tn_grouped_bf16): A is a bf16 activation or
    gradient; B the fp32 master weight (forward, dX) or a bf16 activation (dW); C bf16, or fp32 for logits
    and weight gradients."""
    if accumulate:
        raise NotImplementedError("bf16 products do not accumulate into C")
    if tA:                                          # dW = A^T B: one problem of the grouped launch
        if tB or Bm.dtype != torch.bfloat16 or C.dtype != torch.float32 or ldc != N:
            raise NotImplementedError("bf16 A^T B needs bf16 k-major operands and a dense fp32 output")
        _weight_grads_bf16([(A, lda, Bm, ldb, C, M, N, K, a_col_sum)])
        return
    if a_col_sum is not None:
        raise NotImplementedError("column sums ride on the weight-gradient product only")
    b32 = Bm.dtype == torch.float32
    c32 = C.dtype == torch.float32
    _launch("mpa_gemm_bf16", _p(A), lda, _p(Bm), ldb, 1 if tB else 0, int(b32), _p(bias), _p(C), ldc, int(c32), M, N, K,
            _p(tile_stats), stats_acc, _stream(), algo_bytes=2 * M * K + (4 if b32 else 2) * N * K + (4 if c32 else 2) * M * N,
            algo_flops=2 * M * N * K, tag=(M, N, K, tA, tB, tile_stats is not None))


def _weight_grads_bf16(queue):
    """queue entries: (gy, lda, x, ldb, out, M, N, K, a_col_sum) with bf16 gy / x and fp32 out."""
    n = len(queue)
    arr = (GemmTnProblemBf16 * n)()
    ws_bytes = 0
    for i, (gy, lda, x, ldb, out, M, N, K, acs) in enumerate(queue):
        arr[i].A, arr[i].B, arr[i].out = gy.data_ptr(), x.data_ptr(), out.data_ptr()
        arr[i].a_col_sum = acs.data_ptr() if acs is not None else None
        arr[i].lda, arr[i].ldb, arr[i].M, arr[i].N, arr[i].K = lda, ldb, M, N, K
        tiles = ((M + 63) // 64) * ((N + 63) // 64)
        if tiles < 256 and K >= 2048:
            splits = max(1, min((512 + tiles - 1) // tiles, K // 1024))
            ws_bytes += (splits * M * N * 4 + 255) // 256 * 256
    ws = _workspace(queue[0][0].device, max(ws_bytes, 4))
    _launch("mpa_gemm_tn_grouped_bf16", arr, n, _p(ws), ws.numel() * 4, _stream(),
            algo_bytes=sum(2 * q[7] * (q[5] + q[6]) + 4 * q[5] * q[6] for q in queue),
            algo_flops=sum(2 * q[5] * q[6] * q[7] for q in queue))


def _gemm_group(problems, tB):
    """Independent forward / dX products C = A op(B) (+ bias) as ONE launch (mpa_gemm_grouped_*): problems =
    [(A, lda, B, ldb, bias, C, ldc, M, N, K, tile_stats, stats_acc)].  fp32 problems that the short-K kernel
    takes (whole 64x64 tiles, K = 64 / 128: the fine states' layers, already at their HBM rate) keep their own
    launches; a single problem is a plain _gemm."""
    if not problems:
        return
    bf16 = problems[0][0].dtype == torch.bfloat16

    def shortk(q):
        A, lda, Bm, ldb, bias, C, ldc, M, N, K = q[:10]
        return (M % 64 == 0 and N % 64 == 0 and K in (64, 128) and lda % 4 == 0 and ldb % 4 == 0
                and A.data_ptr() % 16 == 0 and Bm.data_ptr() % 16 == 0)

    if len(problems) == 1 or len(problems) > 8 or (not bf16 and all(shortk(q) for q in problems)):
        for A, lda, Bm, ldb, bias, C, ldc, M, N, K, stats, acc in problems:
            _gemm(A, lda, 0, Bm, ldb, tB, bias, C, ldc, M, N, K, 0, stats, stats_acc=acc)
        return
    n = len(problems)
    arr = (GemmProblem * n)()
    for i, (A, lda, Bm, ldb, bias, C, ldc, M, N, K, stats, acc) in enumerate(problems):
        arr[i].A, arr[i].B, arr[i].C = A.data_ptr(), Bm.data_ptr(), C.data_ptr()
        arr[i].bias = bias.data_ptr() if bias is not None else None
        arr[i].tile_stats = stats.data_ptr() if stats is not None else None
        arr[i].lda, arr[i].ldb, arr[i].ldc, arr[i].M, arr[i].N, arr[i].K = lda, ldb, ldc, M, N, K
        arr[i].stats_replicas = acc
    flops = sum(2 * q[7] * q[8] * q[9] for q in problems)
    if bf16:
        b32 = problems[0][2].dtype == torch.float32
        _launch("mpa_gemm_grouped_bf16", arr, n, 1 if tB else 0, int(b32), _stream(), algo_flops=flops,
                algo_bytes=sum(2 * q[7] * q[9] + (4 if b32 else 2) * q[8] * q[9] + 2 * q[7] * q[8] for q in problems))
    else:
        _launch("mpa_gemm_grouped_f32", arr, n, 1 if tB else 0, _stream(), algo_flops=flops,
                algo_bytes=sum(4 * (q[7] * q[9] + q[8] * q[9] + q[7] * q[8]) for q in problems))


_ZEROS = {}


def _zeros_like_cached(device, n):
    """A shared read-only zero vector (gradients that are identically zero); autograd never
    writes into a returned gradient it does not own."""
    key = (device.index, n)
    z = _ZEROS.get(key)
    if z is None:
        z = _ZEROS[key] = torch.zeros(n, dtype=torch.float32, device=device)
    return z


# ---- deferred, grouped weight gradients ---------------------------------------------------------
# Nothing in a backward pass consumes a weight gradient, so when the gradients are written straight
# into flat buckets (GradReducer direct mode) the dW = dY^T X products can be queued and issued
# together at the end of backward (flush_weight_grads): ~60 latency-bound launches become two.
_DW_QUEUE = None          # None: immediate mode; list: deferral active


def defer_weight_grads(on=True):
    """Enable / disable queuing of direct-mode weight-gradient GEMMs (flush_weight_grads() required
    after every backward while enabled)."""
    global _DW_QUEUE
    if on and _DW_QUEUE is None:
        _DW_QUEUE = []
    elif not on:
        if _DW_QUEUE:
            flush_weight_grads()
        _DW_QUEUE = None


def _weight_grad(gy, lda, x, ldb, out, M, N, K, a_col_sum=None, direct=False):
    """out[M,N] = gy^T x  (gy stored [K][M] with row stride lda, x stored [K][N] with ldb)."""
    if direct and _DW_QUEUE is not None:
        _DW_QUEUE.append((gy, lda, x, ldb, out, M, N, K, a_col_sum))
    else:
        _gemm(gy, lda, 1, x, ldb, 0, None, out, N, M, N, K, a_col_sum=a_col_sum)


def flush_weight_grads(riders=None):
    """Issue every queued weight-gradient product as one grouped launch (+ one reduce) per storage type.  riders: a
    (GeoRider * n) array (GeometryPrefetch.riders()) carried by the fp32 launches, or launched alone if there are none."""
    if not _DW_QUEUE:
        if riders is not None:
            for i in range(len(riders)):
                _launch("mpa_geo_rider_f32", ctypes.byref(riders[i]), _stream())
        return
    q16 = [q for q in _DW_QUEUE if q[0].dtype == torch.bfloat16]
    q32 = [q for q in _DW_QUEUE if q[0].dtype != torch.bfloat16]
    if q16:
        _weight_grads_bf16(q16)
    if q32:
        n = len(q32)
        arr = (GemmTnProblem * n)()
        ws_bytes = 0
        for i, (gy, lda, x, ldb, out, M, N, K, acs) in enumerate(q32):
            arr[i].A, arr[i].B, arr[i].out = gy.data_ptr(), x.data_ptr(), out.data_ptr()
            arr[i].a_col_sum = acs.data_ptr() if acs is not None else None
            arr[i].lda, arr[i].ldb, arr[i].M, arr[i].N, arr[i].K = lda, ldb, M, N, K
            tiles = ((M + 63) // 64) * ((N + 63) // 64)
            if tiles < 256 and K >= 512:
                splits = max(1, min((512 + tiles - 1) // tiles, K // 256))
                ws_bytes += (splits * M * N * 4 + 255) // 256 * 256
        dev = q32[0][0].device
        ws = _workspace(dev, max(ws_bytes, 4))
        if riders is not None:
            _launch("mpa_gemm_tn_grouped_rider_f32", arr, n, _p(ws), ws.numel() * 4, riders, len(riders), _stream(),
                    algo_bytes=sum(4 * (q[7] * (q[5] + q[6]) + q[5] * q[6]) for q in q32),
                    algo_flops=sum(2 * q[5] * q[6] * q[7] for q in q32), timer="mpa_gemm_tn_grouped_f32")
            riders = None
        else:
            _launch("mpa_gemm_tn_grouped_f32", arr, n, _p(ws), ws.numel() * 4, _stream(),
                    algo_bytes=sum(4 * (q[7] * (q[5] + q[6]) + q[5] * q[6]) for q in q32),
                    algo_flops=sum(2 * q[5] * q[6] * q[7] for q in q32))
    if riders is not None:
        for i in range(len(riders)):
            _launch("mpa_geo_rider_f32", ctypes.byref(riders[i]), _stream())
    _DW_QUEUE.clear()


def _col_sum(x2d, ld=None):
    M, C = x2d.shape
    if x2d.dtype != torch.float32:
        return x2d.float().sum(0)
    out = torch.zeros(C, dtype=torch.float32, device=x2d.device)
    _launch("mpa_col_sum_f32", _p(x2d), M, C, C if ld is None else ld, _p(out), _stream())
    return out


def _col_sum_into(x2d, out, ld=None):
    """out[C] += column sums of x2d (out is a cleared flat-gradient view)."""
    M, C = x2d.shape
    if x2d.dtype != torch.float32:
        out.add_(x2d.float().sum(0))
        return
    _launch("mpa_col_sum_f32", _p(x2d), M, C, C if ld is None else ld, _p(out), _stream())


def _direct(t):
    """Flat-gradient view installed by distributed.GradReducer(direct=True): backward kernels write
    the parameter's gradient straight into it (the parameter is used once per step and the flat
    buffers are cleared at the start of the step), and autograd gets None for that input."""
    return getattr(t, "_mpa_grad_buf", None) if t is not None else None


def _rows_ld(t):
    """2-D gradient as (tensor, leading dimension): a column block of a wider row-major tensor (one
    input of a concatenation) is used in place, anything else is made contiguous."""
    if t.stride(1) == 1 and t.stride(0) >= t.shape[1] and t.data_ptr() % 16 == 0:
        return t, t.stride(0)
    t = t.contiguous()
    return t, t.shape[1]


class _Linear(torch.autograd.Function):
    """y[M,N] = x[M,K] W[N,K]^T + b on the fp32-MFMA GEMM; backward = two more GEMMs.
    bias_grad_is_zero: the caller knows the bias gradient vanishes identically (q / k projections
    of the difference attention: a shift of q or of every k_j leaves the softmax unchanged)."""

    @staticmethod
    def forward(ctx, x, W, b, bias_grad_is_zero, out_dtype=None):
        M, K = x.shape
        N = W.shape[0]
        y = torch.empty(M, N, dtype=out_dtype or x.dtype, device=x.device)
        _gemm(x, K, 0, W, K, 1, b, y, N, M, N, K)
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        ctx.zero_bias = bool(bias_grad_is_zero)
        ctx.direct = (_direct(W), _direct(b))
        return y

    @staticmethod
    def backward(ctx, gy):
        x, W = ctx.saved_tensors
        M, K = x.shape
        N = W.shape[0]
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)             # fp32 logits of a bf16 model: their gradient joins the bf16 stream
        gy, ldg = _rows_ld(gy)
        dW, db = ctx.direct
        gx = gW = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(M, K, dtype=x.dtype, device=x.device)
            _gemm(gy, ldg, 0, W, K, 0, None, gx, K, M, K, N)          # gy [M,N] @ W [N,K]
        want_b = ctx.has_bias and ctx.needs_input_grad[2] and not ctx.zero_bias
        gb_buf = None
        if want_b:
            gb_buf = db if db is not None else torch.zeros(N, dtype=torch.float32, device=x.device)
        if ctx.needs_input_grad[1]:
            gW = dW if dW is not None else torch.empty(N, K, dtype=torch.float32, device=x.device)
            # gy^T [N,M] @ x [M,K]; the bias gradient (column sums of gy) rides along
            _weight_grad(gy, ldg, x, K, gW, N, K, M, a_col_sum=gb_buf, direct=dW is not None and (gb_buf is None or db is not None))
            if dW is not None:
                gW = None
        elif want_b:
            _col_sum_into(gy, gb_buf, ld=ldg)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            if ctx.zero_bias:
                gb = None if db is not None else _zeros_like_cached(x.device, N)
            else:
                gb = None if db is not None else gb_buf
        return gx, gW, gb, None, None


SMALL_ROWS = 128


def _small_rows_fp32(x2):
    """bf16 rows of a per-CLOUD layer (the classification head: M = batch rows) are promoted to fp32: with so few
    rows the layer is its weight matrix streamed once (fp32 in HBM either way), the fp32 GEMM splits K over the chip
    where the bf16 kernel's one row tile walks all of K on a handful of workgroups (69 us against 12 for 2048 -> 1024
    on 64 rows), and everything downstream of it stays fp32 (more accurate, same cost)."""
    if x2.dtype == torch.bfloat16 and x2.shape[0] <= SMALL_ROWS:
        return x2.float()
    return x2


def linear(x, weight, bias, bias_grad_is_zero=False, out_dtype=None):
    """y = x W^T + b over the last dimension (nn.Linear), any leading shape.  out_dtype=torch.float32 on
    bf16 features: the product's fp32 results are stored unrounded (logits handed to a loss)."""
    _dev(x, weight)
    lead = x.shape[:-1]
    y = _Linear.apply(_small_rows_fp32(_feat(x).reshape(-1, x.shape[-1])), _f32(weight), bias, bias_grad_is_zero, out_dtype)
    return y.view(*lead, weight.shape[0])


def _stacked(a, b):
    """torch.cat((a, b), 0), as a view when b's storage directly follows a's."""
    if (a.is_contiguous() and b.is_contiguous() and a.shape[1:] == b.shape[1:]
            and b.data_ptr() == a.data_ptr() + a.numel() * a.element_size()
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()):
        rows = a.shape[0] + b.shape[0]
        return a.detach().as_strided((rows,) + tuple(a.shape[1:]), a.stride())
    return torch.cat((a.detach(), b.detach()), 0)


class _LinearKV(torch.autograd.Function):
    """kv[M, 2C] = x [Wk; Wv]^T + [bk; bv]: the key and value projections of LocalTrans' feature
    branch as one GEMM (keys in columns [0,C), values in [C,2C)), without materialising the
    concatenated parameters in autograd: weight gradients are two GEMMs on the two column
    blocks, dbk = 0 identically (see _Linear), dbv = column sums of the value block."""

    @staticmethod
    def forward(ctx, x, Wk, bk, Wv, bv):
        M, K = x.shape
        C = Wk.shape[0]
        # [Wk; Wv] and [bk; bv] without a copy when the parameters sit back to back in memory
        # (optim.FlatAdam's flat parameter buffers keep LocalTrans' k|v pairs adjacent)
        Wkv = _stacked(Wk, Wv)
        bkv = _stacked(bk, bv)
        kv = torch.empty(M, 2 * C, dtype=x.dtype, device=x.device)
        _gemm(x, K, 0, Wkv, K, 1, bkv, kv, 2 * C, M, 2 * C, K)
        ctx.save_for_backward(x, Wkv)
        ctx.direct = (_direct(Wk), _direct(bk), _direct(Wv), _direct(bv))
        return kv

    @staticmethod
    def backward(ctx, gkv):
        x, Wkv = ctx.saved_tensors
        M, K = x.shape
        C = Wkv.shape[0] // 2
        dev = x.device
        gkv = gkv.contiguous()
        dWk, dbk, dWv, dbv = ctx.direct
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(M, K, dtype=x.dtype, device=dev)
            _gemm(gkv, 2 * C, 0, Wkv, K, 0, None, gx, K, M, K, 2 * C)
        gv_view = gkv[:, C:]
        gWk = dWk if dWk is not None else torch.empty(C, K, dtype=torch.float32, device=dev)
        gWv = dWv if dWv is not None else torch.empty(C, K, dtype=torch.float32, device=dev)
        _weight_grad(gkv, 2 * C, x, K, gWk, C, K, M, direct=dWk is not None)                     # gk^T x
        # gv^T x; its A operand's column sums are dbv (accumulated by the same kernel)
        gbv = dbv if dbv is not None else torch.zeros(C, dtype=torch.float32, device=dev)
        _weight_grad(gv_view, 2 * C, x, K, gWv, C, K, M, a_col_sum=gbv, direct=dWv is not None and dbv is not None)
        if dbv is not None:
            gbv = None
        gbk = None if dbk is not None else _zeros_like_cached(dev, C)
        return (gx, None if dWk is not None else gWk, gbk, None if dWv is not None else gWv, gbv)


def linear_kv(x, k_lin, v_lin):
    """Projected keys | values [.., 2C] of LocalTrans' feature branch from the two nn.Linear."""
    _dev(x, k_lin.weight)
    lead = x.shape[:-1]
    kv = _LinearKV.apply(_feat(x).reshape(-1, x.shape[-1]), k_lin.weight, k_lin.bias, v_lin.weight, v_lin.bias)
    return kv.view(*lead, kv.shape[-1])


def _stacked_all(ts):
    """torch.cat(ts, 0) as a view when the tensors sit back to back in one storage."""
    out = ts[0]
    for t in ts[1:]:
        nxt = _stacked(out, t)
        if nxt.data_ptr() != out.data_ptr():             # not adjacent: one real concatenation
            return torch.cat([x.detach() for x in ts], 0)
        out = nxt
    return out.detach()


class _LinearStack(torch.autograd.Function):
    """y[M, sum N_i] = x [W_0; W_1; ...]^T + [b_0; b_1; ...]: several nn.Linear applied to the same
    rows as ONE GEMM (LocalMerge: q1|q2 on the centres, k1|v1|k2|v2 on the base rows); backward is one
    dX GEMM on the stacked weight and one grouped weight-gradient problem per layer.  zero_bias[i]:
    the caller knows that bias's gradient vanishes identically (q and k projections, see _Linear)."""

    @staticmethod
    def forward(ctx, x, zero_bias, *wb):
        Ws, bs = wb[0::2], wb[1::2]
        M, K = x.shape
        Wst, bst = _stacked_all(Ws), _stacked_all(bs)
        Nt = Wst.shape[0]
        y = torch.empty(M, Nt, dtype=x.dtype, device=x.device)
        _gemm(x, K, 0, Wst, K, 1, bst, y, Nt, M, Nt, K)
        ctx.save_for_backward(x, Wst)
        ctx.sizes = [w.shape[0] for w in Ws]
        ctx.zero_bias = tuple(zero_bias)
        ctx.direct = [(_direct(w), _direct(b)) for w, b in zip(Ws, bs)]
        return y

    @staticmethod
    def backward(ctx, gy):
        x, Wst = ctx.saved_tensors
        M, K = x.shape
        Nt = Wst.shape[0]
        dev = x.device
        gy, ldg = _rows_ld(gy)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(M, K, dtype=x.dtype, device=dev)
            _gemm(gy, ldg, 0, Wst, K, 0, None, gx, K, M, K, Nt)
        grads = []
        off = 0
        for n_i, zb, (dW, db) in zip(ctx.sizes, ctx.zero_bias, ctx.direct):
            blk = gy[:, off:off + n_i]
            gW = dW if dW is not None else torch.empty(n_i, K, dtype=torch.float32, device=dev)
            if zb:
                _weight_grad(blk, ldg, x, K, gW, n_i, K, M, direct=dW is not None)
                gb = None if db is not None else _zeros_like_cached(dev, n_i)
            else:
                gb_buf = db if db is not None else torch.zeros(n_i, dtype=torch.float32, device=dev)
                _weight_grad(blk, ldg, x, K, gW, n_i, K, M, a_col_sum=gb_buf, direct=dW is not None and db is not None)
                gb = None if db is not None else gb_buf
            grads += [None if dW is not None else gW, gb]
            off += n_i
        return (gx, None) + tuple(grads)


class _GatherStack(torch.autograd.Function):
    """(index_points(x, idx), x [W_0; W_1; ...]^T + b): the two consumers of a state's features in LocalMerge -- the
    sampled centres and the stacked key | value projections -- as ONE autograd node, so that x receives ONE gradient:
    the dX product writes it and the centres' gradient is scattered INTO it (fp32: row atomics; bf16: compare-and-swap
    on channel pairs), instead of a zero fill, a scatter, (a cast,) a dX product and an addition of the two."""

    @staticmethod
    def forward(ctx, x, idx, zero_bias, *wb):
        Ws, bs = wb[0::2], wb[1::2]
        B, N, K = x.shape
        S = idx.shape[1]
        x2 = x.view(B * N, K)
        Wst, bst = _stacked_all(Ws), _stacked_all(bs)
        Nt = Wst.shape[0]
        fs = torch.empty(B, S, K, dtype=x.dtype, device=x.device)
        _launch("mpa_gather_fwd_" + _sfx(x), _p(x), _p(idx), B, N, S, K, _p(fs), _stream())
        y = torch.empty(B * N, Nt, dtype=x.dtype, device=x.device)
        _gemm(x2, K, 0, Wst, K, 1, bst, y, Nt, B * N, Nt, K)
        ctx.save_for_backward(x2, Wst, idx)
        ctx.dims = (B, N, S, K)
        ctx.sizes = [w.shape[0] for w in Ws]
        ctx.zero_bias = tuple(zero_bias)
        ctx.direct = [(_direct(w), _direct(b)) for w, b in zip(Ws, bs)]
        return fs, y.view(B, N, Nt)

    @staticmethod
    def backward(ctx, g_fs, gy):
        x2, Wst, idx = ctx.saved_tensors
        B, N, S, K = ctx.dims
        M = B * N
        Nt = Wst.shape[0]
        dev = x2.device
        gy, ldg = _rows_ld(gy.reshape(M, Nt))
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(M, K, dtype=x2.dtype, device=dev)
            _gemm(gy, ldg, 0, Wst, K, 0, None, gx, K, M, K, Nt)                       # gradient through the projections
            _launch("mpa_gather_bwd_f32" if gx.dtype == torch.float32 else "mpa_gather_bwd_into_bf16",
                    _p(g_fs.contiguous()), _p(idx), B, N, S, K, _p(gx), _stream())                # += centres' gradient
            gx = gx.view(B, N, K)
        grads = []
        off = 0
        for n_i, zb, (dW, db) in zip(ctx.sizes, ctx.zero_bias, ctx.direct):
            blk = gy[:, off:off + n_i]
            gW = dW if dW is not None else torch.empty(n_i, K, dtype=torch.float32, device=dev)
            if zb:
                _weight_grad(blk, ldg, x2, K, gW, n_i, K, M, direct=dW is not None)
                gb = None if db is not None else _zeros_like_cached(dev, n_i)
            else:
                gb_buf = db if db is not None else torch.zeros(n_i, dtype=torch.float32, device=dev)
                _weight_grad(blk, ldg, x2, K, gW, n_i, K, M, a_col_sum=gb_buf, direct=dW is not None and db is not None)
                gb = None if db is not None else gb_buf
            grads += [None if dW is not None else gW, gb]
            off += n_i
        return (gx, None, None) + tuple(grads)


def gather_and_stack(x, idx, layers, zero_bias):
    """(index_points(x, idx), linear_stack(x, layers, zero_bias)) with one gradient into x (see _GatherStack)."""
    _dev(x, idx, layers[0].weight)
    if idx.dim() != 2 or (x.dtype == torch.bfloat16 and x.shape[-1] % 2):
        return index_points(x, idx), linear_stack(x, layers, zero_bias)
    wb = []
    for l in layers:
        wb += [l.weight, l.bias]
    return _GatherStack.apply(_feat(x), _i64(idx), tuple(zero_bias), *wb)


def linear_stack(x, layers, zero_bias):
    """[.., sum N_i] = the nn.Linear `layers` applied to x side by side (see _LinearStack)."""
    _dev(x, layers[0].weight)
    lead = x.shape[:-1]
    wb = []
    for l in layers:
        wb += [l.weight, l.bias]
    y = _LinearStack.apply(_feat(x).reshape(-1, x.shape[-1]), tuple(zero_bias), *wb)
    return y.view(*lead, y.shape[-1])


_BN_REPLICAS = 8


class ZeroArena:
    """Pre-zeroed fp32 scratch for the accumulators of one training pass (BatchNorm statistics that GEMM epilogues
    add into, the channel sums of the BatchNorm backward): take(n) hands out the next n floats, begin() clears
    what the previous pass used with ONE fill and rewinds.  Inside a captured HIP graph the buffer is persistent
    (allocated before the capture, kept alive by the arena) and the fill is the graph's first node, so every
    replay starts from zeros.  Outside a managed pass -- plain eager use of the ops -- take() falls back to
    torch.zeros (one fill per request)."""

    def __init__(self, device, nfloats=1 << 22):
        self.buf = torch.zeros(nfloats, dtype=torch.float32, device=device)
        self.cursor = 0
        self.high = 0
        self.active = False

    def begin(self):
        if self.high:
            self.buf[:self.high].zero_()
        self.cursor = 0
        self.active = True

    def end(self):
        self.high = max(self.high, self.cursor)
        self.active = False

    def take(self, n, device):
        n64 = (n + 63) // 64 * 64                    # slices start on 256-byte boundaries
        if not self.active or self.buf.device != device or self.cursor + n64 > self.buf.numel():
            return torch.zeros(n, dtype=torch.float32, device=device)
        s = self.buf[self.cursor:self.cursor + n]
        self.cursor += n64
        self.high = max(self.high, self.cursor)
        return s


_ARENA = None
# BatchNorm statistics: the default path accumulates the tiles' sums with float atomics (no finalize launch); the
# order of the adds varies from run to run, so two runs of the same step agree to fp32 rounding, not bit for
# bit (and the nets amplify that: max / LeakyReLU selections flip).  DETERMINISTIC_BN = True selects the
# per-tile statistics + mpa_bn_finalize_f32 path instead (fixed merge order, one more launch per Linear unit).
DETERMINISTIC_BN = False


def set_deterministic(on=True):
    """Bit-reproducible BatchNorm statistics (per-tile pairs merged in a fixed order by mpa_bn_finalize_f32)
    instead of the default atomically accumulated sums.  Returns the previous setting."""
    global DETERMINISTIC_BN
    old, DETERMINISTIC_BN = DETERMINISTIC_BN, bool(on)
    return old


def set_arena(arena):
    """Install (or remove, None) the ZeroArena the Linear units draw their accumulators from."""
    global _ARENA
    _ARENA = arena


def _zeros_acc(n, device):
    if _ARENA is not None:
        return _ARENA.take(n, device)
    return torch.zeros(n, dtype=torch.float32, device=device)


class _LinearBNAct(torch.autograd.Function):
    """Linear -> BatchNorm1d over the rows -> LeakyReLU (+ residual), as one unit: the GEMM epilogue
    yields the batch statistics, one elementwise kernel normalises + activates (+ adds the
    residual), backward is reduce + apply + two GEMMs."""

    @staticmethod
    def forward(ctx, x, W, b, gamma, beta, running_mean, running_var, nbt, residual, training, momentum, eps, slope):
        M, K = x.shape
        N = W.shape[0]
        dev = x.device
        y = torch.empty(M, N, dtype=x.dtype, device=dev)
        # batch statistics: every 64-row tile of the GEMM adds its (sum, within-tile M2, sum^2/rows) per column into
        # a pre-zeroed [R][3][N] accumulator (R replicas spread the float atomics of the many-tile layers); the
        # normalise kernel finishes mean / variance in its prologue -- no separate finalize launch
        out = torch.empty(M, N, dtype=x.dtype, device=dev)
        saved = torch.empty(2, N, dtype=torch.float32, device=dev)
        direct = (_direct(W), _direct(b), _direct(gamma), _direct(beta))
        need = any(ctx.needs_input_grad)
        if DETERMINISTIC_BN:
            # per-tile (sum, M2) pairs, merged in a fixed order by a finalize launch (which also clears `sums`)
            stats = torch.empty((M + 63) // 64, 2, N, dtype=torch.float32, device=dev) if training else None
            _gemm(x, K, 0, W, K, 1, b, y, N, M, N, K, 0, stats)
            sums = torch.empty(_BN_REPLICAS, 2, N, dtype=torch.float32, device=dev) if need else None
            _launch("mpa_bn_finalize_f32", _p(stats), M, N, _p(running_mean), _p(running_var), int(training),
                    float(momentum), float(eps), _p(saved), _p(sums), _BN_REPLICAS * 2 * N, _p(nbt), _stream())
            _launch("mpa_bn_act_fwd_" + _sfx(y), _p(y), _p(saved), _p(gamma), _p(beta), _p(residual), float(slope), M, N,
                    _p(out), _stream())
        else:
            R = 8 if (M + 63) // 64 >= 256 else 1
            stats = _zeros_acc(R * 3 * N, dev) if training else None
            _gemm(x, K, 0, W, K, 1, b, y, N, M, N, K, 0, stats, stats_acc=R)
            # sums: [_BN_REPLICAS][2][N] accumulator of backward's channel reductions (atomics spread over replicas)
            sums = _zeros_acc(_BN_REPLICAS * 2 * N, dev).view(_BN_REPLICAS, 2, N) if need else None
            _launch("mpa_bn_stats_act_fwd_" + _sfx(y), _p(y), _p(stats), R, M, N, _p(running_mean), _p(running_var),
                    int(training), float(momentum), float(eps), _p(nbt), _p(gamma), _p(beta), _p(residual), float(slope),
                    _p(out), _p(saved), _stream())
        ctx.save_for_backward(x, W, y, gamma, beta, saved, sums)
        ctx.cfg = (bool(training), float(slope), b is not None, residual is not None)
        ctx.direct = direct
        ctx.ran_backward = False
        return out

    @staticmethod
    def backward(ctx, gout):
        x, W, y, gamma, beta, saved, sums = ctx.saved_tensors
        training, slope, has_bias, has_res = ctx.cfg
        dW, db, dgamma, dbeta = ctx.direct
        M, K = x.shape
        N = W.shape[0]
        dev = x.device
        gout, ldg = _rows_ld(gout)
        if ctx.ran_backward:            # double backward through the same node: accumulator is dirty
            sums = torch.zeros(_BN_REPLICAS, 2, N, dtype=torch.float32, device=dev)
        ctx.ran_backward = True
        _launch("mpa_bn_act_bwd_reduce_" + _sfx(y), _p(y), _p(gout), _p(saved[0]), _p(saved[1]), _p(gamma), _p(beta), slope,
                M, N, ldg, _p(sums), _BN_REPLICAS, _stream())
        gy = torch.empty(M, N, dtype=y.dtype, device=dev)
        # dgamma / dbeta: stored by the apply pass, straight into the flat gradients when installed
        gg = dgamma if dgamma is not None else torch.empty(N, dtype=torch.float32, device=dev)
        gb_ = dbeta if dbeta is not None else torch.empty(N, dtype=torch.float32, device=dev)
        _launch("mpa_bn_act_bwd_apply_" + _sfx(y), _p(y), _p(gout), _p(saved[0]), _p(saved[1]), _p(gamma), _p(beta),
                _p(sums), _BN_REPLICAS, slope, int(training), M, N, ldg, _p(gy), _p(gg), _p(gb_), _stream())
        gx = gW = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(M, K, dtype=x.dtype, device=dev)
            _gemm(gy, N, 0, W, K, 0, None, gx, K, M, K, N)
        if ctx.needs_input_grad[1]:
            gW = dW if dW is not None else torch.empty(N, K, dtype=torch.float32, device=dev)
            _weight_grad(gy, N, x, K, gW, N, K, M, direct=dW is not None)
            if dW is not None:
                gW = None
        if has_bias and ctx.needs_input_grad[2]:
            # In front of a train-mode BatchNorm the bias gradient is identically zero (the batch
            # mean removes any constant shift); the reference's value there is fp32 rounding noise.
            if training:
                gb = None if db is not None else _zeros_like_cached(dev, N)
            elif db is not None:
                _col_sum_into(gy, db)
            else:
                gb = _col_sum(gy)
        ggamma = None if dgamma is not None else gg
        gbeta = None if dbeta is not None else gb_
        gres = gout if has_res else None
        return gx, gW, gb, ggamma, gbeta, None, None, None, gres, None, None, None, None


class _LinearBNActGroup(torch.autograd.Function):
    """n INDEPENDENT Linear -> BatchNorm1d -> LeakyReLU (+ residual) units in one autograd node: their forward
    products are one grouped launch, their dX products another, and so are the three BatchNorm passes (the units of
    LocalMerge's parallel attention streams and of Fuse's four source states are ~1 GFLOP each: alone they leave
    the chip half empty).  mode:
      "each"    n outputs;
      "concat"  one output [M, sum N_i], the units' outputs side by side (torch.cat((f1, f2), 2) without the copy:
                every unit writes its column block, backward reads its block of the gradient in place);
      "chain"   one output, residual_0 + sum_i unit_i(x_i) added in unit order (Fuse's accumulation over its source
                states: one pass over the rows, no intermediate sums in memory).
    ts: 9 tensors per unit -- x [M,K], W, b, gamma, beta, running_mean, running_var, num_batches_tracked, residual."""

    @staticmethod
    def forward(ctx, cfg, *ts):
        units_cfg, mode = cfg
        chain, concat = mode == "chain", mode == "concat"
        n = len(units_cfg)
        U = [ts[9 * i:9 * i + 9] for i in range(n)]
        dev = U[0][0].device
        ys, stats, Rs = [], [], []
        probs = []
        for (training, momentum, eps, slope), (x, W, b, gamma, beta, rm, rv, nbt, res) in zip(units_cfg, U):
            M, K = x.shape
            N = W.shape[0]
            y = torch.empty(M, N, dtype=x.dtype, device=dev)
            R = 8 if (M + 63) // 64 >= 256 else 1
            st = _zeros_acc(R * 3 * N, dev) if training else None
            probs.append((x, K, W, K, b, y, N, M, N, K, st, R if training else 0))
            ys.append(y); stats.append(st); Rs.append(R)
        _gemm_group(probs, 1)
        outs, saved_all, sums_all = [], [], []
        arr = (BnUnit * n)()
        wide = None
        if concat:
            Nt = sum(y.shape[1] for y in ys)
            wide = torch.empty(ys[0].shape[0], Nt, dtype=ys[0].dtype, device=dev)
        off = 0
        for i, ((training, momentum, eps, slope), (x, W, b, gamma, beta, rm, rv, nbt, res)) in enumerate(zip(units_cfg, U)):
            M, N = ys[i].shape
            if concat:
                out = wide[:, off:off + N]
                off += N
            else:
                out = torch.empty(M, N, dtype=x.dtype, device=dev) if ((not chain) or i == n - 1) else None
            saved = torch.empty(2, N, dtype=torch.float32, device=dev)
            sums = _zeros_acc(_BN_REPLICAS * 2 * N, dev).view(_BN_REPLICAS, 2, N)
            u = arr[i]
            u.x, u.stats, u.running_mean, u.running_var = _p(ys[i]), _p(stats[i]), _p(rm), _p(rv)
            u.num_batches_tracked, u.gamma, u.beta = _p(nbt), _p(gamma), _p(beta)
            u.residual = _p(res if (not chain or i == 0) else None)
            u.y, u.save = _p(out), _p(saved)
            u.ldy = wide.shape[1] if concat else N
            u.M, u.C, u.stats_replicas, u.training = M, N, Rs[i], int(training)
            u.momentum, u.eps, u.slope = float(momentum), float(eps), float(slope)
            outs.append(out); saved_all.append(saved); sums_all.append(sums)
        # one launch normalises + activates every unit (chain: and sums them onto unit 0's residual in one pass)
        _launch("mpa_bn_group_fwd_" + _sfx(ys[0]), arr, n, int(chain), _stream())
        keep = []
        for i, (x, W, b, gamma, beta, rm, rv, nbt, res) in enumerate(U):
            keep += [x, W, ys[i], gamma, beta, saved_all[i], sums_all[i]]
        ctx.save_for_backward(*keep)
        ctx.cfg = (units_cfg, mode, [u[2] is not None for u in U], [u[8] is not None for u in U])
        ctx.direct = [(_direct(u[1]), _direct(u[2]), _direct(u[3]), _direct(u[4])) for u in U]
        ctx.ran_backward = False
        if concat:
            return wide
        return outs[-1] if chain else tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        units_cfg, mode, has_bias, has_res = ctx.cfg
        chain, concat = mode == "chain", mode == "concat"
        n = len(units_cfg)
        S = ctx.saved_tensors
        dev = S[0].device
        if ctx.ran_backward:
            raise RuntimeError("_LinearBNActGroup: a second backward through the same node is not supported")
        ctx.ran_backward = True
        gys, probs, gxs, g_in = [], [], [None] * n, []
        arr = (BnUnit * n)()
        wide = ldw = None
        if concat:
            wide, ldw = _rows_ld(gouts[0])
            if ldw % 4:
                wide = wide.contiguous()
                ldw = wide.shape[1]
        off = 0
        for i in range(n):
            x, W, y, gamma, beta, saved, sums = S[7 * i:7 * i + 7]
            training, momentum, eps, slope = units_cfg[i]
            M, K = x.shape
            N = W.shape[0]
            if concat:
                gout, ldg = wide[:, off:off + N], ldw
                off += N
            else:
                gout, ldg = _rows_ld(gouts[0] if chain else gouts[i])
                if ldg % 4:
                    gout, ldg = gout.contiguous(), N
            dW, db, dgamma, dbeta = ctx.direct[i]
            gy = torch.empty(M, N, dtype=y.dtype, device=dev)
            gg = dgamma if dgamma is not None else torch.empty(N, dtype=torch.float32, device=dev)
            gb_ = dbeta if dbeta is not None else torch.empty(N, dtype=torch.float32, device=dev)
            u = arr[i]
            u.x, u.gamma, u.beta, u.save = _p(y), _p(gamma), _p(beta), _p(saved)
            u.grad_y, u.partial, u.grad_x, u.dgamma, u.dbeta = _p(gout), _p(sums), _p(gy), _p(gg), _p(gb_)
            u.M, u.C, u.ldg, u.training, u.replicas, u.slope = M, N, ldg, int(training), _BN_REPLICAS, float(slope)
            g_in.append(gout)
            gys.append((gy, gg, gb_))
            if ctx.needs_input_grad[1 + 9 * i]:
                gx = torch.empty(M, K, dtype=x.dtype, device=dev)
                gxs[i] = gx
                probs.append((gy, N, W, K, None, gx, K, M, K, N, None, 0))
        _launch("mpa_bn_group_bwd_reduce_" + _sfx(S[2]), arr, n, _stream())
        _launch("mpa_bn_group_bwd_apply_" + _sfx(S[2]), arr, n, _stream())
        _gemm_group(probs, 0)                                     # dX_i = gy_i W_i: one launch
        grads = []
        for i in range(n):
            x, W, y, gamma, beta, saved, sums = S[7 * i:7 * i + 7]
            training = units_cfg[i][0]
            M, K = x.shape
            N = W.shape[0]
            gy, gg, gb_ = gys[i]
            dW, db, dgamma, dbeta = ctx.direct[i]
            gW = gb = None
            if ctx.needs_input_grad[1 + 9 * i + 1]:
                gW = dW if dW is not None else torch.empty(N, K, dtype=torch.float32, device=dev)
                _weight_grad(gy, N, x, K, gW, N, K, M, direct=dW is not None)
                if dW is not None:
                    gW = None
            if has_bias[i] and ctx.needs_input_grad[1 + 9 * i + 2]:
                if training:                       # identically zero in front of a train-mode BatchNorm (see _LinearBNAct)
                    gb = None if db is not None else _zeros_like_cached(dev, N)
                elif db is not None:
                    _col_sum_into(gy, db)
                else:
                    gb = _col_sum(gy)
            gres = g_in[i] if (has_res[i] and (not chain or i == 0)) else None
            grads += [gxs[i], gW, gb, None if dgamma is not None else gg, None if dbeta is not None else gb_, None, None,
                      None, gres]
        return (None,) + tuple(grads)


def linear_bn_act_group(xs, linears, bns, slopes, residuals=None, mode="each"):
    """The n independent Linear units (nn.Linear `linears[i]`, nn.BatchNorm1d `bns[i]`, LeakyReLU slope or None) with
    optional residuals, see _LinearBNActGroup: mode "each" -> [unit_i(x_i)], "concat" -> torch.cat of them along the
    channels, "chain" -> residuals[0] + sum_i unit_i(x_i)."""
    n = len(xs)
    residuals = residuals or [None] * n
    chain, concat = mode == "chain", mode == "concat"
    if (DETERMINISTIC_BN or n == 1 or n > 8 or any(l.weight.shape[0] % 4 for l in linears)
            or any(x.dtype != xs[0].dtype for x in xs)):
        outs, acc = [], residuals[0]
        for i in range(n):
            o = linear_bn_act(xs[i], linears[i].weight, linears[i].bias, bns[i], slopes[i],
                              residual=(acc if chain else residuals[i]))
            outs.append(o)
            acc = o
        return outs[-1] if chain else (torch.cat(outs, -1) if concat else outs)
    _dev(*xs)
    lead = [x.shape[:-1] for x in xs]
    cfg, ts = [], []
    for i in range(n):
        bn = bns[i]
        training = bn.training or bn.running_mean is None
        x2 = _feat(xs[i]).reshape(-1, xs[i].shape[-1])
        if training and x2.shape[0] <= 1:
            raise ValueError("Expected more than 1 value per channel when training (BatchNorm1d)")
        W = linears[i].weight
        res = residuals[i] if (not chain or i == 0) else None
        res2 = None if res is None else _feat(res).to(x2.dtype).reshape(-1, W.shape[0])
        cfg.append((bool(training), 0.1 if bn.momentum is None else bn.momentum, bn.eps,
                    1.0 if slopes[i] is None else slopes[i]))
        ts += [x2, _f32(W), linears[i].bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
               bn.num_batches_tracked if training else None, res2]
    out = _LinearBNActGroup.apply((tuple(cfg), mode), *ts)
    if chain:
        return out.view(*lead[-1], linears[-1].weight.shape[0])
    if concat:
        return out.view(*lead[0], out.shape[-1])
    return [o.view(*lead[i], linears[i].weight.shape[0]) for i, o in enumerate(out)]


def linear_bn_act(x, weight, bias, bn, slope, residual=None):
    """The reference's Linear unit (modules/pointnet2_utils.py:413-425): affine, BatchNorm1d
    over the B*S rows (batch statistics in training, running statistics in eval; `bn` is the
    nn.BatchNorm1d holding gamma/beta/running stats), LeakyReLU(slope) unless slope is None.
    x is [B,S,C] or [M,C].  residual (same shape as the output) is added after the activation in
    the same kernel (LocalTrans' `residual + ffn(context)`, Fuse's `conv(x) + f`)."""
    _dev(x, weight)
    lead = x.shape[:-1]
    x2 = _small_rows_fp32(_feat(x).reshape(-1, x.shape[-1]))
    training = bn.training or bn.running_mean is None
    if training and x2.shape[0] <= 1:
        raise ValueError("Expected more than 1 value per channel when training (BatchNorm1d)")
    momentum = 0.1 if bn.momentum is None else bn.momentum
    res2 = None if residual is None else _feat(residual).to(x2.dtype).reshape(-1, weight.shape[0])
    out = _LinearBNAct.apply(x2, _f32(weight), bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                             bn.num_batches_tracked if training else None, res2, training, momentum, bn.eps,
                             1.0 if slope is None else slope)
    return out.view(*lead, weight.shape[0])
