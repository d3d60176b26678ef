"""GPU parity tests of the L0 point-set ops: HIP kernels (through the C ABI, via mpa_amd.ops)
against (a) golden vectors produced by the reference itself and (b) the C oracle on seeded
inputs.  Index results must be bit-exact; distances bitwise equal."""
import numpy as np
import pytest
import torch

from conftest import bits
from param_fill import unit_cloud, randn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import mpa_amd
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return mpa_amd.ops


@pytest.fixture(scope="module")
def co():
    from oracle import c_oracle
    return c_oracle


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def GL(a):
    return torch.from_numpy(np.ascontiguousarray(a).astype(np.int64)).cuda()


# --------------------------------------------------------------------------- FPS
@pytest.mark.parametrize("tag", ["fps_a", "fps_b", "fps_c", "fps_d"])
def test_fps_golden(ops, golden_index, tag):
    g = golden_index
    S = g[tag + "/idx"].shape[1]
    idx, sub = ops.farthest_point_sample(G(g[tag + "/xyz"]), S, start_idx=torch.from_numpy(g[tag + "/start"]),
                                         return_xyz=True)
    assert np.array_equal(idx.cpu().numpy(), g[tag + "/idx"].astype(np.int64))
    ref_sub = np.take_along_axis(g[tag + "/xyz"], g[tag + "/idx"].astype(np.int64)[..., None], axis=1)
    assert np.array_equal(sub.cpu().numpy(), ref_sub)


def test_fps_chain_golden(ops, golden_index):
    g = golden_index
    cur = G(g["fps_chain/xyz"])
    for lvl, S in enumerate((512, 256, 128, 64, 32)):
        idx, cur = ops.farthest_point_sample(cur, S, start_idx=torch.from_numpy(g["fps_chain/start%d" % lvl]),
                                             return_xyz=True)
        assert np.array_equal(idx.cpu().numpy(), g["fps_chain/idx%d" % lvl].astype(np.int64)), lvl


def test_fps_seeded_like_reference(ops, golden_index):
    """Same torch.manual_seed => same start indices as the reference draws them (CPU generator)."""
    g = golden_index
    torch.manual_seed(1024 + 512)
    idx = ops.farthest_point_sample(G(g["fps_a/xyz"]), 512)
    assert np.array_equal(idx.cpu().numpy(), g["fps_a/idx"].astype(np.int64))


@pytest.mark.parametrize("B,N,S", [(3, 64, 64), (2, 65, 10), (5, 130, 77), (2, 300, 150), (2, 777, 256),
                                   (2, 1500, 300), (1, 3000, 500), (1, 8192, 128), (7, 1, 1)])
def test_fps_vs_oracle(ops, co, B, N, S):
    xyz = unit_cloud(B, N, seed=N * 3 + S) if N > 1 else torch.zeros(B, 1, 3)
    start = torch.randint(0, N, (B,), generator=torch.Generator().manual_seed(N))
    got = ops.farthest_point_sample(xyz.cuda(), S, start_idx=start).cpu().numpy()
    assert np.array_equal(got, co.farthest_point_sample(xyz.numpy(), S, start.numpy()))


def test_fps_duplicates_first_max(ops, co):
    """Duplicate points: once all distances are 0 the reference's argmax returns index 0."""
    xyz = unit_cloud(2, 40, seed=9).repeat(1, 3, 1).contiguous()       # every point three times
    start = torch.tensor([5, 77])
    got = ops.farthest_point_sample(xyz.cuda(), 100, start_idx=start).cpu().numpy()
    assert np.array_equal(got, co.farthest_point_sample(xyz.numpy(), 100, start.numpy()))


def test_fps_full_size_properties(ops):
    """BASELINE config size (B=64, N=1024 -> 512): samples are distinct and start where told."""
    xyz = unit_cloud(64, 1024, seed=1234).cuda()
    start = torch.arange(64) * 7
    idx = ops.farthest_point_sample(xyz, 512, start_idx=start)
    assert (idx[:, 0].cpu() == start).all()
    s = idx.sort(dim=1)[0]
    assert (s[:, 1:] != s[:, :-1]).all()
    assert int(idx.min()) >= 0 and int(idx.max()) < 1024


# --------------------------------------------------------------------------- kNN
@pytest.mark.parametrize("tag", ["knn_a", "knn_b", "knn_c", "knn_d", "knn_e", "knn_f", "knn_g", "knn_h"])
def test_knn_golden(ops, golden_index, tag):
    g = golden_index
    dist, idx = ops.knn_point(8, G(g[tag + "/base"]), G(g[tag + "/query"]))
    assert np.array_equal(idx.cpu().numpy(), g[tag + "/idx"].astype(np.int64))
    assert np.array_equal(bits(dist.cpu().numpy()), bits(g[tag + "/dist"]))


@pytest.mark.parametrize("B,S,N,C,K", [(2, 70, 100, 3, 8), (1, 64, 64, 3, 3), (2, 1, 9, 3, 9), (2, 200, 300, 3, 16),
                                       (1, 33, 400, 3, 32), (2, 100, 200, 16, 8), (2, 90, 150, 32, 8),
                                       (1, 65, 130, 8, 4), (2, 40, 300, 64, 3), (1, 64, 96, 128, 16),
                                       (1, 20, 50, 256, 8), (1, 30, 70, 512, 8), (2, 17, 23, 5, 2)])
def test_knn_vs_oracle(ops, co, B, S, N, C, K):
    base = unit_cloud(B, N, seed=S + N) if C == 3 else randn((B, N, C), seed=S * N + C)
    query = randn((B, S, C), seed=C + 17, scale=0.5)
    dist, idx = ops.knn_point(K, base.cuda(), query.cuda())
    od, oi = co.knn_point(K, base.numpy(), query.numpy())
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(bits(dist.cpu().numpy()), bits(od))


def test_knn_ties_lowest_index_first(ops):
    base = torch.zeros(1, 16, 3).cuda()
    _, idx = ops.knn_point(8, base, base[:, :4].contiguous())
    assert (idx[0, 0].cpu() == torch.arange(8)).all()


def test_knn_full_size_properties(ops):
    """B=64, (S,N)=(512,1024): ascending distances, in-range distinct indices, and the query
    (a member of the base set) finds itself first."""
    xyz = unit_cloud(64, 1024, seed=1234).cuda()
    q = xyz[:, ::2].contiguous()
    dist, idx = ops.knn_point(8, xyz, q)
    assert (dist[:, :, 1:] >= dist[:, :, :-1]).all()
    assert int(idx.min()) >= 0 and int(idx.max()) < 1024
    assert (idx[:, :, 0] == (torch.arange(512, device="cuda") * 2)).all()
    s = idx.sort(dim=2)[0]
    assert (s[:, :, 1:] != s[:, :, :-1]).all()


def test_square_distance_golden(ops, golden_index):
    g = golden_index
    out = ops.square_distance(G(g["sqd/src"]), G(g["sqd/dst"]))
    assert np.array_equal(bits(out.cpu().numpy()), bits(g["sqd/out"]))


@pytest.mark.parametrize("tag", ["ball_a", "ball_b", "ball_c"])
def test_ball_query_golden(ops, golden_index, tag):
    g = golden_index
    idx = ops.query_ball_point(float(g[tag + "/radius"]), 24, G(g[tag + "/base"]), G(g[tag + "/query"]))
    assert np.array_equal(idx.cpu().numpy(), g[tag + "/idx"].astype(np.int64))


def test_ball_query_no_hit_rows(ops, co):
    base = unit_cloud(1, 50, seed=3)
    query = base[:, :10] + 10.0          # far away: no hit, rows are filled with N
    got = ops.query_ball_point(0.1, 6, base.cuda(), query.contiguous().cuda()).cpu().numpy()
    assert np.array_equal(got, co.query_ball_point(0.1, 6, base.numpy(), query.numpy()))
    assert (got == 50).all()


def test_three_nn_golden(ops, golden_index):
    g = golden_index
    dist, idx = ops.three_nn(G(g["nn3/xyz1"]), G(g["nn3/xyz2"]))
    assert np.array_equal(idx.cpu().numpy(), g["nn3/idx"].astype(np.int64))
    assert np.array_equal(bits(dist.cpu().numpy()), bits(g["nn3/dist"]))


# --------------------------------------------------------------------------- gathers
@pytest.mark.parametrize("C", [3, 64, 10])
def test_index_points_fwd_bwd(ops, C):
    B, N, S, K = 3, 50, 20, 8
    pts = randn((B, N, C), seed=C).cuda().requires_grad_(True)
    idx = torch.randint(0, N, (B, S, K), generator=torch.Generator().manual_seed(1)).cuda()
    out = ops.index_points(pts, idx)
    ref = pts.detach()[torch.arange(B, device="cuda").view(B, 1, 1), idx]
    assert torch.equal(out.detach(), ref)
    g = randn(out.shape, seed=2).cuda()
    out.backward(g)
    want = torch.zeros(B, N, C, device="cuda")
    want.view(B * N, C).index_add_(0, (idx + torch.arange(B, device="cuda").view(B, 1, 1) * N).view(-1),
                                    g.view(-1, C))
    assert (pts.grad - want).abs().max() < 1e-5
    out2 = ops.index_points(pts, idx[:, :, 0])
    assert torch.equal(out2.detach(), ref[:, :, 0])


def test_ops_reject_cpu_tensors(ops):
    with pytest.raises(RuntimeError):
        ops.knn_point(8, torch.zeros(1, 16, 3), torch.zeros(1, 4, 3))
    with pytest.raises(RuntimeError):
        ops.farthest_point_sample(torch.zeros(1, 16, 3), 4)


# --------------------------------------------------------------------------- fp32 MFMA GEMM / BN
@pytest.mark.parametrize("M,N,K,tA,tB", [(300, 64, 64, 0, 1), (1000, 40, 256, 0, 1), (513, 128, 3, 0, 1),
                                         (2048, 512, 1024, 0, 1), (4096, 64, 128, 0, 0), (333, 3, 64, 0, 0),
                                         (64, 64, 8192, 1, 0), (128, 3, 5000, 1, 0), (1024, 2048, 64, 1, 0),
                                         (70, 50, 30, 1, 1), (65536, 64, 64, 0, 1),
                                         (8192, 128, 64, 0, 0), (2048, 256, 128, 0, 1)])      # (+ short-K kernel: NN/NT, K = 64/128)
def test_gemm_vs_torch(ops, M, N, K, tA, tB):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if tA else (M, K), generator=g).cuda()
    Bm = torch.randn((N, K) if tB else (K, N), generator=g).cuda()
    bias = torch.randn(N, generator=g).cuda()
    C = torch.empty(M, N, device="cuda")
    tiles = (M + 63) // 64
    st3 = torch.full((tiles, 2, N), float("nan"), device="cuda")
    acs = torch.zeros(M, device="cuda") if tA else None
    ops._gemm(A, A.shape[1], tA, Bm, Bm.shape[1], tB, bias, C, N, M, N, K, 0, st3, a_col_sum=acs)
    if tA:
        want = A.double().sum(0)
        assert float((acs.double() - want).abs().max()) < 1e-5 * float(A.double().abs().sum(0).max())
    ref = (A.double().t() if tA else A.double()) @ (Bm.double().t() if tB else Bm.double()) + bias.double()
    scale = float(ref.abs().max())
    assert float((C.double() - ref).abs().max()) < 2e-6 * scale * max(1.0, K ** 0.5 / 8)
    # tile statistics: per 64-row tile, column sum and squared deviations from the tile mean
    pad = tiles * 64 - M
    rp = torch.cat([ref, ref.new_full((pad, N), float("nan"))]) if pad else ref
    rt = rp.view(tiles, 64, N)
    tsum = torch.nansum(rt, 1)
    cnt = (~torch.isnan(rt)).sum(1)
    tm2 = torch.nansum((rt - (tsum / cnt).unsqueeze(1)) ** 2, 1)
    assert float((st3[:, 0].double() - tsum).abs().max()) < 1e-5 * float(ref.abs().max()) * 64
    assert float((st3[:, 1].double() - tm2).abs().max()) < 1e-4 * float(tm2.max())


def test_gemm_accumulate(ops):
    A = torch.randn(200, 96, device="cuda")
    Bm = torch.randn(96, 80, device="cuda")
    C = torch.ones(200, 80, device="cuda")
    ops._gemm(A, 96, 0, Bm, 80, 0, None, C, 80, 200, 80, 96, 1)
    assert float((C - (A @ Bm + 1)).abs().max()) < 1e-3


def test_gemm_is_fmaf_chain(ops):
    """f32 MFMA == sequential fmaf over k: integer-valued operands give exact results and the
    result does not depend on the tile configuration (M large vs small picks different tiles)."""
    g = torch.Generator().manual_seed(0)
    A = torch.randint(-8, 9, (4096, 64), generator=g).float().cuda()
    W = torch.randint(-8, 9, (64, 64), generator=g).float().cuda()
    C1 = torch.empty(4096, 64, device="cuda")
    ops._gemm(A, 64, 0, W, 64, 1, None, C1, 64, 4096, 64, 64)
    assert torch.equal(C1, A @ W.t())
    x = torch.randn(70000, 64, device="cuda")
    big = torch.empty(70000, 64, device="cuda")
    small = torch.empty(100, 64, device="cuda")
    ops._gemm(x, 64, 0, W, 64, 1, None, big, 64, 70000, 64, 64)
    ops._gemm(x, 64, 0, W, 64, 1, None, small, 64, 100, 64, 64)
    assert torch.equal(big[:100], small)


@pytest.mark.parametrize("M,C", [(1000, 64), (50, 10), (4096, 512)])
def test_linear_bn_act_vs_torch(ops, M, C):
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, 32, generator=g).cuda().requires_grad_(True)
    lin = torch.nn.Linear(32, C).cuda()
    bn = torch.nn.BatchNorm1d(C).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
    bn2 = torch.nn.BatchNorm1d(C).cuda()
    bn2.load_state_dict(bn.state_dict())
    w = torch.randn(M, C, generator=g).cuda()
    for training in (True, False):
        bn.train(training); bn2.train(training)
        for p in (x, lin.weight, lin.bias, bn.weight, bn.bias):
            p.grad = None
        out = ops.linear_bn_act(x, lin.weight, lin.bias, bn, 0.2)
        (out * w).sum().backward()
        got = [out.detach().clone()] + [p.grad.clone() for p in (x, lin.weight, bn.weight, bn.bias)]
        gb = lin.bias.grad.clone()
        for p in (x, lin.weight, lin.bias, bn.weight, bn.bias):
            p.grad = None
        x2 = x.detach().double().requires_grad_(True)
        lw, lb = lin.weight.detach().double().requires_grad_(True), lin.bias.detach().double().requires_grad_(True)
        gw_, gb_ = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
        y = torch.nn.functional.linear(x2, lw, lb)
        rm = None if training else bn2.running_mean.double()
        rv = None if training else bn2.running_var.double()
        ref = torch.nn.functional.leaky_relu(torch.nn.functional.batch_norm(y, rm, rv, gw_, gb_, training, 0.1, bn.eps),
                                             0.2)
        (ref * w.double()).sum().backward()
        want = [ref.detach(), x2.grad, lw.grad, gw_.grad, gb_.grad]
        gscale = max(float(t.abs().max()) for t in want[1:])
        for a, b, name in zip(got, want, ("out", "gx", "gW", "ggamma", "gbeta")):
            lim = 1e-4 * max(1.0, float(b.abs().max()) if name == "out" else gscale)
            assert float((a.double() - b).abs().max()) < lim, (name, training)
        if not training:
            assert float((gb.double() - lb.grad).abs().max()) < 1e-4 * gscale
        if training:    # running statistics follow nn.BatchNorm1d
            bn2(torch.nn.functional.linear(x.detach(), lin.weight, lin.bias))
            assert torch.allclose(bn.running_mean, bn2.running_mean, atol=1e-5)
            assert torch.allclose(bn.running_var, bn2.running_var, atol=1e-5)
            assert int(bn.num_batches_tracked) == int(bn2.num_batches_tracked)


# ----------------------------------------------------------------- difference-wise attention
def _diffattn_torch(q, kv, idx):
    """fp64 restatement of reference modules/pointnet2_utils.py:558-569 on gathered rows."""
    B, S, C = q.shape
    k, v = kv[..., :C], kv[..., C:]
    bi = torch.arange(B, device=q.device)[:, None, None]
    gk, gv = k[bi, idx], v[bi, idx]                       # [B,S,K,C]
    e = (q[:, :, None, :] - gk) / (C ** 0.5)
    a = torch.softmax(e, dim=2)
    w = a - a.sum(dim=2, keepdim=True)
    return (w * gv).max(dim=2)[0]


def _no_workspace(real_empty):
    """torch.empty that hands the backward a 0-byte workspace (forces the atomic fallback)."""
    def empty(*size, **kw):
        if kw.get("dtype") is torch.uint8 and len(size) == 1 and isinstance(size[0], int) and size[0] > 4096:
            return real_empty(0, **kw)
        return real_empty(*size, **kw)
    return empty


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,S,C,K,path", [
    (8, 256, 128, 64, 8, "csr"),
    (4, 96, 96, 128, 8, "csr"),         # ragged row counts
    (16, 64, 32, 512, 8, "csr"),        # two channel rounds per lane
    (8, 2048, 64, 64, 8, "csr"),        # most base rows have no entry
    (3, 50, 20, 24, 5, "csr"),          # generic K, float4 lanes over 24 channels
    (2, 40, 30, 13, 8, "csr"),          # C % 4 != 0 -> scalar lanes
    (2, 5000, 16, 64, 8, "csr"),        # N > 1024 threads * 4
    (2, 40, 30, 12, 8, "atomic"),       # no workspace -> global atomics (entry point clears outputs)
    (1, 13000, 16, 64, 8, "atomic"),    # a large cloud through the atomic path
    (1, 70000, 16, 64, 8, "atomic"),    # N beyond the inverted table's row limit -> size query returns 0
])
def test_diffattn_forward_backward(ops, monkeypatch, B, N, S, C, K, path):
    g = torch.Generator().manual_seed(B * 1000 + N + C)
    q = torch.randn(B, S, C, generator=g).cuda().requires_grad_()
    kv = torch.randn(B, N, 2 * C, generator=g).cuda().requires_grad_()
    idx = torch.randint(0, N, (B, S, K), generator=g).cuda()
    idx[:, 0, :] = idx[:, 0, :1]                          # a point whose neighbours all coincide
    idx[:, 1:, 0] = 3                                     # a hub row: S-1 entries (> the sorted-list cap)
    go = torch.randn(B, S, C, generator=g).cuda()
    from mpa_amd._lib import lib
    need = int(lib.mpa_diffattn_bwd_workspace_bytes(B, N, S, K, C))
    assert (need > 0) == (N <= 65536)
    if path == "atomic" and need:
        monkeypatch.setattr(torch, "empty", _no_workspace(torch.empty))
    out = ops.diffattn(q, kv, idx)
    out.backward(go)
    q64 = q.detach().double().requires_grad_()
    kv64 = kv.detach().double().requires_grad_()
    ref = _diffattn_torch(q64, kv64, idx)
    ref.backward(go.double())
    assert (out.double() - ref).abs().max().item() < 1e-5
    # gradients: same arg-max wherever the max is not a near-tie, so compare with a tolerance on
    # all but the handful of elements whose top two candidates are within fp32 noise
    # (grad_q is identically zero in exact arithmetic -- the energies are shift-invariant in q --
    #  so it is held to the key gradients' scale)
    scale = kv64.grad.abs().max().item()
    for got, want in ((q.grad, q64.grad), (kv.grad, kv64.grad)):
        err = (got.double() - want).abs()
        assert (err > 1e-4 * scale).float().mean().item() < 1e-4
        assert torch.isfinite(got).all()


@pytest.mark.parametrize("B,fN,fS,N,S", [(4, 1024, 512, 1024, 1024), (3, 512, 256, 1024, 512), (2, 256, 128, 512, 256),
                                         (2, 2048, 1024, 2048, 2048), (2, 64, 32, 128, 64), (5, 300, 100, 300, 300)])
def test_fused_fps_knn_xyz_equals_separate_launches(ops, B, fN, fS, N, S):
    """mpa_fps_knn_xyz_f32 (one launch: sampling + xyz kNN side by side) == the two entry points."""
    base = unit_cloud(B, N, seed=fN + S)
    query = base[:, :S].contiguous()
    fin = (query if fN == S else unit_cloud(B, fN, seed=7))
    start = torch.randint(0, fN, (B,), generator=torch.Generator().manual_seed(1))
    base, query, fin = base.cuda(), query.cuda(), fin.cuda()
    fidx, fxyz, dist, idx = ops.fps_and_knn_xyz(fin, fS, 8, base, query, start_idx=start)
    fidx0, fxyz0 = ops.farthest_point_sample(fin, fS, start_idx=start, return_xyz=True)
    dist0, idx0 = ops.knn_point(8, base, query)
    assert torch.equal(fidx, fidx0) and torch.equal(fxyz, fxyz0)
    assert torch.equal(idx, idx0) and torch.equal(dist, dist0)          # same bits: fp32 compared exactly


# --------------------------------------------------------------------------- FPS on rows of any width
@pytest.mark.parametrize("tag", ["fpsc_2", "fpsc_6", "fpsc_10", "fpsc_64"])
def test_fps_any_channel_count_golden(ops, golden_round2, tag):
    """reference farthest_point_sample on [B,N,C], C != 3 (modules/pointnet2_utils.py:84-109): bit-exact indices."""
    g = golden_round2
    S = g[tag + "/idx"].shape[1]
    idx, sub = ops.farthest_point_sample(G(g[tag + "/x"]), S, start_idx=torch.from_numpy(g[tag + "/start"]),
                                         return_xyz=True)
    assert np.array_equal(idx.cpu().numpy(), g[tag + "/idx"].astype(np.int64))
    want = np.take_along_axis(g[tag + "/x"], g[tag + "/idx"].astype(np.int64)[..., None], 1)
    assert np.array_equal(sub.cpu().numpy(), want)


@pytest.mark.parametrize("B,N,C,S", [(3, 1000, 6, 400), (2, 513, 16, 513), (2, 64, 128, 20), (1, 12000, 5, 64), (2, 33, 9, 40)])
def test_fps_any_channel_count_oracle(ops, co, B, N, C, S):
    x = randn((B, N, C), seed=N + C)
    x[0, 5] = x[0, 2]                                    # a duplicated row: ties -> lowest index
    start = torch.randint(0, N, (B,), generator=torch.Generator().manual_seed(C))
    idx = ops.farthest_point_sample(x.cuda(), S, start_idx=start)
    assert np.array_equal(idx.cpu().numpy(), co.farthest_point_sample(x.numpy(), S, start.numpy()))


# --------------------------------------------------------------------------- grouped weight gradients
def test_grouped_weight_gradients_vs_torch(ops):
    """mpa_gemm_tn_grouped_f32 through ops' queue: out_p = A_p^T B_p (+ column sums of A_p) for a mix of
    products -- 128 x 128-tile ones (M, N multiples of 128; split over K and not), 64 x 64-tile ones,
    streamed single-tile ones, ragged ones -- against torch in float64."""
    g = torch.Generator().manual_seed(77)
    shapes = [(512, 256, 4096), (256, 128, 8192), (128, 128, 512), (1024, 512, 2048), (128, 256, 300), (64, 64, 16384),
              (128, 64, 4096), (40, 256, 64), (64, 3, 8192), (256, 384, 1024)]
    queue, want = [], []
    for M, N, K in shapes:
        gy = (torch.randint(-4, 5, (K, M), generator=g).float() * 0.25).cuda()
        x = (torch.randint(-4, 5, (K, N), generator=g).float() * 0.5).cuda()
        out = torch.full((M, N), float("nan"), device="cuda")
        acs = torch.zeros(M, device="cuda")
        queue.append((gy, M, x, N, out, M, N, K, acs))
        want.append((gy.double().t() @ x.double(), gy.double().sum(0)))
    ops.defer_weight_grads(True)
    try:
        ops._DW_QUEUE.extend(queue)
        ops.flush_weight_grads()
    finally:
        ops.defer_weight_grads(False)
    torch.cuda.synchronize()
    for (M, N, K), q, (w, ws) in zip(shapes, queue, want):
        # operands are multiples of 1/4 and 1/2 with small magnitudes: every partial sum is exact in fp32
        assert torch.equal(q[4].double(), w), (M, N, K)
        assert torch.equal(q[8].double(), ws), (M, N, K)


@pytest.mark.gpu
@pytest.mark.parametrize("B,S,N,C", [(32, 1024, 1000, 64), (40, 800, 333, 32), (2, 96, 130, 128), (3, 64, 77, 256)])
def test_knn_wide_groups_and_given_norms_vs_oracle(ops, co, B, S, N, C):
    """The two fast forms of the MFMA search against the C oracle, bit for bit: (i) >= 1024 workgroups of 32 queries
    -> two query groups per workgroup (C = 32 / 64), (ii) base-row norms computed once by mpa_row_norms_f32 and handed
    to mpa_knn_norms_f32 (what ops.knn_point does for C in {32,64,128,256}); ragged N (padding rows carry +inf)."""
    from mpa_amd._lib import lib, check
    base = randn((B, N, C), seed=N + C).cuda()
    query = randn((B, S, C), seed=S + C, scale=0.7).cuda()
    K = 8
    od, oi = co.knn_point(K, base.cpu().numpy(), query.cpu().numpy())
    dist, idx = ops.knn_point(K, base, query)                       # (ii) through the op
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(bits(dist.cpu().numpy()), bits(od))
    dist1 = torch.empty_like(dist)
    idx1 = torch.empty_like(idx)
    st = torch.cuda.current_stream().cuda_stream
    check(lib.mpa_knn_f32(base.data_ptr(), query.data_ptr(), B, N, S, C, K, dist1.data_ptr(), idx1.data_ptr(), st),
          "mpa_knn_f32")                                            # norms recomputed inside the search
    assert torch.equal(idx1, idx) and torch.equal(dist1, dist)
    norms = torch.empty(B, (N + 31) // 32 * 32, device="cuda")
    check(lib.mpa_row_norms_f32(base.data_ptr(), B, N, C, norms.data_ptr(), st), "mpa_row_norms_f32")
    want = torch.from_numpy(co.row_norms(base.cpu().numpy())) if hasattr(co, "row_norms") else None
    assert torch.isinf(norms[:, N:]).all()
    if want is not None:
        assert np.array_equal(bits(norms[:, :N].cpu().numpy()), bits(want.numpy()))


@pytest.mark.gpu
def test_xyz_knn_memo_hits_only_on_unchanged_coordinates(ops):
    """Coordinate searches are remembered per (base, query) tensor pair: the same tensors return the same result
    object, an in-place edit or another tensor of equal content searches again (and still agrees)."""
    xyz = unit_cloud(2, 256, seed=5).cuda()
    q = xyz[:, ::4].contiguous()
    d0, i0 = ops.knn_point(8, xyz, q)
    d1, i1 = ops.knn_point(8, xyz, q)
    assert i1 is i0 and d1 is d0                                   # remembered
    d2, i2 = ops.knn_point(8, xyz.clone(), q)                      # other storage, same content: fresh search
    assert i2 is not i0 and torch.equal(i2, i0) and torch.equal(d2, d0)
    xyz.mul_(2.0)                                                  # in-place edit: the version moves on
    d3, i3 = ops.knn_point(8, xyz, q)
    assert i3 is not i0
    want_d, want_i = ops.knn_point(8, xyz.clone(), q)
    assert torch.equal(i3, want_i) and torch.equal(d3, want_d)


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,C,dtype", [(2, 2048, 64, torch.float32), (3, 130, 20, torch.float32), (1, 1, 7, torch.float32),
                                         (2, 512, 256, torch.bfloat16), (2, 77, 24, torch.bfloat16)])
def test_max_over_points_matches_torch(ops, B, N, C, dtype):
    """ops.max_over_points == x.max(dim=1, keepdim=True)[0]: values exactly, the gradient routed to the first row
    attaining the maximum (ties: lowest row), zeros elsewhere."""
    g = torch.Generator().manual_seed(N + C)
    x = torch.randn(B, N, C, generator=g).to(dtype)
    if N > 4:
        x[:, 3] = x.max(dim=1)[0]                     # a tie between row 3 and the original arg-max row of every column
    x = x.cuda().requires_grad_(True)
    out = ops.max_over_points(x)
    want_v, want_i = x.detach().float().cpu().max(dim=1, keepdim=True)
    assert out.shape == (B, 1, C) and out.dtype == dtype
    assert torch.equal(out.float().cpu(), want_v)
    w = torch.randn(B, 1, C, generator=g).to(dtype).cuda()
    (out * w).sum().backward()
    xc = x.detach().float().cpu()
    first = (xc == want_v).float().argmax(dim=1, keepdim=True)          # first row equal to the maximum
    want_g = torch.zeros(B, N, C).scatter_(1, first, w.float().cpu())
    assert torch.equal(x.grad.float().cpu(), want_g)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_max_over_points_propagates_nan(ops, dtype):
    """x.max(dim=1) propagates NaN (reference modules/pointnet2_utils.py:846-850): a column holding a NaN returns NaN
    with the gradient routed to its first NaN row, an all-NaN column likewise; clean columns are untouched -- a
    diverged run must show up in the loss instead of as finite global features."""
    B, N, C = 2, 700, 24
    x = torch.randn(B, N, C, generator=torch.Generator().manual_seed(1)).to(dtype)
    x[0, 650, 3] = float("nan")            # one NaN late in a column
    x[0, 10, 3] = float("nan")             # ... and an earlier one: the first NaN row is the arg
    x[1, :, 7] = float("nan")              # an all-NaN column
    x = x.cuda().requires_grad_(True)
    out = ops.max_over_points(x)
    want = x.detach().float().cpu().max(dim=1, keepdim=True)[0]
    assert torch.equal(torch.isnan(out.float().cpu()), torch.isnan(want))
    ok = ~torch.isnan(want)
    assert torch.equal(out.float().cpu()[ok], want[ok])
    out.backward(torch.ones_like(out))
    g = x.grad.float().cpu()
    assert g[0, 10, 3] == 1 and g[0, :, 3].sum() == 1 and g[1, 0, 7] == 1 and g[1, :, 7].sum() == 1


@pytest.mark.gpu
@pytest.mark.parametrize("B,fN,fS,N,S,C,with_xyz", [(4, 512, 256, 1024, 512, 64, True), (40, 1024, 512, 2048, 1024, 64, True),
                                                     (3, 300, 77, 150, 90, 128, False), (2, 2048, 1000, 333, 333, 64, True),
                                                     (2, 100, 50, 200, 100, 64, True), (2, 512, 256, 300, 200, 32, True)])
def test_fused_fps_and_searches_equal_separate_launches(ops, B, fN, fS, N, S, C, with_xyz):
    """ops.fps_knn_fused (mpa_fps_knn_feat_f32: next state's FPS + this state's coordinate search + its feature
    search in one launch; the last two rows fall back to the separate launches) == the three entry points."""
    fin = unit_cloud(B, fN, seed=fN).cuda()
    xb = unit_cloud(B, N, seed=N + 1).cuda()
    xq = xb[:, :S].contiguous()
    fb = randn((B, N, C), seed=C + N).cuda()
    fq = randn((B, S, C), seed=C + S, scale=0.7).cuda()
    start = torch.arange(B) % fN
    fidx, fxyz, rx, (df, jf) = ops.fps_knn_fused(fin, fS, 8, xb if with_xyz else None, xq if with_xyz else None, 8, fb, fq,
                                                  start_idx=start)
    fidx0, fxyz0 = ops.farthest_point_sample(fin, fS, start_idx=start, return_xyz=True)
    assert torch.equal(fidx, fidx0) and torch.equal(fxyz, fxyz0)
    df0, jf0 = ops.knn_point(8, fb, fq)
    assert torch.equal(jf, jf0) and torch.equal(df, df0)
    if with_xyz:
        dx0, ix0 = ops.knn_point(8, xb.clone(), xq)
        assert torch.equal(rx[1], ix0) and torch.equal(rx[0], dx0)
    else:
        assert rx is None


def test_tiled_gemm_exact_on_small_integers_repeatedly(ops):
    """The LDS-tiled GEMM's software-pipelined loop (>= 4 K-slabs: straight-line prologue / steady loop / tail, operand reads
    and the next slab's LDS writes interleaved with the MFMAs) on operands whose every partial sum is exact in fp32, forward
    (x W^T) and input gradient (dy W), 2 and 3 tail slabs, whole and ragged tiles, repeated: a missing barrier between the
    last multiply and the epilogue that reuses the slab buffers showed up as a few wrong elements in one run out of several."""
    g = torch.Generator().manual_seed(5)
    for M, N, K in [(2048, 512, 1024), (4096, 256, 320), (1024, 1024, 256), (520, 136, 448), (64, 2048, 1024)]:
        x = (torch.randint(-4, 5, (M, K), generator=g).float() * 0.25).cuda().requires_grad_(True)
        w = (torch.randint(-4, 5, (N, K), generator=g).float() * 0.5).cuda()
        dy = (torch.randint(-4, 5, (M, N), generator=g).float() * 0.5).cuda()
        want = x.detach().double() @ w.double().t()
        want_dx = dy.double() @ w.double()
        for rep in range(6):
            y = ops.linear(x, w, None)
            (dx,) = torch.autograd.grad(y, x, dy)
            assert torch.equal(y.double(), want), (M, N, K, rep, "forward")
            assert torch.equal(dx.double(), want_dx), (M, N, K, rep, "dX")
