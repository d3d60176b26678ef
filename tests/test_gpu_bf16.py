"""GPU tests of the bf16 feature path (BASELINE configs 3 and 5; SURVEY.md 8d: features and GEMM operands
bf16 with fp32 accumulation; coordinates, distances, indices, BatchNorm statistics, parameters fp32).

PARITY UNPINNED for bf16 numerics: the reference has no reduced-precision mode (no autocast, no
GradScaler), so there are no reference bf16 outputs to pin to.  What IS pinned:
  * every bf16 kernel against the SAME formula evaluated in fp32 / fp64 on the bf16-rounded inputs -- the
    element-wise kernels must equal the fp32 kernels' results rounded once (bit for bit), the MFMA products
    must equal an fp64 product of the rounded operands to accumulation-order accuracy (this is what catches a
    wrong fragment / transposed-read layout: every output element is checked, operands are asymmetric);
  * blocks and whole models in bf16 against the fp32 reference fixtures with a stated tolerance (relative L2
    error and arg-max agreement), which bounds the cost of the precision, not the correctness of a kernel.
"""
import ctypes
from argparse import Namespace

import numpy as np
import pytest
import torch

from param_fill import fill_state, randn, unit_cloud

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    import mpa_amd
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return mpa_amd.ops


@pytest.fixture(scope="module")
def lib():
    from mpa_amd import _lib
    return _lib.lib


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def GL(a):
    return torch.from_numpy(np.ascontiguousarray(a).astype(np.int64)).cuda()


def rel_l2(a, b):
    a, b = a.detach().double().flatten(), b.detach().double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


# ------------------------------------------------------------------------------------------ MFMA products
GEMM_SHAPES = [(256, 64, 64), (384, 128, 128), (200, 50, 96), (128, 192, 896), (130, 70, 40), (64, 64, 3),
               (1000, 256, 512), (2048, 64, 64), (4096, 320, 128), (37, 40, 1024), (8192, 128, 64)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("b_f32", [True, False])
def test_gemm_bf16_forward_and_dx(lib, M, N, K, b_f32):
    """C = A W^T + b (transB = 1) and dX = G W (transB = 0) on v_mfma_f32_32x32x16_bf16, every element against an
    fp64 product of the bf16-rounded operands; fp32 output (unrounded accumulators) to 1e-5 of the row scale,
    bf16 output = that fp32 output rounded once; BatchNorm tile statistics of the unrounded results."""
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g).to(BF).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5)
    bias = torch.randn(N, generator=g).cuda()
    Wd = W.cuda() if b_f32 else W.to(BF).cuda()
    Wr = W.to(BF).double().cuda()                      # what the kernel multiplies with, either way
    ref = A.double() @ Wr.t() + bias.double()
    scale = float(ref.abs().max())
    C32 = torch.full((M, N), float("nan"), device="cuda")
    stats = torch.full(((M + 63) // 64, 2, N), float("nan"), device="cuda")
    rc = lib.mpa_gemm_bf16(p(A), K, p(Wd), K, 1, int(b_f32), p(bias), p(C32), N, 1, M, N, K, p(stats), 0, None)
    assert rc == 0
    torch.cuda.synchronize()
    assert float((C32.double() - ref).abs().max()) <= 2e-6 * scale * max(1.0, K ** 0.5 / 8), "fp32 output"
    C16 = torch.zeros(M, N, dtype=BF, device="cuda")
    assert lib.mpa_gemm_bf16(p(A), K, p(Wd), K, 1, int(b_f32), p(bias), p(C16), N, 0, M, N, K, None, 0, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(C16, C32.to(BF)), "bf16 output is the fp32 result rounded once"
    # tile statistics: per 64-row tile the sum and the sum of squared deviations from the tile mean
    for t in range((M + 63) // 64):
        blk = C32[t * 64:(t + 1) * 64].double()
        s, m2 = blk.sum(0), ((blk - blk.mean(0)) ** 2).sum(0)
        assert torch.allclose(stats[t, 0].double(), s, rtol=1e-5, atol=1e-4 * scale)
        assert torch.allclose(stats[t, 1].double(), m2, rtol=1e-4, atol=1e-4 * scale * scale)
    # dX = G [M,N] * W [N][K]: the weight walked along its rows (reduction over N)
    Gy = torch.randn(M, N, generator=g).to(BF).cuda()
    refx = Gy.double() @ Wr
    X32 = torch.full((M, K), float("nan"), device="cuda")
    assert lib.mpa_gemm_bf16(p(Gy), N, p(Wd), K, 0, int(b_f32), None, p(X32), K, 1, M, K, N, None, 0, None) == 0
    torch.cuda.synchronize()
    assert float((X32.double() - refx).abs().max()) <= 2e-6 * float(refx.abs().max()) * max(1.0, N ** 0.5 / 8), "dX"
    if K % 2 == 0:
        X16 = torch.zeros(M, K, dtype=BF, device="cuda")
        assert lib.mpa_gemm_bf16(p(Gy), N, p(Wd), K, 0, int(b_f32), None, p(X16), K, 0, M, K, N, None, 0, None) == 0
        torch.cuda.synchronize()
        assert torch.equal(X16, X32.to(BF))


def test_gemm_bf16_strided_rows(lib):
    """leading dimensions larger than the row (a column block of a wider tensor is read / written in place)"""
    g = torch.Generator().manual_seed(3)
    M, N, K = 320, 64, 128
    Awide = torch.randn(M, 3 * K, generator=g).to(BF).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    Cwide = torch.zeros(M, 2 * N, dtype=BF, device="cuda")
    A = Awide[:, K:2 * K]
    rc = lib.mpa_gemm_bf16(ctypes.c_void_p(A.data_ptr()), 3 * K, p(W), K, 1, 1, None,
                           ctypes.c_void_p(Cwide.data_ptr() + 2 * N), 2 * N, 0, M, N, K, None, 0, None)
    assert rc == 0
    torch.cuda.synchronize()
    ref = (A.double() @ W.to(BF).double().t())
    assert rel_l2(Cwide[:, N:], ref) < 4e-3 and float(Cwide[:, :N].abs().max()) == 0.0


@pytest.mark.parametrize("K,M,N", [(4096, 64, 64), (8192, 128, 64), (2048, 50, 128), (1000, 64, 3), (16384, 256, 192),
                                   (640, 96, 896), (65536, 64, 64)])
def test_grouped_weight_gradients_bf16(ops, K, M, N):
    """out[M,N] = A^T B with both operands row = reduction index (dW = dY^T X): transposed LDS reads feeding the
    MFMA, split-K slabs, column sums of A (the bias gradient) -- against fp64 on the rounded operands."""
    g = torch.Generator().manual_seed(K + M + N)
    A = torch.randn(K, M, generator=g).to(BF).cuda()
    Bm = torch.randn(K, N, generator=g).to(BF).cuda()
    out = torch.full((M, N), float("nan"), device="cuda")
    acs = torch.zeros(M, device="cuda")
    ops._weight_grads_bf16([(A, M, Bm, N, out, M, N, K, acs)])
    torch.cuda.synchronize()
    ref = A.double().t() @ Bm.double()
    assert float((out.double() - ref).abs().max()) <= 3e-6 * float(ref.abs().max()) * max(1.0, (K / 64) ** 0.5)
    assert torch.allclose(acs.double(), A.double().sum(0), rtol=1e-4, atol=1e-3 * K ** 0.5)
    # several problems of different shapes in one launch, one of them on column blocks of wider tensors
    A2 = torch.randn(3000, 200, generator=g).to(BF).cuda()
    B2 = torch.randn(3000, 72, generator=g).to(BF).cuda()
    o1 = torch.empty(M, N, device="cuda")
    o2 = torch.empty(64, 72, device="cuda")
    ops._weight_grads_bf16([(A, M, Bm, N, o1, M, N, K, None), (A2[:, 64:128], 200, B2, 72, o2, 64, 72, 3000, None)])
    torch.cuda.synchronize()
    assert torch.allclose(o1, out, rtol=1e-5, atol=1e-5 * float(ref.abs().max()))        # (split-K sums: order varies)
    ref2 = A2[:, 64:128].double().t() @ B2.double()
    assert float((o2.double() - ref2).abs().max()) <= 3e-6 * float(ref2.abs().max()) * 8


# ------------------------------------------------------------------------------------------ element-wise kernels
def test_bn_act_bf16_equals_fp32_kernels_rounded(ops, lib):
    """BatchNorm + LeakyReLU (+ residual) forward / backward on bf16 rows = the fp32 kernels on the same
    (rounded) inputs, results rounded once."""
    g = torch.Generator().manual_seed(0)
    M, C = 1000, 96
    x = torch.randn(M, C, generator=g).to(BF).cuda()
    res = torch.randn(M, C, generator=g).to(BF).cuda()
    save = torch.stack([torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5]).cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.1).cuda()
    y16 = torch.empty(M, C, dtype=BF, device="cuda")
    y32 = torch.empty(M, C, device="cuda")
    assert lib.mpa_bn_act_fwd_bf16(p(x), p(save), p(gamma), p(beta), p(res), ctypes.c_float(0.2), M, C, p(y16), None) == 0
    xf, rf = x.float(), res.float()
    assert lib.mpa_bn_act_fwd_f32(p(xf), p(save), p(gamma), p(beta), p(rf), ctypes.c_float(0.2), M, C, p(y32), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(y16, y32.to(BF))
    gy = torch.randn(M, C, generator=g).to(BF).cuda()
    parts = []
    for sfx, xx, gg in (("bf16", x, gy), ("f32", xf, gy.float())):
        part = torch.zeros(8, 2, C, device="cuda")
        fn = getattr(lib, "mpa_bn_act_bwd_reduce_" + sfx)
        assert fn(p(xx), p(gg), p(save[0]), p(save[1]), p(gamma), p(beta), ctypes.c_float(0.2), M, C, C, p(part), 8, None) == 0
        gx = torch.empty_like(xx)
        dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        fn = getattr(lib, "mpa_bn_act_bwd_apply_" + sfx)
        assert fn(p(xx), p(gg), p(save[0]), p(save[1]), p(gamma), p(beta), p(part), 8, ctypes.c_float(0.2), 1, M, C, C,
                  p(gx), p(dg), p(db), None) == 0
        torch.cuda.synchronize()
        parts.append((part.sum(0), gx, dg, db))
    assert torch.allclose(parts[0][0], parts[1][0], rtol=1e-5, atol=1e-4)          # atomics: order differs
    assert rel_l2(parts[0][1].float(), parts[1][1]) < 3e-3                         # one rounding (+ ulp-level sums)
    assert torch.allclose(parts[0][2], parts[1][2], rtol=1e-4, atol=1e-3)


def test_gather_and_upsample_bf16(ops):
    g = torch.Generator().manual_seed(1)
    B, N, S, C, K = 3, 200, 100, 72, 8
    f = torch.randn(B, N, C, generator=g).to(BF).cuda().requires_grad_(True)
    idx = torch.randint(0, N, (B, S, K), generator=g).cuda()
    out = ops.index_points(f, idx)
    assert out.dtype == BF and torch.equal(out, torch.gather(f.detach(), 1, idx.view(B, -1, 1).expand(-1, -1, C)).view(B, S, K, C))
    w = torch.randn(B, S, K, C, generator=g).to(BF).cuda()
    (out.float() * w.float()).sum().backward()
    want = torch.zeros(B, N, C, device="cuda").index_put_(
        (torch.arange(B, device="cuda").view(B, 1, 1).expand_as(idx).reshape(-1), idx.reshape(-1)), w.float().view(-1, C),
        accumulate=True)
    assert f.grad.dtype == BF and rel_l2(f.grad.float(), want) < 3e-3
    # upsample: bf16 rows = the fp32 op on the rounded rows, rounded once (sum and division in fp32)
    pts = torch.randn(B, S, C, generator=g).to(BF)
    pts[:, ::7, 0] = 0
    pts = pts.cuda().requires_grad_(True)
    kidx = torch.randint(0, 2 * S, (B, S, K), generator=g).cuda()
    up16 = ops.upsample(pts, kidx)
    p32 = pts.detach().float().requires_grad_(True)
    up32 = ops.upsample(p32, kidx)
    assert up16.dtype == BF and torch.equal(up16, up32.to(BF))
    gg = torch.randn(B, 2 * S, C, generator=g).to(BF).cuda()
    up16.backward(gg)
    up32.backward(gg.float())
    assert torch.equal(pts.grad, p32.grad.to(BF))


@pytest.mark.parametrize("B,N,S,C,K", [(2, 256, 128, 64, 8), (1, 100, 100, 128, 8), (2, 64, 32, 256, 5), (1, 300, 77, 20, 8)])
def test_diffattn_bf16(ops, B, N, S, C, K):
    """difference-wise attention on bf16 q / k / v: forward = the fp32 kernel on the rounded inputs, rounded once
    (same arg-max); backward within bf16 rounding of the fp32 kernel's gradients (per-slot key gradients are
    stored bf16 before the per-row sum)."""
    g = torch.Generator().manual_seed(B * N + C)
    q = torch.randn(B, S, C, generator=g).to(BF).cuda().requires_grad_(True)
    kv = torch.randn(B, N, 2 * C, generator=g).to(BF).cuda().requires_grad_(True)
    idx = torch.randint(0, N, (B, S, K), generator=g).cuda()
    q32 = q.detach().float().requires_grad_(True)
    kv32 = kv.detach().float().requires_grad_(True)
    o16 = ops.diffattn(q, kv, idx)
    o32 = ops.diffattn(q32, kv32, idx)
    assert o16.dtype == BF and torch.equal(o16, o32.to(BF))
    w = torch.randn(B, S, C, generator=g).to(BF).cuda()
    o16.backward(w)
    o32.backward(w.float())
    assert q.grad.dtype == BF and kv.grad.dtype == BF
    assert rel_l2(q.grad.float(), q32.grad) < 4e-3
    assert rel_l2(kv.grad.float(), kv32.grad) < 8e-3
    # the paired form on stacked projections (LocalMerge's two feature streams)
    qq = torch.randn(B, S, 2 * C, generator=g).to(BF).cuda().requires_grad_(True)
    kvkv = torch.randn(B, N, 4 * C, generator=g).to(BF).cuda().requires_grad_(True)
    idx2 = torch.randint(0, N, (B, S, K), generator=g).cuda()
    c1, c2 = ops.diffattn_pair(qq, kvkv, idx, idx2)
    r1 = ops.diffattn(qq.detach()[..., :C].contiguous(), kvkv.detach()[..., :2 * C].contiguous(), idx)
    r2 = ops.diffattn(qq.detach()[..., C:].contiguous(), kvkv.detach()[..., 2 * C:].contiguous(), idx2)
    assert torch.equal(c1, r1) and torch.equal(c2, r2)
    (c1.float().sum() + (c2.float() * 2).sum()).backward()
    assert torch.isfinite(qq.grad.float()).all() and torch.isfinite(kvkv.grad.float()).all()


def test_diffattn_xyz_bf16(ops):
    g = torch.Generator().manual_seed(4)
    B, N, S, C, K = 2, 256, 128, 64, 8
    xyz = unit_cloud(B, N, seed=1).cuda()
    ctr = xyz[:, :S].contiguous()
    idx = torch.randint(0, N, (B, S, K), generator=g).cuda()
    Ws = [(torch.randn(C, 3, generator=g) * 0.5).cuda().requires_grad_(True) for _ in range(3)]
    bs = [(torch.randn(C, generator=g) * 0.1).cuda().requires_grad_(True) for _ in range(3)]
    args = (xyz, ctr, idx, Ws[0], bs[0], Ws[1], bs[1], Ws[2], bs[2])
    o32 = ops.diffattn_xyz(*args)
    with ops.feature_dtype(BF):
        o16 = ops.diffattn_xyz(*args)
    assert o32.dtype == torch.float32 and o16.dtype == BF and torch.equal(o16, o32.to(BF))
    w = torch.randn(B, S, C, generator=g).to(BF).cuda()
    g32 = torch.autograd.grad(o32, Ws + bs, w.float())
    g16 = torch.autograd.grad(o16, Ws + bs, w)
    for a, b in zip(g16, g32):
        assert a.dtype == torch.float32 and torch.allclose(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()))


# ------------------------------------------------------------------------------------------ the Linear unit and blocks
def test_three_interpolate_bf16_twin(ops, golden_blocks):
    """mpa_three_interp_{fwd,bwd}_bf16 (north_star's "interpolate" decoder on the bf16 feature stream, reference
    modules/pointnet2_utils.py:896-906): forward equals the fp32 kernel on the bf16-rounded rows, rounded once (bit for
    bit: same fp32 arithmetic); the backward sums in fp32 like its twin (float atomics: equal to accumulation order,
    then one rounding); PointNetFeaturePropagation stays on the bf16 stream end to end and matches the reference's
    fp32 fixture at the bf16 storage tolerance."""
    from mpa_amd.modules import pointnet2_utils as P
    B, N, S, D = 3, 700, 150, 48
    x1, x2 = unit_cloud(B, N, seed=1).cuda(), unit_cloud(B, S, seed=2).cuda()
    p16 = randn((B, S, D), seed=3).to(BF).cuda().requires_grad_(True)
    p32 = p16.detach().float().requires_grad_(True)
    o16, o32 = ops.three_interpolate(x1, x2, p16), ops.three_interpolate(x1, x2, p32)
    assert o16.dtype == BF and torch.equal(o16, o32.to(BF))
    w = randn((B, N, D), seed=4).to(BF).cuda()
    o16.backward(w)
    o32.backward(w.float())
    assert p16.grad.dtype == BF
    assert rel_l2(p16.grad.float(), p32.grad) < 4e-3                      # one bf16 rounding of equal fp32 sums
    assert (p16.grad.float() - p32.grad.to(BF).float()).abs().max() <= 2 ** -7 * p32.grad.abs().max()
    g = golden_blocks
    xyz, fps = G(g["geo/xyz"]), GL(g["geo/fps"])
    sub = P.index_points(xyz, fps)
    m = fill_state(P.PointNetFeaturePropagation(32, [48], act=True), seed=5).cuda().train()
    pts2 = G(g["fp/points2"]).to(BF).requires_grad_(True)
    out = m(xyz, sub, None, pts2)
    assert out.dtype == BF and rel_l2(out.float(), G(g["fp/out"])) < 1.5e-2
    out.backward(randn(out.shape, seed=4242).cuda().to(BF))
    assert pts2.grad.dtype == BF and rel_l2(pts2.grad.float(), G(g["fp/gpoints2"])) < 3e-2


def test_linear_unit_bf16_against_fp32_fixture(golden_blocks):
    """reference Linear (Linear -> BatchNorm1d over the rows -> LeakyReLU) in bf16 against the reference's own fp32
    outputs and gradients (tests/golden/blocks.npz): tolerance = the precision's cost, stated here: 1.5e-2 relative
    L2 (bf16 has 8 significant bits: 2^-9 per rounding, a handful of roundings per unit)."""
    import mpa_amd  # noqa: F401
    from mpa_amd.modules import pointnet2_utils as P
    g = golden_blocks
    for tag, ci, co_, act in (("lin_a", 64, 128, True), ("lin_c", 128, 64, False)):
        m = fill_state(P.Linear(ci, co_, bn=False, act=act), seed=1).cuda().train()
        x = G(g[tag + "/x"]).to(BF).requires_grad_(True)
        y = m(x)
        assert y.dtype == BF
        assert rel_l2(y.float(), G(g[tag + "/y_train"])) < 1.5e-2, tag
        y.backward(randn(y.shape, seed=4242).cuda().to(BF))
        assert x.grad.dtype == BF and rel_l2(x.grad.float(), G(g[tag + "/gx"])) < 3e-2, tag
        assert m.linear.weight.grad.dtype == torch.float32
        assert rel_l2(m.linear.weight.grad, G(g[tag + "/gw"])) < 3e-2, tag
        assert rel_l2(m.norm2.weight.grad, G(g[tag + "/ggamma"])) < 3e-2, tag
        m = fill_state(P.Linear(ci, co_, bn=False, act=act), seed=1).cuda().eval()       # fresh running statistics
        assert rel_l2(m(x.detach()).float(), G(g[tag + "/y_eval"])) < 1.5e-2, tag


def test_local_merge_bf16_against_fp32_fixture(golden_blocks):
    """LocalMerge (xyz + feature neighbourhoods, attention streams, fc2) in bf16 against the reference's fp32
    output with the reference's neighbourhoods; 3e-2 relative L2."""
    import mpa_amd  # noqa: F401
    from mpa_amd import ops
    from mpa_amd.modules import pointnet2_utils as P, repsurface_utils as RS
    g = golden_blocks
    xyz, fps = G(g["geo/xyz"]), GL(g["geo/fps"])
    sub = ops.index_points(xyz, fps)
    for tag, cls in (("lm_cls", RS.LocalMerge), ("lm_seg", P.LocalMerge)):
        with ops.feature_dtype(BF):
            m0 = fill_state(cls(32, 64, 8, usetanh=False, residual=True), seed=3).cuda().train()
            f0 = m0(xyz=xyz, base_xyz=xyz, normal=xyz)[0]
            assert f0.dtype == BF and rel_l2(f0.float(), G(g[tag + "/f0"])) < 2e-2, tag
            m1 = fill_state(cls(64, 64, 8, usetanh=False, residual=False), seed=4).cuda().train()
            feat = G(g[tag + "/f0"]).to(BF).requires_grad_(True)
            real = ops.knn_point
            forced = [GL(g[tag + "/idx1"]), GL(g[tag + "/idx1_feat"])]
            try:       # the reference's neighbourhoods (near-ties flip under bf16 rounding of the features)
                P.knn_point = RS.knn_point = lambda k, a, b: (None, forced.pop(0))
                f1 = m1(xyz=sub, base_xyz=xyz, normal=xyz, feature=feat, FPS_idx=fps)[0]
            finally:
                P.knn_point = RS.knn_point = real
            assert rel_l2(f1.float(), G(g[tag + "/f1"])) < 3e-2, tag
            f1.backward(randn(f1.shape, seed=4242).cuda().to(BF))
            # (the max over K re-routes gradient wherever bf16 rounding flips a near-tied selection: measured 5-9e-2)
            assert rel_l2(feat.grad.float(), G(g[tag + "/gfeat"])) < 0.15, tag
        # every parameter gradient of the block against the fp32 HIP path on the SAME bf16-rounded input and the same
        # neighbourhoods (not only gradient norms: a wrong bf16 dW of a single unit shows up here)
        m32 = fill_state(cls(64, 64, 8, usetanh=False, residual=False), seed=4).cuda().train()
        feat32 = G(g[tag + "/f0"]).to(BF).float().requires_grad_(True)
        forced = [GL(g[tag + "/idx1"]), GL(g[tag + "/idx1_feat"])]
        try:
            P.knn_point = RS.knn_point = lambda k, a, b: (None, forced.pop(0))
            f32 = m32(xyz=sub, base_xyz=xyz, normal=xyz, feature=feat32, FPS_idx=fps)[0]
        finally:
            P.knn_point = RS.knn_point = real
        f32.backward(randn(f32.shape, seed=4242).cuda().to(BF).float())
        assert rel_l2(f1.float(), f32) < 2e-2, tag
        assert rel_l2(feat.grad.float(), feat32.grad) < 0.12, tag
        # error of every parameter's gradient tensor relative to its own norm -- or, for the gradients that are
        # cancelling sums (q projections: softmax_j(q - k_j) does not depend on q at all, so dL/dWq is identically zero in
        # exact arithmetic and pure rounding noise in fp32; biases summed over all rows), to 5 % of the block's largest
        worst = {}
        top = max(float(p32.grad.double().norm()) for p32 in m32.parameters() if p32.grad is not None)
        for (n16, p16), (n32, p32) in zip(m1.named_parameters(), m32.named_parameters()):
            if p32.grad is None or float(p32.grad.abs().max()) == 0:
                continue
            assert p16.grad is not None and p16.grad.dtype == torch.float32, n16
            den = max(float(p32.grad.double().norm()), 0.05 * top)
            worst[n16] = float((p16.grad.double() - p32.grad.double()).norm()) / den
        print(tag, "bf16 vs fp32-HIP parameter gradients, worst:", sorted(worst.items(), key=lambda kv: -kv[1])[:4])
        assert max(worst.values()) < 0.12, (tag, sorted(worst.items(), key=lambda kv: -kv[1])[:4])


def test_partseg_model_bf16_against_fp32_fixture(golden_seg):
    """The part-segmentation model (BASELINE configs[2]'s model) in bf16 against the reference's fp32 logits on the
    golden clouds, neighbourhoods teacher-forced.  Measured on MI355X: eval (running statistics) relative L2 of the
    logits 6.6e-3, per-point arg-max agreement 95.4 % (random weights: many near-tied classes); train mode 8.4e-2 /
    89.6 % -- the fixture has B = 2, so the coarse states' BatchNorm statistics are taken over 256 rows and
    amplify any perturbation (the fp32 path needs 5x its eval tolerance there too).  Limits: eval 2e-2 / 93 %,
    train 0.15 / 85 %.  PARITY UNPINNED (no reference bf16 numerics): this bounds the precision's cost."""
    import mpa_amd  # noqa: F401
    from mpa_amd import ops
    from mpa_amd.models.repsurf.pointnet2_part_seg_msg import get_model
    from mpa_amd.modules import pointnet2_utils as P
    from test_gpu_blocks import _run_model
    g = golden_seg
    model = fill_state(get_model(50), seed=0).cuda()
    model.drop1.p = 0.0
    pts, lab = G(g["points"]), G(g["label"])
    for mode, prefix in (("eval", ""), ("train", "train_")):
        model.train(mode == "train")
        with ops.feature_dtype(BF), torch.set_grad_enabled(mode == "train"):
            out, forced = _run_model(g, model, lambda m: m(pts, lab)[0], [P], prefix)
        ref = G(g["out_" + mode])
        assert out.dtype == torch.float32                                   # logits leave the bf16 stream in fp32
        err = rel_l2(out, ref)
        agree = float((out.argmax(-1) == ref.argmax(-1)).float().mean())
        print("part-seg bf16 %s: rel L2 %.3e, arg-max agreement %.4f" % (mode, err, agree))
        lim_err, lim_agree = (2e-2, 0.93) if mode == "eval" else (0.15, 0.85)
        assert err < lim_err and agree > lim_agree, (mode, err, agree)
    (out * randn(out.shape, seed=31337).cuda()).sum().backward()
    # gradient norms per parameter: within 25 % of the reference's (+ a floor of 1e-3 of the largest norm: biases
    # whose exact gradient is zero carry rounding noise in the reference and exact zeros here), 5 % outliers
    # allowed (near-tied max / LeakyReLU selections re-route gradient: the fp32 path sees the same)
    names = list(g["grad_names"])
    gmax = float(g["grad_norms"].max())
    bad = []
    for n, prm in model.named_parameters():
        ref = float(g["grad_norms"][names.index(n)])
        if ref > 0:
            assert prm.grad is not None and prm.grad.dtype == torch.float32, n
            got = float(prm.grad.double().norm())
            if abs(got - ref) > 0.25 * ref + 1e-3 * gmax:
                bad.append((n, got, ref))
    assert len(bad) <= 0.05 * len(names), (len(bad), bad[:10])


def test_cls_model_bf16_trains():
    """The classification model in bf16 through the captured training step: finite, falling loss (B = 8)."""
    import mpa_amd  # noqa: F401
    from mpa_amd import ops
    from mpa_amd.models.repsurf.repsurf_ssg_umb import Model, SmoothClsLoss
    from mpa_amd.runtime import GraphedTrainStep
    dev = torch.device("cuda")
    x = unit_cloud(8, 1024, seed=3).transpose(1, 2).contiguous().to(dev)
    y = torch.arange(8, device=dev) % 40
    torch.manual_seed(0)
    model = Model(Namespace(num_point=1024, return_dist=True, cuda_ops=True, num_class=40)).to(dev).train()
    with ops.feature_dtype(BF):
        step = GraphedTrainStep(model, SmoothClsLoss(), (x, y), lr=1e-3)
        try:
            losses = [float(step(x, y)) for _ in range(12)]
        finally:
            step.close()
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


@pytest.mark.parametrize("dtype", [torch.float32, BF])
def test_cat_broadcast(ops, dtype):
    """per-cloud rows broadcast next to per-point features: forward = torch.cat((a, rows.expand)), backward's
    per-cloud column sums from mpa_group_col_sum (fp32 accumulation)."""
    g = torch.Generator().manual_seed(5)
    B, N, Ca, Cr = 3, 2048, 256, 640
    a = torch.randn(B, N, Ca, generator=g).to(dtype).cuda().requires_grad_(True)
    rows = torch.randn(B, 1, Cr, generator=g).to(dtype).cuda().requires_grad_(True)
    out = ops.cat_broadcast(a, rows)
    assert torch.equal(out, torch.cat((a, rows.expand(-1, N, -1)), 2))
    w = torch.randn(B, N, Ca + Cr, generator=g).to(dtype).cuda()
    out.backward(w)
    assert torch.equal(a.grad, w[:, :, :Ca])
    want = w[:, :, Ca:].double().sum(1, keepdim=True)
    tol = 1e-5 if dtype == torch.float32 else 4e-3
    assert rows.grad.dtype == dtype and float((rows.grad.double() - want).abs().max()) <= tol * float(want.abs().max())


def test_graphed_partseg_step_bf16_replays_stay_finite():
    """The captured part-seg training step on bf16 features, replayed: finite, falling loss, and the same
    gradients on every replay of the same batch (regression: autograd's expand() backward -- a torch
    multi-workgroup sum -- produced NaN gradients from the second replay on; see ops._CatBroadcast)."""
    import mpa_amd  # noqa: F401
    import mpa_amd.runtime as rt
    from mpa_amd import ops
    from mpa_amd.models.repsurf.pointnet2_part_seg_msg import get_model, get_loss
    B, N = 4, 2048
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(B, 3, N, generator=g) * 2 - 1).to(dev)
    label = torch.zeros(B, 1, 16)
    label[:, 0, 3] = 1
    label = label.to(dev)
    target = torch.randint(0, 50, (B, N), generator=g).to(dev)
    torch.manual_seed(0)
    model = get_model(50).to(dev).train()

    def compute_loss(model, crit, x, label, target):
        pred, _ = model(x, label)
        return crit(pred.reshape(-1, 50), target.reshape(-1))

    with ops.feature_dtype(BF):
        step = rt.GraphedTrainStep(model, get_loss(), (x, label, target), lr=1e-3, compute_loss=compute_loss)
        try:
            losses = []
            for _ in range(8):
                losses.append(float(step(x, label, target).detach()))
                torch.cuda.synchronize()
                for prm in model.parameters():
                    assert prm.grad is None or torch.isfinite(prm.grad).all()
            assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
        finally:
            step.close()


@pytest.mark.gpu
@pytest.mark.parametrize("specs,M,chain", [
    (((64, 128, True), (64, 128, True), (64, 128, True)), 4096, False),
    (((32, 64, True), (40, 64, False)), 1000, False),
    (((64, 256, True), (64, 256, True), (128, 256, True), (64, 256, True)), 640, True),
])
def test_linear_unit_group_bf16_matches_single_units(specs, M, chain):
    """bf16 storage: the grouped launch runs the same tile code as the single-problem GEMM, so outputs and
    gradients agree with the one-by-one units to bf16 rounding of accumulated statistics."""
    import copy
    from mpa_amd import ops
    from mpa_amd.modules.pointnet2_utils import Linear, _unit_group
    torch.manual_seed(3)
    units = [Linear(k, n, bn=False, act=a).cuda() for k, n, a in specs]
    ref_units = copy.deepcopy(units)
    xs = [torch.randn(2, M // 2, k, device="cuda").bfloat16() for k, _, _ in specs]
    xs_g = [x.clone().requires_grad_(True) for x in xs]
    xs_r = [x.clone().requires_grad_(True) for x in xs]
    with ops.feature_dtype(torch.bfloat16):
        if chain:
            res = torch.randn(2, M // 2, specs[0][1], device="cuda").bfloat16()
            outs_g = [_unit_group(units, xs_g, residuals=[res] + [None] * (len(units) - 1), mode="chain")]
            acc = res
            for u, x in zip(ref_units, xs_r):
                acc = u.fused(x, acc)
            outs_r = [acc]
        else:
            outs_g = _unit_group(units, xs_g)
            outs_r = [u(x) for u, x in zip(ref_units, xs_r)]
        ws = [torch.randn_like(o) for o in outs_r]
        sum((o.float() * w.float()).sum() for o, w in zip(outs_g, ws)).backward()
        sum((o.float() * w.float()).sum() for o, w in zip(outs_r, ws)).backward()

    def close(a, b, what, tol):
        err = ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()
        assert err < tol, (what, err)

    for i, (o, r) in enumerate(zip(outs_g, outs_r)):
        assert o.dtype == torch.bfloat16
        close(o, r, "out%d" % i, 4e-3)
    for i, (u, r) in enumerate(zip(units, ref_units)):
        close(xs_g[i].grad, xs_r[i].grad, "dx%d" % i, 8e-3)
        close(u.linear.weight.grad, r.linear.weight.grad, "dW%d" % i, 8e-3)
        close(u.norm2.running_var, r.norm2.running_var, "rvar%d" % i, 1e-4)


@pytest.mark.gpu
def test_index_points_backward_bf16_scatters_into_bf16():
    """index_points backward on bf16 rows with an FPS-like map (every row listed once): the gradient is exactly the
    scattered rows (no fp32 staging / cast); with repeated rows the sums are bf16 sums of the listed rows."""
    from mpa_amd import ops
    torch.manual_seed(0)
    B, N, S, C = 3, 300, 120, 64
    x = torch.randn(B, N, C, device="cuda").bfloat16().requires_grad_(True)
    idx = torch.stack([torch.randperm(N)[:S] for _ in range(B)]).cuda()
    w = torch.randn(B, S, C, device="cuda").bfloat16()
    (ops.index_points(x, idx) * w).sum().backward()
    want = torch.zeros(B, N, C, device="cuda", dtype=torch.bfloat16)
    want.scatter_(1, idx.unsqueeze(-1).expand(-1, -1, C), w)
    assert torch.equal(x.grad, want)
    # repeated rows (S <= N but with duplicates): sum of the listed rows, within bf16 rounding of the partial sums
    idx2 = idx.clone()
    idx2[:, 1] = idx2[:, 0]
    x.grad = None
    (ops.index_points(x, idx2) * w).sum().backward()
    ref = torch.zeros(B, N, C, device="cuda").index_put_((torch.arange(B, device="cuda").view(B, 1).expand(B, S), idx2),
                                                        w.float(), accumulate=True)
    assert torch.allclose(x.grad.float(), ref, rtol=2e-2, atol=2e-2)


@pytest.mark.gpu
def test_gather_and_stack_bf16_matches_separate_ops():
    import copy
    from mpa_amd import ops
    torch.manual_seed(4)
    B, N, S, K = 2, 256, 100, 64
    layers = [torch.nn.Linear(K, n).cuda() for n in (64, 64, 64, 64)]
    ref_layers = copy.deepcopy(layers)
    x = torch.randn(B, N, K, device="cuda").bfloat16()
    idx = torch.stack([torch.randperm(N)[:S] for _ in range(B)]).cuda()
    xg, xr = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    zb = (True, False, True, False)
    with ops.feature_dtype(torch.bfloat16):
        fs_g, y_g = ops.gather_and_stack(xg, idx, layers, zb)
        fs_r, y_r = ops.index_points(xr, idx), ops.linear_stack(xr, ref_layers, zb)
        assert fs_g.dtype == torch.bfloat16 and torch.equal(fs_g, fs_r) and torch.equal(y_g, y_r)
        w1, w2 = torch.randn_like(fs_r), torch.randn_like(y_r)
        ((fs_g.float() * w1.float()).sum() + (y_g.float() * w2.float()).sum()).backward()
        ((fs_r.float() * w1.float()).sum() + (y_r.float() * w2.float()).sum()).backward()
    err = ((xg.grad.float() - xr.grad.float()).norm() / xr.grad.float().norm()).item()
    assert err < 4e-3, err                      # one bf16 rounding of the sum in a different place
    for a, b in zip(layers, ref_layers):
        assert torch.allclose(a.weight.grad, b.weight.grad, rtol=1e-3, atol=1e-3 * float(b.weight.grad.abs().max()))
