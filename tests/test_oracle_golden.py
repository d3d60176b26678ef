"""Pins the oracle (oracle/mpa_oracle.c and oracle/ref_cpu.py) to golden vectors that were
produced by importing the reference itself (tests/golden/make_golden.py).  CPU only."""
from argparse import Namespace

import numpy as np
import pytest
import torch

from conftest import bits
from oracle import c_oracle as co
from oracle import ref_cpu as R
from param_fill import fill_state, randn

TOL = 1e-4   # fp32 feature tolerance stated by BASELINE.json north_star


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def L(a):
    return torch.from_numpy(np.ascontiguousarray(a).astype(np.int64))


# ---------------------------------------------------------------- C oracle, bit-exact
@pytest.mark.parametrize("tag", ["fps_a", "fps_b", "fps_c", "fps_d"])
def test_c_fps(golden_index, tag):
    g = golden_index
    S = g[tag + "/idx"].shape[1]
    got = co.farthest_point_sample(g[tag + "/xyz"], S, g[tag + "/start"])
    assert np.array_equal(got, g[tag + "/idx"].astype(np.int64))


def test_c_fps_chain(golden_index):
    g = golden_index
    cur = g["fps_chain/xyz"]
    for lvl, S in enumerate((512, 256, 128, 64, 32)):
        idx = co.farthest_point_sample(cur, S, g["fps_chain/start%d" % lvl])
        assert np.array_equal(idx, g["fps_chain/idx%d" % lvl].astype(np.int64)), lvl
        cur = np.take_along_axis(cur, idx[..., None], axis=1)


@pytest.mark.parametrize("tag", ["knn_a", "knn_b", "knn_c", "knn_d", "knn_e", "knn_f", "knn_g", "knn_h"])
def test_c_knn(golden_index, tag):
    g = golden_index
    dist, idx = co.knn_point(8, g[tag + "/base"], g[tag + "/query"])
    assert np.array_equal(idx, g[tag + "/idx"].astype(np.int64))
    assert np.array_equal(bits(dist), bits(g[tag + "/dist"]))


def test_c_square_distance(golden_index):
    g = golden_index
    out = co.square_distance(g["sqd/src"], g["sqd/dst"])
    assert np.array_equal(bits(out), bits(g["sqd/out"]))


@pytest.mark.parametrize("tag", ["ball_a", "ball_b", "ball_c"])
def test_c_ball_query(golden_index, tag):
    g = golden_index
    idx = co.query_ball_point(float(g[tag + "/radius"]), 24, g[tag + "/base"], g[tag + "/query"])
    assert np.array_equal(idx, g[tag + "/idx"].astype(np.int64))


def test_c_three_nn(golden_index):
    g = golden_index
    dist, idx = co.three_nn(g["nn3/xyz1"], g["nn3/xyz2"])
    assert np.array_equal(idx, g["nn3/idx"].astype(np.int64))
    assert np.array_equal(bits(dist), bits(g["nn3/dist"]))


def test_c_knn_ties_lowest_index_first():
    base = np.zeros((1, 16, 3), np.float32)      # all points identical: every distance ties
    dist, idx = co.knn_point(8, base, base[:, :4])
    assert np.array_equal(idx[0, 0], np.arange(8))


# ---------------------------------------------------------------- torch restatement vs golden
def test_ref_cpu_index_ops(golden_index):
    g = golden_index
    for tag in ("knn_a", "knn_c", "knn_e"):
        dist, idx = R.knn_point(8, T(g[tag + "/base"]), T(g[tag + "/query"]))
        assert np.array_equal(idx.numpy(), g[tag + "/idx"].astype(np.int64))
    idx = R.farthest_point_sample(T(g["fps_a/xyz"]), 512, start_idx=L(g["fps_a/start"]))
    assert np.array_equal(idx.numpy(), g["fps_a/idx"].astype(np.int64))
    idx = R.query_ball_point(float(g["ball_b/radius"]), 24, T(g["ball_b/base"]), T(g["ball_b/query"]))
    assert np.array_equal(idx.numpy(), g["ball_b/idx"].astype(np.int64))


@pytest.mark.parametrize("tag,ci,co_,act", [("lin_a", 64, 128, True), ("lin_b", 3, 64, True), ("lin_c", 128, 64, False)])
def test_ref_cpu_linear(golden_blocks, tag, ci, co_, act):
    g = golden_blocks
    m = fill_state(R.Linear(ci, co_, bn=False, act=act), seed=1).train()
    x = T(g[tag + "/x"]).requires_grad_(True)
    y = m(x)
    assert np.abs(y.detach().numpy() - g[tag + "/y_train"]).max() < TOL
    y.backward(randn(y.shape, seed=4242))
    assert np.abs(x.grad.numpy() - g[tag + "/gx"]).max() < TOL
    assert np.abs(m.linear.weight.grad.numpy() - g[tag + "/gw"]).max() < TOL
    assert np.abs(m.norm2.running_var.numpy() - g[tag + "/running_var"]).max() < 1e-6
    m = fill_state(R.Linear(ci, co_, bn=False, act=act), seed=1).eval()
    assert np.abs(m(x).detach().numpy() - g[tag + "/y_eval"]).max() < TOL


LT_CASES = {"lt_xyz_self": (3, 64, True, False, True), "lt_xyz_fps": (3, 64, True, True, True),
            "lt_feat_id": (64, 64, False, True, False), "lt_feat_res": (64, 128, True, True, False),
            "lt_feat_self": (32, 32, False, False, False)}


@pytest.mark.parametrize("tag", sorted(LT_CASES))
def test_ref_cpu_local_trans(golden_blocks, tag):
    g = golden_blocks
    ci, co_, residual, use_fps, is_xyz = LT_CASES[tag]
    m = fill_state(R.LocalTrans(ci, co_, 8, residual=residual), seed=2).train()
    f = T(g[tag + "/f"]).requires_grad_(True)
    idx = L(g["geo/idx"] if use_fps else g["geo/idx_self"])
    fps = L(g["geo/fps"]) if use_fps else None
    out = m(f, idx, T(g["geo/xyz"]), FPS_idx=fps, xyz=is_xyz)
    assert np.abs(out.detach().numpy() - g[tag + "/out"]).max() < TOL
    out.backward(randn(out.shape, seed=4242))
    assert np.abs(f.grad.numpy() - g[tag + "/gf"]).max() < TOL
    for n, p in m.named_parameters():
        key = tag + "/g." + n
        if key in g:
            assert np.abs(p.grad.numpy() - g[key]).max() < TOL, n
        else:
            assert p.grad is None, n


@pytest.mark.parametrize("tag,cls", [("lm_cls", R.LocalMergeCls), ("lm_seg", R.LocalMergeSeg)])
def test_ref_cpu_local_merge(golden_blocks, tag, cls):
    g = golden_blocks
    xyz, fps = T(g["geo/xyz"]), L(g["geo/fps"])
    sub = R.index_points(xyz, fps)
    m0 = fill_state(cls(32, 64, 8, residual=True), seed=3).train()
    f0, _, i0, d0 = m0(xyz=xyz, base_xyz=xyz, normal=xyz)
    assert np.array_equal(i0.numpy(), g[tag + "/idx0"])
    assert np.abs(f0.detach().numpy() - g[tag + "/f0"]).max() < TOL
    m1 = fill_state(cls(64, 64, 8, residual=False), seed=4).train()
    feat = T(g[tag + "/f0"]).requires_grad_(True)
    f1, n1, i1, _ = m1(xyz=sub, base_xyz=xyz, normal=xyz, feature=feat, FPS_idx=fps)
    assert np.array_equal(i1.numpy(), g[tag + "/idx1"])
    assert np.abs(f1.detach().numpy() - g[tag + "/f1"]).max() < TOL
    assert bool(g[tag + "/normal1_is_indexed"]) == (n1.shape[1] == sub.shape[1])


def test_ref_cpu_upsample(golden_blocks):
    g = golden_blocks
    pts = T(g["up/pts"]).requires_grad_(True)
    up = R.upsample(pts, L(g["up/idx"]))
    assert np.abs(up.detach().numpy() - g["up/out"]).max() < 1e-6
    assert int(g["up/uncovered"]) > 0
    up.backward(randn(up.shape, seed=4242))
    assert np.abs(pts.grad.numpy() - g["up/gpts"]).max() < 1e-6
    up4 = R.upsample(T(g["up4/pts"]), L(g["up4/idx"]), scale_ratio=4)
    assert np.abs(up4.numpy() - g["up4/out"]).max() < 1e-6


def test_ref_cpu_feature_propagation(golden_blocks):
    g = golden_blocks
    xyz, fps = T(g["geo/xyz"]), L(g["geo/fps"])
    sub = R.index_points(xyz, fps)
    m = fill_state(R.PointNetFeaturePropagation(32, [48], act=True), seed=5).train()
    out = m(xyz, sub, None, T(g["fp/points2"]))
    assert np.abs(out.detach().numpy() - g["fp/out"]).max() < TOL


def test_ref_cpu_fuse(golden_fuse):
    g = golden_fuse
    x0 = T(g["x0"])
    fps = [L(g["fps%d" % l]) for l in range(4)]
    knn = [L(g["knn%d" % l]) for l in range(5)]
    feats = [T(g["f%d" % l]) for l in range(5)]
    xs = [x0]
    for p in fps:
        xs.append(R.index_points(xs[-1], p))
    m = fill_state(R.Fuse(64, 64, 64, 128, 256), seed=6).train()
    for lvl in range(5):
        out = m(xs[lvl].shape[1], f0=feats[0], f1=feats[1], f2=feats[2], f3=feats[3], f4=feats[4],
                FPS_0=fps[0], FPS_1=fps[1], FPS_2=fps[2], FPS_3=fps[3],
                knn_0=knn[0], knn_1=knn[1], knn_2=knn[2], knn_3=knn[3], knn_4=knn[4],
                xyz0=xs[0], xyz1=xs[1], xyz2=xs[2], xyz3=xs[3], xyz4=xs[4])
        assert np.abs(out[lvl].detach().numpy() - g["out%d" % lvl]).max() < TOL, lvl


def _check_model(g, model, run):
    model.eval()
    torch.manual_seed(2024)
    with torch.no_grad():
        assert np.abs(run(model).numpy() - g["out_eval"]).max() < TOL
    model.train()
    torch.manual_seed(2024)
    out = run(model)
    assert np.abs(out.detach().numpy() - g["out_train"]).max() < TOL
    (out * randn(out.shape, seed=31337)).sum().backward()
    names = list(g["grad_names"])
    for n, p in model.named_parameters():
        i = names.index(n)
        assert (p.grad is not None) == bool(g["has_grad"][i]), n
        if p.grad is not None:
            ref = float(g["grad_norms"][i])
            assert abs(float(p.grad.double().norm()) - ref) <= 1e-3 * max(ref, 1e-3), n
        if "grad." + n in g:
            assert np.abs(p.grad.numpy() - g["grad." + n]).max() < TOL * max(1.0, np.abs(g["grad." + n]).max()), n


def test_ref_cpu_cls_model(golden_cls):
    torch.set_num_threads(1)
    args = Namespace(num_point=1024, return_dist=True, cuda_ops=False, num_class=40)
    model = fill_state(R.ClsModel(args), seed=0)
    model.drop1.p = model.drop2.p = 0.0
    pts = T(golden_cls["points"])
    _check_model(golden_cls, model, lambda m: m(pts))


def test_ref_cpu_seg_model(golden_seg):
    torch.set_num_threads(1)
    model = fill_state(R.PartSegModel(50), seed=0)
    model.drop1.p = 0.0
    pts, lab = T(golden_seg["points"]), T(golden_seg["label"])
    _check_model(golden_seg, model, lambda m: m(pts, lab)[0])


def test_ref_cpu_umbrella_front_end():
    """the oracle's restatement of group_by_umbrella / cal_normal / cal_center / xyz2sphere /
    cal_const / check_nan_umb / UmbrellaSurfaceConstructor against vectors produced by the reference."""
    from conftest import load_golden
    g = load_golden("umbrella.npz")
    xyz = torch.from_numpy(g["xyz"])
    torch.set_num_threads(1)
    tri = R.group_by_umbrella(xyz, xyz, k=9)
    assert np.array_equal(tri.numpy(), g["triangles"])
    m = fill_state(R.UmbrellaSurfaceConstructor(9, 10, aggr_type="sum", return_dist=True, random_inv=False), seed=13)
    assert np.array_equal(m.features(xyz).numpy(), g["features"])
    for tag, rinv in (("det", False), ("rinv", True)):
        m = fill_state(R.UmbrellaSurfaceConstructor(9, 10, aggr_type="sum", return_dist=True, random_inv=rinv), seed=13).train()
        torch.manual_seed(77)
        out = m(xyz.transpose(1, 2).clone())
        assert np.abs(out.detach().numpy() - g[tag + "/out"]).max() < 1e-6 * np.abs(g[tag + "/out"]).max()
        (out * randn(out.shape, seed=5)).sum().backward()
        for n, p in m.named_parameters():
            ref = g[tag + "/grad." + n]
            assert np.abs(p.grad.numpy() - ref).max() < 1e-5 * max(1.0, np.abs(ref).max()), n
    m = fill_state(R.UmbrellaSurfaceConstructor(9, 10, aggr_type="sum", return_dist=True, random_inv=False), seed=13).eval()
    with torch.no_grad():
        assert np.abs(m(xyz.transpose(1, 2).clone()).numpy() - g["eval/out"]).max() < 1e-6 * np.abs(g["eval/out"]).max()


def test_ref_cpu_repsurf_2x_model():
    """The oracle's restatement of the RepSurf baseline classifier (models/repsurf/repsurf_ssg_umb_2x.py)
    against outputs of the reference itself."""
    from conftest import load_golden
    g = load_golden("repsurf2x_model.npz")
    args = Namespace(return_center=True, return_polar=True, num_point=1024, return_dist=True, group_size=8,
                     umb_pool="sum", cuda_ops=False, num_class=40)
    torch.set_num_threads(1)
    m = fill_state(R.RepSurf2xModel(args), seed=21)
    for mm in m.modules():
        if isinstance(mm, torch.nn.Dropout):
            mm.p = 0.0
    pts = torch.from_numpy(g["points"])
    m.eval()
    torch.manual_seed(5)
    with torch.no_grad():
        assert np.abs(m(pts.clone()).numpy() - g["out_eval"]).max() < 1e-5
    m.train()
    torch.manual_seed(5)
    assert np.abs(m(pts.clone()).detach().numpy() - g["out_train"]).max() < 1e-5


# ------------------------------------------------------------------ round 2 fixtures (tests/golden/round2.npz)
@pytest.mark.parametrize("tag", ["fpsc_2", "fpsc_6", "fpsc_10", "fpsc_64"])
def test_c_fps_any_channel_count(golden_round2, tag):
    """farthest_point_sample on rows of width C != 3 (reference modules/pointnet2_utils.py:84-109): the
    channel sum follows torch.sum's order (Appendix A2, incl. the C % 8 tail at C = 10)."""
    g = golden_round2
    S = g[tag + "/idx"].shape[1]
    idx = co.farthest_point_sample(g[tag + "/x"], S, g[tag + "/start"])
    assert np.array_equal(idx, g[tag + "/idx"].astype(np.int64))


def test_ref_cpu_losses(golden_round2):
    """SmoothClsLoss (util/utils.py:74-88) and get_loss (pointnet2_part_seg_msg.py:159-180)."""
    g = golden_round2
    pred = T(g["cls_loss/pred"]).requires_grad_(True)
    loss = R.smooth_cls_loss(pred, L(g["cls_loss/target"]))
    assert abs(float(loss) - float(g["cls_loss/loss"])) < 1e-6
    assert np.abs(torch.autograd.grad(loss, pred)[0].numpy() - g["cls_loss/gpred"]).max() < 1e-7
    pred = T(g["seg_loss/pred"]).requires_grad_(True)
    loss = R.partseg_loss(pred, L(g["seg_loss/target"]))
    assert abs(float(loss) - float(g["seg_loss/loss"])) < 1e-6
    assert np.abs(torch.autograd.grad(loss, pred)[0].numpy() - g["seg_loss/gpred"]).max() < 1e-8


@pytest.mark.parametrize("lvl", [3, 1])
def test_ref_cpu_fuse_backward(golden_fuse, golden_round2, lvl):
    g, g2 = golden_fuse, golden_round2
    x0 = T(g["x0"])
    fps = [L(g["fps%d" % l]) for l in range(4)]
    knn = [L(g["knn%d" % l]) for l in range(5)]
    feats = [T(g["f%d" % l]).requires_grad_(True) for l in range(5)]
    xs = [x0]
    for p in fps:
        xs.append(R.index_points(xs[-1], p))
    m = fill_state(R.Fuse(64, 64, 64, 128, 256), seed=6).train()
    out = m(xs[lvl].shape[1], f0=feats[0], f1=feats[1], f2=feats[2], f3=feats[3], f4=feats[4],
            FPS_0=fps[0], FPS_1=fps[1], FPS_2=fps[2], FPS_3=fps[3],
            knn_0=knn[0], knn_1=knn[1], knn_2=knn[2], knn_3=knn[3], knn_4=knn[4],
            xyz0=xs[0], xyz1=xs[1], xyz2=xs[2], xyz3=xs[3], xyz4=xs[4])[lvl]
    (out * randn(out.shape, seed=4242)).sum().backward()
    for l in range(5):
        ref = g2["fuse_bwd%d/gf%d" % (lvl, l)]
        assert np.abs(feats[l].grad.numpy() - ref).max() < TOL * max(1.0, np.abs(ref).max()), l
    for n, p_ in m.named_parameters():
        key = "fuse_bwd%d/g.%s" % (lvl, n)
        assert (p_.grad is not None) == (key in g2), n
        if p_.grad is not None:
            assert np.abs(p_.grad.numpy() - g2[key]).max() < TOL * max(1.0, np.abs(g2[key]).max()), n
