"""Generates the golden vectors under tests/golden/*.npz by IMPORTING THE REFERENCE.

Runs only in the build container (the reference tree is read-only at /root/reference and
never travels).  Every expected output below is produced by the reference's own functions
and nn.Modules (through the aliasing shim in _ref_shim.py); inputs come from seeded CPU
generators and are stored next to the outputs.  Weights are filled from name-hashed seeds
(param_fill.py), identically on the reference, oracle and HIP sides.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

torch.set_num_threads(1) is used so the files are reproducible.
"""
import os
import sys
from argparse import Namespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_shim import load_reference  # noqa: E402
from param_fill import fill_state, unit_cloud, randn  # noqa: E402

torch.set_num_threads(1)
p2, rs, cls_mod, seg_mod = load_reference()


def npy(t):
    return t.detach().cpu().numpy()


def save(name, d):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **d)
    print("%-18s %7.1f KiB  %d arrays" % (name, os.path.getsize(path) / 1024, len(d)))


# ------------------------------------------------------------------ index ops (bit-exact)
def gen_index_ops():
    d = {}
    # FPS single level + the cls 5-level chain + larger clouds
    for tag, (B, N, S) in {"fps_a": (2, 1024, 512), "fps_b": (1, 2048, 1024), "fps_c": (1, 4096, 2048),
                           "fps_d": (3, 100, 37)}.items():
        xyz = unit_cloud(B, N, seed=11 + N)
        torch.manual_seed(N + S)
        start = torch.randint(0, N, (B,), dtype=torch.long)
        torch.manual_seed(N + S)
        idx = p2.farthest_point_sample(xyz, S)
        assert (idx[:, 0] == start).all()
        d[tag + "/xyz"], d[tag + "/start"], d[tag + "/idx"] = npy(xyz), npy(start), npy(idx).astype(np.int16)
    xyz = unit_cloud(2, 1024, seed=77)
    d["fps_chain/xyz"] = npy(xyz)
    cur = xyz
    torch.manual_seed(99)
    for lvl, S in enumerate((512, 256, 128, 64, 32)):
        st = torch.get_rng_state()
        start = torch.randint(0, cur.shape[1], (2,), dtype=torch.long)
        torch.set_rng_state(st)
        idx = rs.farthest_point_sample(cur, S)
        d["fps_chain/start%d" % lvl], d["fps_chain/idx%d" % lvl] = npy(start), npy(idx).astype(np.int16)
        cur = rs.index_points(cur, idx)

    # kNN: real stage shapes (S, N, C); xyz-space uses a sub-sampled query set like the model
    for tag, (B, S, N, C) in {"knn_a": (2, 1024, 1024, 3), "knn_b": (2, 512, 1024, 3), "knn_c": (1, 512, 1024, 64),
                              "knn_d": (2, 64, 128, 128), "knn_e": (2, 32, 64, 256), "knn_f": (1, 2048, 2048, 3),
                              "knn_g": (2, 50, 70, 3), "knn_h": (1, 128, 512, 3)}.items():
        if C == 3:
            base = unit_cloud(B, N, seed=200 + S + N)
        else:
            base = randn((B, N, C), seed=300 + S + C, scale=0.7)
        g = torch.Generator().manual_seed(S * 7 + C)
        sel = torch.stack([torch.randperm(N, generator=g)[:S] for _ in range(B)])
        query = p2.index_points(base, sel) if S < N else base.clone()
        dist, idx = p2.knn_point(8, base, query)
        d[tag + "/base"], d[tag + "/query"] = npy(base), npy(query)
        d[tag + "/dist"], d[tag + "/idx"] = npy(dist), npy(idx).astype(np.int16)
    # square_distance itself (bitwise) on one small case
    a, b = unit_cloud(1, 96, seed=5), unit_cloud(1, 160, seed=6)
    d["sqd/src"], d["sqd/dst"], d["sqd/out"] = npy(a), npy(b), npy(p2.square_distance(a, b))

    # ball query: the RepSurf baseline's three stages
    for tag, (S, N, r) in {"ball_a": (512, 1024, 0.1), "ball_b": (128, 512, 0.2), "ball_c": (32, 128, 0.4)}.items():
        base = unit_cloud(2, N, seed=400 + N)
        query = base[:, :S].contiguous()
        idx = p2.query_ball_point(r, 24, base, query)
        d[tag + "/base"], d[tag + "/query"], d[tag + "/idx"] = npy(base), npy(query), npy(idx).astype(np.int16)
        d[tag + "/radius"] = np.float64(r)
    # 3-NN as PointNetFeaturePropagation computes it (pointnet2_utils.py:899-901)
    x1, x2 = unit_cloud(2, 512, seed=31), unit_cloud(2, 128, seed=32)
    dd, ii = p2.square_distance(x1, x2).sort(dim=-1)
    d["nn3/xyz1"], d["nn3/xyz2"] = npy(x1), npy(x2)
    d["nn3/dist"], d["nn3/idx"] = npy(dd[:, :, :3]), npy(ii[:, :, :3]).astype(np.int16)
    save("index_ops.npz", d)


# ------------------------------------------------------------------ blocks (1e-4 features)
def grads_of(out, wrt):
    g = torch.autograd.grad(out, wrt, grad_outputs=randn(out.shape, seed=4242), allow_unused=True)
    return [None if t is None else npy(t) for t in g]


def gen_blocks():
    d = {}
    B, N, S, K = 2, 256, 128, 8
    xyz = unit_cloud(B, N, seed=501)
    torch.manual_seed(3)
    fps = p2.farthest_point_sample(xyz, S)
    sub = p2.index_points(xyz, fps)
    _, idx = p2.knn_point(K, xyz, sub)
    _, idx_self = p2.knn_point(K, xyz, xyz)
    d["geo/xyz"], d["geo/fps"], d["geo/idx"], d["geo/idx_self"] = npy(xyz), npy(fps), npy(idx), npy(idx_self)

    # Linear: train (batch statistics) and eval (running statistics), act on/off
    for tag, (ci, co, act) in {"lin_a": (64, 128, True), "lin_b": (3, 64, True), "lin_c": (128, 64, False)}.items():
        m = fill_state(p2.Linear(ci, co, bn=False, act=act), seed=1)
        x = randn((B, S, ci), seed=600 + ci).requires_grad_(True)
        m.train()
        y = m(x)
        gx, gw, gb, gg, gbeta = grads_of(y, [x, m.linear.weight, m.linear.bias, m.norm2.weight, m.norm2.bias])
        d[tag + "/x"], d[tag + "/y_train"] = npy(x), npy(y)
        d[tag + "/gx"], d[tag + "/gw"], d[tag + "/gb"], d[tag + "/ggamma"], d[tag + "/gbeta"] = gx, gw, gb, gg, gbeta
        d[tag + "/running_mean"], d[tag + "/running_var"] = npy(m.norm2.running_mean), npy(m.norm2.running_var)
        m = fill_state(p2.Linear(ci, co, bn=False, act=act), seed=1).eval()
        d[tag + "/y_eval"] = npy(m(x))

    # LocalTrans: xyz branch (self and FPS-subsampled) and feature branch, residual on/off
    for tag, (ci, co, residual, use_fps, is_xyz) in {
        "lt_xyz_self": (3, 64, True, False, True), "lt_xyz_fps": (3, 64, True, True, True),
        "lt_feat_id": (64, 64, False, True, False), "lt_feat_res": (64, 128, True, True, False),
        "lt_feat_self": (32, 32, False, False, False),
    }.items():
        m = fill_state(p2.LocalTrans(ci, co, K, usetanh=False, residual=residual), seed=2).train()
        f = (xyz.clone() if is_xyz else randn((B, N, ci), seed=700 + ci)).requires_grad_(True)
        out = m(features=f, idx=idx if use_fps else idx_self, pos=xyz, FPS_idx=fps if use_fps else None, xyz=is_xyz)
        names = [n for n, p in m.named_parameters()]
        gs = grads_of(out, [f] + [p for _, p in m.named_parameters()])
        d[tag + "/f"], d[tag + "/out"], d[tag + "/gf"] = npy(f), npy(out), gs[0]
        for n, g in zip(names, gs[1:]):
            if g is not None:
                d[tag + "/g." + n] = g
        m.eval()
        d[tag + "/out_eval"] = npy(m(features=f, idx=idx if use_fps else idx_self, pos=xyz,
                                     FPS_idx=fps if use_fps else None, xyz=is_xyz))

    # LocalMerge, cls and seg variants, first level (feature=None) and a down-sampling level
    for tag, cls in {"lm_cls": rs.LocalMerge, "lm_seg": p2.LocalMerge}.items():
        m0 = fill_state(cls(32, 64, K, usetanh=False, residual=True), seed=3).train()
        f0, n0, i0, dist0 = m0(xyz=xyz, base_xyz=xyz, normal=xyz)
        d[tag + "/f0"], d[tag + "/idx0"], d[tag + "/dist0"] = npy(f0), npy(i0), npy(dist0)
        m1 = fill_state(cls(64, 64, K, usetanh=False, residual=False), seed=4).train()
        feat = f0.detach().clone().requires_grad_(True)
        f1, n1, i1, _ = m1(xyz=sub, base_xyz=xyz, normal=xyz, feature=feat, FPS_idx=fps)
        d[tag + "/f1"], d[tag + "/idx1"] = npy(f1), npy(i1)
        # feature-space kNN indices the block chose (for teacher forcing on the GPU side)
        d[tag + "/idx1_feat"] = npy(p2.knn_point(K, feat, p2.index_points(feat, fps))[1])
        d[tag + "/gfeat"] = grads_of(f1, [feat])[0]
        d[tag + "/normal1_is_indexed"] = np.bool_(n1.shape[1] == S)

    # upsample: plain, ratio 4, and the two quirks (a coarse row whose channel 0 is exactly
    # zero is not counted; an uncovered fine point stays zero)
    pts = randn((B, S, 16), seed=800).clone()
    pts[0, 5, 0] = 0.0
    pts[1, 9, :] = 0.0
    kidx = idx.clone()                      # values < N = 2*S
    kidx[0][kidx[0] == 7] = 8                # fine point 7 of cloud 0 is covered by nobody
    pts.requires_grad_(True)
    up = p2.upsample(pts, kidx)
    d["up/pts"], d["up/idx"], d["up/out"], d["up/gpts"] = npy(pts), npy(kidx), npy(up), grads_of(up, [pts])[0]
    d["up/uncovered"] = np.int64((npy(up).reshape(B, N, -1) == 0).all(-1).sum())
    sub4 = p2.index_points(sub, p2.farthest_point_sample(sub, S // 2))
    _, kidx4 = p2.knn_point(K, xyz, sub4)
    pts4 = randn((B, S // 2, 8), seed=801)
    d["up4/pts"], d["up4/idx"], d["up4/out"] = npy(pts4), npy(kidx4), npy(p2.upsample(pts4, kidx4, scale_ratio=4))

    # PointNetFeaturePropagation (3-NN inverse-distance interpolation + Linear)
    m = fill_state(p2.PointNetFeaturePropagation(32, [48], act=True), seed=5).train()
    p2f = randn((B, S, 32), seed=810).requires_grad_(True)
    out = m(xyz, sub, None, p2f)
    d["fp/points2"], d["fp/out"], d["fp/gpoints2"] = npy(p2f), npy(out), grads_of(out, [p2f])[0]
    save("blocks.npz", d)


# ------------------------------------------------------------------ Fuse (seg cross-state)
def gen_fuse():
    d = {}
    B, N = 1, 2048
    x0 = unit_cloud(B, N, seed=900)
    torch.manual_seed(17)
    xs, fps, knn = [x0], [], []
    for S in (1024, 512, 256, 128):
        p = p2.farthest_point_sample(xs[-1], S)
        fps.append(p)
        xs.append(p2.index_points(xs[-1], p))
    knn.append(p2.knn_point(8, x0, x0)[1])
    for lvl in range(1, 5):
        knn.append(p2.knn_point(8, xs[lvl - 1], xs[lvl])[1])
    chans = (64, 64, 64, 128, 256)
    feats = [randn((B, xs[l].shape[1], chans[l]), seed=910 + l) for l in range(5)]
    d["x0"] = npy(x0)
    for l in range(4):
        d["fps%d" % l] = npy(fps[l]).astype(np.int16)
    for l in range(5):
        d["knn%d" % l] = npy(knn[l]).astype(np.int16)
        d["f%d" % l] = npy(feats[l])
    m = fill_state(p2.Fuse(*chans), seed=6).train()
    for lvl, npnt in zip((4, 3, 2, 1, 0), (128, 256, 512, 1024, 2048)):
        out = m(npnt, f0=feats[0], f1=feats[1], f2=feats[2], f3=feats[3], f4=feats[4],
                FPS_0=fps[0], FPS_1=fps[1], FPS_2=fps[2], FPS_3=fps[3],
                knn_0=knn[0], knn_1=knn[1], knn_2=knn[2], knn_3=knn[3], knn_4=knn[4],
                xyz0=xs[0], xyz1=xs[1], xyz2=xs[2], xyz3=xs[3], xyz4=xs[4])
        d["out%d" % lvl] = npy(out[lvl])
    save("fuse.npz", d)


# ------------------------------------------------------------------ RepSurf-style set abstraction (a14)
def gen_sa():
    d = {}
    B, N = 2, 512
    xyz = unit_cloud(B, N, seed=950)
    nrm = randn((B, N, 3), seed=951)
    feat = randn((B, N, 16), seed=952)
    torch.manual_seed(21)
    c, n, f = rs.sample_and_group(128, 0.2, 24, xyz, nrm, feat, return_normal=True, return_polar=False, cuda=False)
    d["xyz"], d["normal"], d["feature"] = npy(xyz), npy(nrm), npy(feat)
    d["sg/center"], d["sg/normal"], d["sg/feature"] = npy(c), npy(n), npy(f)
    m = fill_state(rs.SurfaceAbstractionCD(npoint=128, radius=0.2, nsample=24, feat_channel=16 + 3, pos_channel=3,
                                           mlp=[32, 64], group_all=False, return_polar=False, cuda=False), seed=7).train()
    torch.manual_seed(21)
    oc, on, of = m(xyz.transpose(1, 2), nrm.transpose(1, 2), feat.transpose(1, 2))
    d["sa/center"], d["sa/normal"], d["sa/feature"] = npy(oc), npy(on), npy(of)
    # the same with the spherical coordinates of the grouped offsets (return_polar)
    torch.manual_seed(21)
    c, n, f = rs.sample_and_group(128, 0.2, 24, xyz, nrm, feat, return_normal=True, return_polar=True, cuda=False)
    d["sgp/feature"] = npy(f)
    m = fill_state(rs.SurfaceAbstractionCD(npoint=128, radius=0.2, nsample=24, feat_channel=16 + 3, pos_channel=6,
                                           mlp=[32, 64], group_all=False, return_polar=True, cuda=False), seed=8).train()
    torch.manual_seed(21)
    oc, on, of = m(xyz.transpose(1, 2), nrm.transpose(1, 2), feat.transpose(1, 2))
    d["sap/feature"] = npy(of)
    save("sa.npz", d)


# ------------------------------------------------------------------ umbrella surface front-end
def gen_umbrella():
    import importlib
    recons = importlib.import_module("modules.recons_utils")
    polar = importlib.import_module("modules.polar_utils")
    d = {}
    B, N = 2, 512
    xyz = unit_cloud(B, N, seed=1234)
    xyz[0, 7] = xyz[0, 3]                      # a duplicated point: degenerate triangles -> the NaN replacement path
    d["xyz"] = npy(xyz)
    tri = rs.group_by_umbrella(xyz, xyz, k=9)
    d["triangles"] = npy(tri)
    normal = recons.cal_normal(tri, random_inv=False, is_group=True)
    center = recons.cal_center(tri)
    pol = polar.xyz2sphere(center)
    pos = recons.cal_const(normal, center)
    normal, center, pos = recons.check_nan_umb(normal, center, pos)
    d["features"] = npy(torch.cat([center, pol, normal, pos], dim=-1))          # [B,N,8,10]
    for tag, rinv in (("det", False), ("rinv", True)):
        m = fill_state(rs.UmbrellaSurfaceConstructor(9, 10, aggr_type='sum', return_dist=True, random_inv=rinv,
                                                     cuda=False), seed=13).train()
        torch.manual_seed(77)
        out = m(xyz.transpose(1, 2).clone())
        d[tag + "/out"] = npy(out)
        (out * randn(out.shape, seed=5)).sum().backward()
        for n, p_ in m.named_parameters():
            d[tag + "/grad." + n] = npy(p_.grad)
        for n, b_ in m.named_buffers():
            if "running" in n:
                d[tag + "/buf." + n] = npy(b_)
    m = fill_state(rs.UmbrellaSurfaceConstructor(9, 10, aggr_type='sum', return_dist=True, random_inv=False,
                                                 cuda=False), seed=13).eval()
    with torch.no_grad():
        d["eval/out"] = npy(m(xyz.transpose(1, 2).clone()))
    save("umbrella.npz", d)


# ------------------------------------------------------------------ RepSurf baseline classifier (repsurf_ssg_umb_2x)
def gen_repsurf2x():
    import importlib
    mod = importlib.import_module("models.repsurf.repsurf_ssg_umb_2x")
    args = Namespace(return_center=True, return_polar=True, num_point=1024, return_dist=True, group_size=8,
                     umb_pool="sum", cuda_ops=False, num_class=40)
    d = {}
    # (seed chosen so that no point has an exact tie at its 9th-nearest distance: which of two equidistant
    #  neighbours torch.topk keeps is implementation-defined; seed 4242 has two such points)
    pts = unit_cloud(2, 1024, seed=4243).transpose(1, 2).contiguous()
    d["points"] = npy(pts)
    model = fill_state(mod.Model(args), seed=21)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.eval()
    torch.manual_seed(5)
    with torch.no_grad():
        d["out_eval"] = npy(model(pts.clone()))
    model.train()
    torch.manual_seed(5)
    out = model(pts.clone())
    d["out_train"] = npy(out)
    (out * randn(out.shape, seed=99)).sum().backward()
    names = [n for n, p_ in model.named_parameters()]
    d["grad_names"] = np.array(names)
    d["grad_norms"] = np.array([float(p_.grad.double().norm()) if p_.grad is not None else 0.0
                                for _, p_ in model.named_parameters()])
    for n, p_ in model.named_parameters():
        if n.startswith("surface_constructor") or n.startswith("classfier.8"):
            d["grad." + n] = npy(p_.grad)
    save("repsurf2x_model.npz", d)


# ------------------------------------------------------------------ whole models
def model_golden(model, run, params_full):
    """run(model) -> output tensor.  Returns eval output, train output, loss grads."""
    d = {}
    model.eval()
    torch.manual_seed(2024)
    d["out_eval"] = npy(run(model))
    model.train()
    torch.manual_seed(2024)
    out = run(model)
    d["out_train"] = npy(out)
    loss = (out * randn(out.shape, seed=31337)).sum()
    loss.backward()
    names, norms = [], []
    for n, p in model.named_parameters():
        names.append(n)
        norms.append(0.0 if p.grad is None else float(p.grad.double().norm()))
        if n in params_full and p.grad is not None:
            d["grad." + n] = npy(p.grad)
    d["grad_names"] = np.array(names)
    d["grad_norms"] = np.array(norms)
    d["has_grad"] = np.array([p.grad is not None for _, p in model.named_parameters()])
    return d


def gen_cls():
    args = Namespace(num_point=1024, return_dist=True, cuda_ops=False, num_class=40)
    model = fill_state(cls_mod.Model(args), seed=0)
    model.drop1.p = 0.0
    model.drop2.p = 0.0
    B = 4
    pts = unit_cloud(B, 1024, seed=1234).transpose(1, 2).contiguous()
    # trace the indices the reference chose (teacher forcing of the feature-space kNN)
    trace = {"fps": [], "knn": []}
    orig_knn, orig_fps = rs.knn_point, rs.farthest_point_sample

    def knn_t(k, a, b):
        r = orig_knn(k, a, b)
        trace["knn"].append(r[1])
        return r

    def fps_t(x, n):
        r = orig_fps(x, n)
        trace["fps"].append(r)
        return r

    rs.knn_point, rs.farthest_point_sample = knn_t, fps_t
    try:
        model.eval()
        torch.manual_seed(2024)
        model(pts)
        d_idx = {"fps%d" % i: npy(t).astype(np.int16) for i, t in enumerate(trace["fps"])}
        d_idx.update({"knn%d" % i: npy(t).astype(np.int16) for i, t in enumerate(trace["knn"])})
        trace["fps"].clear(); trace["knn"].clear()
        model.train()
        torch.manual_seed(2024)
        model(pts)
        d_idx.update({"train_knn%d" % i: npy(t).astype(np.int16) for i, t in enumerate(trace["knn"])})
    finally:
        rs.knn_point, rs.farthest_point_sample = orig_knn, orig_fps
    model = fill_state(cls_mod.Model(args), seed=0)
    model.drop1.p = 0.0
    model.drop2.p = 0.0
    full = {"fc3.weight", "keepHigh.la0.xyz_Trans.q.weight", "keepHigh.la1.feature_Trans.k.weight",
            "keepHigh.la3.fc2.linear.weight", "keepHigh.la5.feature_Trans2.ffn.norm2.weight",
            "keepHigh.final_class.bias", "keepHigh.la2.feature_Trans.v.bias"}
    d = model_golden(model, lambda m: m(pts), full)
    d.update(d_idx)
    d["points"] = npy(pts)
    save("cls_model.npz", d)


def gen_seg():
    model = fill_state(seg_mod.get_model(50), seed=0)
    model.drop1.p = 0.0
    B = 2
    pts = unit_cloud(B, 2048, seed=4321).transpose(1, 2).contiguous()
    lab = torch.eye(16)[[3, 11]].view(B, 1, 16)
    trace = {"knn": []}
    orig_knn = p2.knn_point

    def knn_t(k, a, b):
        r = orig_knn(k, a, b)
        trace["knn"].append(r[1])
        return r

    p2.knn_point = knn_t
    try:
        model.eval()
        torch.manual_seed(2024)
        model(pts, lab)
        d_idx = {"knn%d" % i: npy(t).astype(np.int16) for i, t in enumerate(trace["knn"])}
        trace["knn"].clear()
        model.train()
        torch.manual_seed(2024)
        model(pts, lab)
        d_idx.update({"train_knn%d" % i: npy(t).astype(np.int16) for i, t in enumerate(trace["knn"])})
    finally:
        p2.knn_point = orig_knn
    model = fill_state(seg_mod.get_model(50), seed=0)
    model.drop1.p = 0.0
    full = {"conv11.weight", "keepHigh.la0.xyz_Trans.k.weight", "keepHigh.fuse3.conv42.linear.weight",
            "keepHigh.la2_up.feature_Trans1.q.weight", "keepHigh.up_conv3.linear.weight"}
    d = model_golden(model, lambda m: m(pts, lab)[0], full)
    d.update(d_idx)
    d["points"], d["label"] = npy(pts), npy(lab)
    save("seg_model.npz", d)


# ------------------------------------------------------------------ round 2: generic-C FPS, losses, augmentation, Fuse gradients
def gen_round2():
    import importlib
    d = {}
    # farthest_point_sample on C != 3 (modules/pointnet2_utils.py:84-109 takes any C: the dataset reader
    # samples on [1,N,6] xyz|normal rows, the stale part-seg variant in feature space)
    for tag, (B, N, C, S) in {"fpsc_6": (2, 300, 6, 120), "fpsc_10": (2, 200, 10, 64), "fpsc_64": (2, 256, 64, 64),
                              "fpsc_2": (1, 100, 2, 30)}.items():
        x = randn((B, N, C), seed=60 + C, scale=0.5)
        torch.manual_seed(N + C)
        start = torch.randint(0, N, (B,), dtype=torch.long)
        torch.manual_seed(N + C)
        idx = p2.farthest_point_sample(x, S)
        assert (idx[:, 0] == start).all()
        d[tag + "/x"], d[tag + "/start"], d[tag + "/idx"] = npy(x), npy(start), npy(idx).astype(np.int16)

    # label-smoothed losses (util/utils.py:74-88 on log-probabilities; pointnet2_part_seg_msg.py:159-180 on logits)
    utils = importlib.import_module("util.utils")
    pred = torch.log_softmax(randn((16, 40), seed=71), -1).requires_grad_(True)
    tgt = torch.randint(0, 40, (16,), generator=torch.Generator().manual_seed(72))
    loss = utils.SmoothClsLoss()(pred, tgt)
    d["cls_loss/pred"], d["cls_loss/target"], d["cls_loss/loss"] = npy(pred), npy(tgt), npy(loss)
    d["cls_loss/gpred"] = npy(torch.autograd.grad(loss, pred)[0])
    logits = randn((2 * 2048, 50), seed=73, scale=2.0).requires_grad_(True)
    tgt = torch.randint(0, 50, (2, 2048), generator=torch.Generator().manual_seed(74))
    loss = seg_mod.get_loss()(logits, tgt, None)
    d["seg_loss/pred"], d["seg_loss/target"], d["seg_loss/loss"] = npy(logits), npy(tgt).astype(np.int16), npy(loss)
    d["seg_loss/gpred"] = npy(torch.autograd.grad(loss, logits)[0])

    # train-time augmentation (modules/ptaug_utils.py:22-62): scale then shift, per cloud, drawn with torch.rand on the
    # batch's device -- here the CPU generator; the draws are stored so the device side can be fed the same numbers
    ptaug = importlib.import_module("modules.ptaug_utils")
    batch = randn((4, 6, 128), seed=75)
    a = Namespace(aug_scale=True, aug_shift=True, dataset="ScanObjectNN")
    aug_args = ptaug.get_aug_args(a)
    torch.manual_seed(76)
    draws = [torch.rand(4, 3, 1), torch.rand(4, 3, 1)]          # what the two torch.rand calls will return
    torch.manual_seed(76)
    out = ptaug.transform_point_cloud(batch.clone(), a, aug_args)
    d["aug/batch"], d["aug/out"] = npy(batch), npy(out)
    d["aug/draw_scale"], d["aug/draw_shift"] = npy(draws[0]), npy(draws[1])
    d["aug/scale_factor"], d["aug/shift_factor"] = np.float64(aug_args["scale_factor"]), np.float64(aug_args["shift_factor"])
    torch.manual_seed(76)
    a2 = Namespace(aug_scale=False, aug_shift=True, dataset="ScanObjectNN")
    d["aug/out_shift_only"] = npy(ptaug.transform_point_cloud(batch.clone(), a2, aug_args))

    # Fuse backward (modules/pointnet2_utils.py:576-709) at two target levels: the same inputs as fuse.npz
    B, N = 1, 2048
    x0 = unit_cloud(B, N, seed=900)
    torch.manual_seed(17)
    xs, fps, knn = [x0], [], []
    for S in (1024, 512, 256, 128):
        p = p2.farthest_point_sample(xs[-1], S)
        fps.append(p)
        xs.append(p2.index_points(xs[-1], p))
    knn.append(p2.knn_point(8, x0, x0)[1])
    for lvl in range(1, 5):
        knn.append(p2.knn_point(8, xs[lvl - 1], xs[lvl])[1])
    chans = (64, 64, 64, 128, 256)
    for lvl, npnt in ((3, 256), (1, 1024)):
        feats = [randn((B, xs[l].shape[1], chans[l]), seed=910 + l).requires_grad_(True) for l in range(5)]
        m = fill_state(p2.Fuse(*chans), seed=6).train()
        out = m(npnt, f0=feats[0], f1=feats[1], f2=feats[2], f3=feats[3], f4=feats[4],
                FPS_0=fps[0], FPS_1=fps[1], FPS_2=fps[2], FPS_3=fps[3],
                knn_0=knn[0], knn_1=knn[1], knn_2=knn[2], knn_3=knn[3], knn_4=knn[4],
                xyz0=xs[0], xyz1=xs[1], xyz2=xs[2], xyz3=xs[3], xyz4=xs[4])[lvl]
        (out * randn(out.shape, seed=4242)).sum().backward()
        for l in range(5):
            d["fuse_bwd%d/gf%d" % (lvl, l)] = npy(feats[l].grad)
        for n, p_ in m.named_parameters():
            if p_.grad is not None:
                d["fuse_bwd%d/g.%s" % (lvl, n)] = npy(p_.grad)
    save("round2.npz", d)


if __name__ == "__main__":
    which = sys.argv[1:] or ["index", "blocks", "fuse", "sa", "umbrella", "repsurf2x", "cls", "seg", "round2"]
    if "round2" in which:
        gen_round2()
    if "umbrella" in which:
        gen_umbrella()
    if "repsurf2x" in which:
        gen_repsurf2x()
    if "sa" in which:
        gen_sa()
    if "index" in which:
        gen_index_ops()
    if "blocks" in which:
        gen_blocks()
    if "fuse" in which:
        gen_fuse()
    if "cls" in which:
        gen_cls()
    if "seg" in which:
        gen_seg()
