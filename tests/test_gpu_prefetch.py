"""Cross-step geometry (ops.GeometryPrefetch, csrc/geo_rider.h): the NEXT batch's sampling chain and first two coordinate
searches computed by rider workgroups inside the grouped weight-gradient launches of the current step.  Everything the
riders produce must equal the stand-alone entry points bit for bit (same device bodies), and a training step that reads
prefetched geometry must equal the step that runs the chain inside its own forward pass."""
from argparse import Namespace

import numpy as np
import pytest
import torch

from param_fill import fill_state, randn, unit_cloud

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import mpa_amd
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return mpa_amd.ops


def _starts(B, sizes):
    g = torch.Generator().manual_seed(99)
    return [torch.randint(0, n, (B,), generator=g).cuda() for n in sizes]


def _reference(ops, xyz, npoints, k, starts):
    idxs, xyzs, knns = [], [], []
    cur = xyz
    for s, st in zip(npoints, starts):
        i, x = ops.farthest_point_sample(cur, s, start_idx=st, return_xyz=True)
        idxs.append(i); xyzs.append(x)
        cur = x
    knns.append(ops.knn_point(k, xyz, xyz))
    knns.append(ops.knn_point(k, xyz, xyzs[0]))
    return idxs, xyzs, knns


@pytest.mark.parametrize("B,N,npoints", [(6, 1024, (512, 256, 128, 64, 32)), (3, 2048, (1024, 512, 256, 128)),
                                         (2, 4096, (2048, 1024, 512, 256)), (4, 300, (150, 70)), (5, 700, (300,))])
def test_riders_equal_separate_entry_points(ops, monkeypatch, B, N, npoints):
    """The two riders as launches of their own (mpa_geo_rider_f32) and inside a grouped weight-gradient call of 2
    launches (mpa_gemm_tn_grouped_rider_f32): FPS indices / coordinates of every level and both searches equal
    farthest_point_sample / knn_point; the weight gradients that carried them equal torch's."""
    k = 8
    xyz = unit_cloud(B, N, seed=N + B).cuda()
    starts = _starts(B, (N,) + tuple(npoints[:-1]))
    want = _reference(ops, xyz, npoints, k, starts)
    for carried in (False, True):
        pf = ops.GeometryPrefetch()
        pf.spec = ((B, N, 3), tuple(npoints), k)
        assert pf.supported()
        pf.allocate(xyz.device)
        monkeypatch.setattr(pf, "_draw_starts", lambda device: starts)
        if not carried:
            pf.compute_now(xyz)
        else:
            # 45 small weight-gradient products -> two launches (40 problems each at most): one rider per launch
            g = torch.Generator().manual_seed(1)
            gy = [torch.randn(500 + 8 * i, 64, generator=g).cuda() for i in range(45)]
            xs = [torch.randn(500 + 8 * i, 32, generator=g).cuda() for i in range(45)]
            outs = [torch.empty(64, 32, device="cuda") for _ in range(45)]
            ops.defer_weight_grads(True)
            try:
                for a, b_, o in zip(gy, xs, outs):
                    ops._weight_grad(a, 64, b_, 32, o, 64, 32, a.shape[0], direct=True)
                pf.next_xyz.copy_(xyz)
                pf.starts = starts
                ops.flush_weight_grads(riders=pf.riders())
            finally:
                ops.defer_weight_grads(False)
            for a, b_, o in zip(gy, xs, outs):
                assert torch.allclose(o, a.t() @ b_, rtol=1e-4, atol=1e-3)
        torch.cuda.synchronize()
        for lvl in range(len(npoints)):
            assert torch.equal(pf.fps_idx[lvl], want[0][lvl]), (carried, lvl)
            assert torch.equal(pf.fps_xyz[lvl], want[1][lvl]), (carried, lvl)
        for i in range(2):
            assert torch.equal(pf.knn[i][1], want[2][i][1]) and torch.equal(pf.knn[i][0], want[2][i][0]), (carried, i)


def test_fused_searches_without_sampling(ops):
    """ops.knn_xyz_and_feature (the fused search launch with no sampling workgroups) == two knn_point calls."""
    for B, N, S, C in [(40, 1024, 512, 64), (3, 256, 128, 128), (2, 300, 77, 64), (2, 100, 50, 32)]:
        xb = unit_cloud(B, N, seed=N).cuda()
        xq = xb[:, :S].contiguous()
        fb, fq = randn((B, N, C), seed=1).cuda(), randn((B, S, C), seed=2).cuda()
        (dx, ix), (df, jf) = ops.knn_xyz_and_feature(8, xb, xq, 8, fb, fq)
        ops.clear_knn_memo()
        d0, i0 = ops.knn_point(8, xb, xq)
        d1, i1 = ops.knn_point(8, fb, fq)
        assert torch.equal(ix, i0) and torch.equal(dx, d0) and torch.equal(jf, i1) and torch.equal(df, d1)


@pytest.mark.parametrize("form", [True, "riders"])
def test_prefetched_step_equals_in_pass_chain(ops, monkeypatch, form):
    """GraphedTrainStep(prefetch_geometry=True: the next batch's chain carried by this batch's search launches,
    ops.GeometryPipeline; "riders": by the closing weight-gradient launches, ops.GeometryPrefetch) alternating between
    distinct batches, each announced a step ahead, against the plain step (chain inside the forward pass) on the same
    batches: same losses and same gradients.  (Fixed
    parameters -- lr = 0 -- bit-reproducible BatchNorm statistics and sampling starts that depend on the level only, so
    both runs see identical inputs.)  An unannounced batch still gets the right geometry (computed on the spot)."""
    from mpa_amd.models.repsurf.repsurf_ssg_umb import Model, SmoothClsLoss
    from mpa_amd.runtime import GraphedTrainStep
    B, N = 6, 1024
    monkeypatch.setattr(ops, "_fps_start", lambda B_, N_, device, start_idx=None: (
        (torch.arange(B_, device=device) * 7 + 3) % N_ if start_idx is None else start_idx.to(device)))
    old = ops.set_deterministic(True)
    try:
        batches = [(unit_cloud(B, N, seed=s).transpose(1, 2).contiguous().cuda(), ((torch.arange(B) + s) % 40).cuda())
                   for s in (11, 12, 13)]

        def run(prefetch):
            torch.manual_seed(0)
            model = fill_state(Model(Namespace(num_point=N, return_dist=True, cuda_ops=True, num_class=40)), seed=1).cuda().train()
            model.drop1.p = model.drop2.p = 0.0
            step = GraphedTrainStep(model, SmoothClsLoss(), batches[0], lr=0.0, prefetch_geometry=prefetch)
            assert (step.prefetch is not None) == bool(prefetch)
            if prefetch:
                assert isinstance(step.prefetch, ops.GeometryPrefetch if prefetch == "riders" else ops.GeometryPipeline)
            out = []
            try:
                order = [0, 1, 0, 2, 2, 1]
                for t, bi in enumerate(order):
                    nxt = batches[order[t + 1]] if t + 1 < len(order) and t != 3 else None     # step 3 announces nothing:
                    loss = step(*batches[bi], next_batch=nxt)                                  # -> step 4's batch is computed on the spot
                    torch.cuda.synchronize()
                    assert ops._PREFETCH is None        # installed only inside the step's own passes: any other forward
                                                        # of the model (an evaluation between steps) runs its own chain
                    out.append((float(loss), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
            finally:
                step.close()
            return out

        ref, got = run(False), run(form)
        for t, ((l0, g0), (l1, g1)) in enumerate(zip(ref, got)):
            assert abs(l0 - l1) < 1e-6, (t, l0, l1)
            assert g0.keys() == g1.keys()
            for n in g0:
                err = float((g0[n] - g1[n]).norm() / g0[n].norm().clamp_min(1e-12))
                assert err < 1e-3 or float((g0[n] - g1[n]).abs().max()) < 1e-6, (t, n, err)
        assert abs(ref[0][0] - ref[1][0]) > 1e-4           # (the batches are distinct: a stale geometry would show)
    finally:
        ops.set_deterministic(old)


# ------------------------------------------------------------------------------- coarse states: one launch per state
@pytest.mark.parametrize("B,fN,fS,xN,xS,N,S,C,K", [
    (64, 128, 64, 256, 128, 256, 128, 64, 8),        # cls la3: 128 queries in 256 rows of 64 channels + FPS 128 -> 64
    (64, 64, 32, 128, 64, 128, 64, 128, 8),          # cls la4
    (64, None, None, 64, 32, 64, 32, 256, 8),        # cls la5: no next state
    (3, 100, 33, 200, 77, 200, 77, 32, 5),           # ragged sizes, K = 5
    (2, 50, 50, None, None, 40, 40, 128, 8),         # no coordinate search; K = 8 of 40
    (2, None, None, 9, 9, 9, 9, 64, 8),              # fewer base rows than a tile, K = 8 of 9
    (5, 128, 100, 256, 256, 96, 200, 256, 3),        # the widest rows: 96 of them fit the LDS (128 do not)
])
def test_coarse_level_equals_separate_entry_points(ops, B, fN, fS, xN, xS, N, S, C, K):
    """mpa_coarse_level_f32 (a coarse state's sampling + coordinate search + feature search as one launch of small
    workgroups: whole base staged at once, one pass of MFMA distances, selection by 64-bit (distance, index) minima)
    against farthest_point_sample / knn_point: indices and distance bits, including duplicated rows (exact ties ->
    lowest index first)."""
    g = torch.Generator().manual_seed(N * 7 + C)
    fin = unit_cloud(B, fN, seed=fN).cuda() if fN else None
    xb = unit_cloud(B, xN, seed=xN + 1).cuda() if xN else None
    xq = xb[:, :xS].contiguous() if xN else None
    fb = torch.randn(B, N, C, generator=g)
    if N > 8:
        fb[:, 5] = fb[:, 2]                      # a duplicated base row: an exact tie in every query's list
    fb = fb.cuda()
    fq = torch.cat((fb[:, :S // 2], torch.randn(B, S - S // 2, C, generator=g).cuda() * 0.7), 1).contiguous()
    start = (torch.arange(B) * 3) % fN if fN else None
    assert ops._coarse_ok(fN, xN, N, C, K if xN else None, K, fb, fq)
    if fN:
        fidx, fxyz, rx, (df, jf) = ops.fps_knn_fused(fin, fS, K, xb, xq, K, fb, fq, start_idx=start)
        fidx0, fxyz0 = ops.farthest_point_sample(fin, fS, start_idx=start, return_xyz=True)
        assert torch.equal(fidx, fidx0) and torch.equal(fxyz, fxyz0)
    else:
        rx, (df, jf) = ops.knn_xyz_and_feature(K, xb, xq, K, fb, fq)
    ops.clear_knn_memo()
    old = ops.set_fps_feature_fusion(False)
    try:
        df0, jf0 = ops.knn_point(K, fb, fq)
        assert torch.equal(jf, jf0) and torch.equal(df, df0)
        if xN:
            dx0, ix0 = ops.knn_point(K, xb.clone(), xq)
            assert torch.equal(rx[1], ix0) and torch.equal(rx[0], dx0)
        else:
            assert rx is None
    finally:
        ops.set_fps_feature_fusion(old)


def test_pipelined_partseg_step_at_4096_points(ops, monkeypatch):
    """The cross-step pipeline on the part-seg wiring with 4096-point blocks (BASELINE configs[3]'s shape: the level-1
    sampling of the next batch -- 4096 -> 2048, sixteen points per lane -- rides in state 1's search launch): losses and
    gradients equal the in-pass chain's over three announced steps (fixed parameters, deterministic statistics)."""
    from mpa_amd.models.repsurf.pointnet2_part_seg_msg import get_model, get_loss
    from mpa_amd.runtime import GraphedTrainStep
    B, N, NC = 2, 4096, 13
    monkeypatch.setattr(ops, "_fps_start", lambda B_, N_, device, start_idx=None: (
        (torch.arange(B_, device=device) * 5 + 1) % N_ if start_idx is None else start_idx.to(device)))
    old = ops.set_deterministic(True)
    try:
        g = torch.Generator().manual_seed(3)
        label = torch.zeros(B, 1, 16)
        label[:, 0, 2] = 1
        batches = [(unit_cloud(B, N, seed=s).transpose(1, 2).contiguous().cuda(), label.cuda(),
                    torch.randint(0, NC, (B, N), generator=g).cuda()) for s in (21, 22)]

        def compute_loss(model, crit, x, lab, tgt):
            return crit(model(x, lab)[0].reshape(-1, NC), tgt.reshape(-1))

        def run(prefetch):
            torch.manual_seed(0)
            model = fill_state(get_model(NC), seed=2).cuda().train()
            model.drop1.p = 0.0
            step = GraphedTrainStep(model, get_loss(), batches[0], lr=0.0, compute_loss=compute_loss,
                                    prefetch_geometry=prefetch)
            assert (step.prefetch is not None) == prefetch
            out = []
            try:
                for t, bi in enumerate((0, 1, 0)):
                    loss = step(*batches[bi], next_batch=batches[1 - bi] if t < 2 else None)
                    torch.cuda.synchronize()
                    out.append((float(loss.detach()), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
            finally:
                step.close()
            return out

        for t, ((l0, g0), (l1, g1)) in enumerate(zip(run(False), run(True))):
            # (`upsample` sums a fine point's contributors in the inverted table's list order: rows of up to 32 entries are
            # sorted by the build, so the forward is reproducible run to run -- tools/determinism_probe.py -- but a longer
            # row keeps the order its atomic cursor gave it, ~1e-6 at the first decoder stage, which the feature-space
            # searches behind it amplify: losses are compared at 5e-5 relative, not bit for bit)
            assert abs(l0 - l1) < 5e-5 * max(1.0, abs(l0)), (t, l0, l1)
            gmax = max(float(v.abs().max()) for v in g0.values())
            for n in g0:
                err = float((g0[n] - g1[n]).norm() / g0[n].norm().clamp_min(1e-12))
                # (parameter gradients are sums with float atomics -- the xyz branch's over 8192 rows, the split-K
                # reductions: two runs of the SAME step differ by ~1e-5 absolute on gradients of 1e-2)
                # (and, through the upsample's entry order, of the forward itself: wrong geometry would be O(1) here)
                assert err < 2e-2 or float((g0[n] - g1[n]).abs().max()) < 1e-3 * gmax, (t, n, err)
    finally:
        ops.set_deterministic(old)
