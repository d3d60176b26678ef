"""GPU parity at the sizes of BASELINE.json's other configurations -- part segmentation (2048
points, batch 32), S3DIS-style blocks (4096 points, batch 16 per GPU) and the completion decoder
(1024 -> 16384 points, batch 8 per GPU).  Index work is compared bit for bit with the C oracle where
it finishes in seconds (full batches for FPS, a few clouds for the searches); feature work is
checked through size-independent properties and an independent fp32 torch formulation on the
device.  The reference has no model for the last two configurations: these are op-level
and wiring-level checks, "parity unpinned beyond op level"."""
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


@pytest.mark.parametrize("B,N,S", [(32, 2048, 1024), (16, 4096, 2048), (8, 8192, 4096)])
def test_fps_config_sizes_vs_oracle(ops, co, B, N, S):
    """Whole batches of the part-seg / S3DIS-block sizes (and the largest supported cloud): every
    sampled index equals the C oracle's."""
    xyz = unit_cloud(B, N, seed=N + B)
    start = (torch.arange(B) * 37) % N
    idx, sub = ops.farthest_point_sample(xyz.cuda(), S, start_idx=start, return_xyz=True)
    want = co.farthest_point_sample(xyz.numpy(), S, start.numpy())
    assert np.array_equal(idx.cpu().numpy(), want)
    assert np.array_equal(sub.cpu().numpy(), np.take_along_axis(xyz.numpy(), want[..., None], axis=1))


@pytest.mark.parametrize("B,S,N,C", [(4, 2048, 2048, 3), (2, 4096, 4096, 3), (1, 8192, 16384, 3), (1, 2048, 2048, 64),
                                     (1, 2048, 4096, 64)])
def test_knn_config_sizes_vs_oracle(ops, co, B, S, N, C):
    """Neighbour search at the part-seg, S3DIS-block and completion sizes: indices and distance bits."""
    base = unit_cloud(B, N, seed=N + C) if C == 3 else randn((B, N, C), seed=N + C)
    query = base[:, :S].contiguous() if C == 3 else randn((B, S, C), seed=S + C + 1, scale=0.7)
    dist, idx = ops.knn_point(8, base.cuda(), query.cuda())
    od, oi = co.knn_point(8, base.numpy(), query.numpy())
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(bits(dist.cpu().numpy()), bits(od))


def test_completion_decoder_chain(ops):
    """Config 5's op chain at full size (batch 8 per GPU): 1024 -> 2048 -> 4096 -> 8192 -> 16384 points,
    each step `knn_point` (coarse state in the fine state) + `upsample` (the last one beyond the
    inverted-table limit, i.e. on the atomic path); the fine clouds nest the
    coarse ones, as FPS-derived states do.  Each upsample is compared with an independent torch
    formulation on the device (index_add of the coarse rows / number of contributors), forward and
    the gradient w.r.t. the coarse features."""
    B, C, K = 8, 64, 8
    fine = unit_cloud(B, 16384, seed=77).cuda()
    feats = randn((B, 1024, C), seed=78).cuda()
    for S in (1024, 2048, 4096, 8192):
        Nf = 2 * S
        coarse_xyz, fine_xyz = fine[:, :S].contiguous(), fine[:, :Nf].contiguous()
        _, idx = ops.knn_point(K, fine_xyz, coarse_xyz)            # [B,S,K], distinct within a row
        p = feats.clone().requires_grad_(True)
        up = ops.upsample(p, idx, scale_ratio=2)
        assert up.shape == (B, Nf, C)
        flat = (idx + (torch.arange(B, device="cuda") * Nf)[:, None, None]).reshape(-1)
        tot = torch.zeros(B * Nf, C, device="cuda").index_add_(0, flat, p.detach()[:, :, None, :].expand(B, S, K, C).reshape(-1, C))
        # the reference's divisor: contributors whose channel-0 value is non-zero (uncovered rows of the
        # previous step are exactly 0 and count as contributors of nothing)
        nz = (p.detach()[:, :, 0] != 0).float()[:, :, None].expand(B, S, K).reshape(-1)
        cnt = torch.zeros(B * Nf, device="cuda").index_add_(0, flat, nz)
        want = (tot / cnt.clamp(min=1)[:, None]).view(B, Nf, C)
        assert torch.allclose(up, want, rtol=1e-5, atol=1e-6)
        cover = torch.zeros(B * Nf, device="cuda").index_add_(0, flat, torch.ones_like(nz))
        assert (up.detach().abs().sum(-1).view(-1)[cover == 0] == 0).all()            # uncovered fine points stay 0
        assert int((cover == 0).sum()) > 0
        g = randn((B, Nf, C), seed=S).cuda()
        up.backward(g)
        gw = (g / cnt.clamp(min=1).view(B, Nf, 1)).reshape(-1, C)[flat].view(B, S, K, C).sum(2)
        assert torch.allclose(p.grad, gw, rtol=1e-5, atol=1e-5)
        feats = up.detach()


def test_diffattn_at_completion_size(ops):
    """The attention op on a 16384-point state (batch 8): forward against the reference formulation
    evaluated with torch ops on the device, on one cloud."""
    B, N, C, K = 8, 16384, 64, 8
    q = randn((B, N, C), seed=1).cuda()
    kv = randn((B, N, 2 * C), seed=2).cuda()
    idx = torch.randint(0, N, (B, N, K), generator=torch.Generator().manual_seed(3)).cuda()
    k, v = kv[..., :C], kv[..., C:]
    out = ops.diffattn(q, kv, idx)
    b = 5
    kg, vg = k[b][idx[b]], v[b][idx[b]]                                  # [N,K,C]
    a = torch.softmax((q[b][:, None, :] - kg) / (C ** 0.5), dim=1)
    want = ((a - a.sum(1, keepdim=True)) * vg).max(dim=1)[0]
    assert torch.allclose(out[b], want, rtol=1e-4, atol=1e-5)


def test_partseg_wiring_at_4096_points(ops, monkeypatch):
    """Config 4's wiring: the part-seg encoder-decoder on 4096-point blocks (states 4096 -> 2048 -> 1024
    -> 512 -> 256; `Fuse` picks its target by state size instead of the reference's literal counts).
    Properties: shapes, finiteness of outputs and gradients, and -- in eval mode with fixed sampling
    starts -- a cloud's result does not depend on the rest of its batch."""
    from mpa_amd.models.repsurf.pointnet2_part_seg_msg import get_model
    monkeypatch.setattr(ops, "_fps_start", lambda B, N, device, start_idx=None: torch.zeros(B, dtype=torch.int64, device=device))
    B, N = 3, 4096
    x = unit_cloud(B, N, seed=11).transpose(1, 2).contiguous().cuda()
    label = torch.zeros(B, 1, 16, device="cuda")
    label[:, 0, 4] = 1
    torch.manual_seed(0)
    model = get_model(13).cuda().train()
    pred, _ = model(x, label)
    assert pred.shape == (B, N, 13) and torch.isfinite(pred).all()
    pred.square().mean().backward()
    used = [p.grad for p in model.parameters() if p.grad is not None]
    assert len(used) > 100 and all(torch.isfinite(g).all() for g in used)
    model.eval()
    with torch.no_grad():
        full, _ = model(x, label)
        one, _ = model(x[1:2].contiguous(), label[1:2].contiguous())
    assert torch.allclose(full[1:2], one, rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------------------- bf16 feature path at full size
def test_partseg_bf16_full_batch_step():
    """BASELINE configs[2] at its full size -- part segmentation, 2048 points, batch 32, bf16 features -- through
    the captured training step: finite and falling loss, every live parameter receives a finite fp32 gradient,
    logits fp32.  (Numerics against the fp32 fixtures: tests/test_gpu_bf16.py; parity unpinned for bf16.)"""
    import mpa_amd  # noqa: F401
    from mpa_amd import ops
    from mpa_amd.models.repsurf.pointnet2_part_seg_msg import get_model, get_loss
    from mpa_amd.runtime import GraphedTrainStep
    B, N = 32, 2048
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(11)
    x = unit_cloud(B, N, seed=11).transpose(1, 2).contiguous().to(dev)
    label = torch.zeros(B, 1, 16)
    label[torch.arange(B), 0, torch.randint(0, 16, (B,), generator=g)] = 1
    label = label.to(dev)
    target = torch.randint(0, 50, (B, N), generator=g).to(dev)
    torch.manual_seed(0)
    model = get_model(50).to(dev).train()

    def compute_loss(model, crit, x, label, target):
        pred, _ = model(x, label)
        assert pred.dtype == torch.float32 and pred.shape == (B, N, 50)
        return crit(pred.reshape(-1, 50), target.reshape(-1))

    with ops.feature_dtype(torch.bfloat16):
        step = GraphedTrainStep(model, get_loss(), (x, label, target), lr=1e-3, compute_loss=compute_loss)
        try:
            losses = [float(step(x, label, target).detach()) for _ in range(10)]
            torch.cuda.synchronize()
            live = [p for p in model.parameters() if p.grad is not None]
            assert len(live) > 400 and all(p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all() for p in live)
            assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
        finally:
            step.close()


def test_completion_chain_16384_bf16(ops):
    """BASELINE configs[4]'s op chain 1024 -> 2048 -> 4096 -> 8192 -> 16384 points (B = 8, C = 64) on bf16 features:
    kNN on the rounded features' exact values, `upsample` forward / backward and the difference attention on a
    16,384-point state -- each equal to the fp32 op on the same rounded inputs, rounded once (the reference has
    no model for this configuration and no bf16 numerics: op-level, parity unpinned)."""
    B, C, K = 8, 64, 8
    g = torch.Generator().manual_seed(3)
    xyz = unit_cloud(B, 16384, seed=5).cuda()
    feats = torch.randn(B, 1024, C, generator=g).to(torch.bfloat16).cuda()
    lvl = [16384 >> s for s in (4, 3, 2, 1, 0)]           # 1024 ... 16384
    cur16, cur32 = feats.clone().requires_grad_(True), feats.float().requires_grad_(True)
    f16, f32 = cur16, cur32
    for i in range(4):
        coarse, fine = xyz[:, :lvl[i]].contiguous(), xyz[:, :lvl[i + 1]].contiguous()
        idx = ops.knn_point(K, fine, coarse)[1]            # coarse rows list their K nearest fine points
        f16, f32 = ops.upsample(f16, idx), ops.upsample(f32, idx)
        assert f16.dtype == torch.bfloat16 and f16.shape == (B, lvl[i + 1], C)
        assert torch.equal(f16, f32.to(torch.bfloat16)), i
        f32 = f16.float()                                  # (keep the two chains on identical inputs)
    q = torch.randn(B, 16384, C, generator=g).to(torch.bfloat16).cuda()
    kv = torch.cat((f16, f16 * 0.5), 2).contiguous()
    nidx = ops.knn_point(K, xyz, xyz)[1]
    o16 = ops.diffattn(q, kv, nidx)
    o32 = ops.diffattn(q.float(), kv.float(), nidx)
    assert torch.equal(o16, o32.to(torch.bfloat16))
    # feature-space search on bf16 features = the fp32 search on their exact values (bit-exact indices)
    d16, i16 = ops.knn_point(K, f16[:, :4096].contiguous(), f16[:, :2048].contiguous())
    d32, i32 = ops.knn_point(K, f16[:, :4096].float().contiguous(), f16[:, :2048].float().contiguous())
    assert torch.equal(i16, i32) and torch.equal(d16, d32)
    # backward through one upsample at the largest state
    w = torch.randn(B, 16384, C, generator=g).to(torch.bfloat16).cuda()
    p16 = feats.new_zeros(B, 8192, C).copy_(torch.randn(B, 8192, C, generator=g)).requires_grad_(True)
    p32 = p16.detach().float().requires_grad_(True)
    idx = ops.knn_point(K, xyz, xyz[:, :8192].contiguous())[1]
    ops.upsample(p16, idx).backward(w)
    ops.upsample(p32, idx).backward(w.float())
    assert torch.equal(p16.grad, p32.grad.to(torch.bfloat16))


# ------------------------------------------------------------------- configs[3] / configs[4] as full-batch workloads
def _captured_steps(model, crit, batch, compute_loss, steps):
    from mpa_amd.runtime import GraphedTrainStep
    step = GraphedTrainStep(model, crit, batch, lr=1e-3, compute_loss=compute_loss)
    try:
        losses = [float(step(*batch).detach()) for _ in range(steps)]
        torch.cuda.synchronize()
    finally:
        step.close()
    return losses


def test_s3dis_full_batch_step(ops, monkeypatch):
    """BASELINE configs[3] at its per-GPU size -- the part-seg encoder-decoder wiring on 4096-point blocks (states 4096
    -> 2048 -> 1024 -> 512 -> 256), 13 classes, batch 16, fp32 -- through the captured training step: finite and
    falling loss, every live parameter receives a finite gradient; in eval mode with fixed sampling starts a block's
    logits do not depend on the rest of its batch.  (No reference model exists for this configuration: the blocks are
    pinned one by one in test_gpu_blocks.py, the wiring by the 2048-point golden model; "parity unpinned beyond op
    level")"""
    from mpa_amd.models.repsurf.pointnet2_part_seg_msg import get_model, get_loss
    B, N, NC = 16, 4096, 13
    g = torch.Generator().manual_seed(5)
    x = unit_cloud(B, N, seed=21).transpose(1, 2).contiguous().cuda()
    label = torch.zeros(B, 1, 16)
    label[:, 0, 0] = 1
    label = label.cuda()
    target = torch.randint(0, NC, (B, N), generator=g).cuda()
    torch.manual_seed(0)
    model = get_model(NC).cuda().train()

    def compute_loss(model, crit, x, label, target):
        pred, _ = model(x, label)
        assert pred.shape == (B, N, NC) and pred.dtype == torch.float32
        return crit(pred.reshape(-1, NC), target.reshape(-1))

    losses = _captured_steps(model, get_loss(), (x, label, target), compute_loss, 8)
    live = [p for p in model.parameters() if p.grad is not None]
    assert len(live) > 400 and all(torch.isfinite(p.grad).all() for p in live)
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    monkeypatch.setattr(ops, "_fps_start", lambda B, N, device, start_idx=None: torch.zeros(B, dtype=torch.int64, device=device))
    model.eval()
    with torch.no_grad():
        full, _ = model(x, label)
        one, _ = model(x[5:6].contiguous(), label[5:6].contiguous())
    # (after eight optimizer steps a few feature-space neighbourhoods sit on near-ties that the batch-size dependent
    # GEMM tiling can flip: the untrained-model form of this check at 1e-4 is test_partseg_wiring_at_4096_points)
    assert ((full[5:6] - one).norm() / one.norm()).item() < 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_completion_full_batch_step(ops, dtype):
    """BASELINE configs[4] at its per-GPU size -- the completion chain `upsample` + `LocalMerge`, 1024 -> 2048 -> 4096 ->
    8192 -> 16,384 points, batch 8 (131,072 rows in the last state: beyond the inverted-table and grouped-launch
    limits of the smaller configurations) -- through the captured training step, on fp32 and on bf16 features: finite
    and falling loss, every live parameter receives a finite fp32 gradient; in eval mode a cloud's prediction does
    not depend on the rest of its batch."""
    from mpa_amd.models.completion import CompletionDecoder, CoordinateLoss, sampling_order
    B, N = 8, 16384
    x = sampling_order(unit_cloud(B, N, seed=31).cuda(), start_idx=torch.zeros(B, dtype=torch.long))
    x = x.transpose(1, 2).contiguous()
    torch.manual_seed(0)
    model = CompletionDecoder().cuda().train()

    def compute_loss(model, crit, x):
        pred = model(x)
        assert pred.shape == (B, N, 3) and pred.dtype == torch.float32
        return crit(pred, x)

    with ops.feature_dtype(dtype):
        losses = _captured_steps(model, CoordinateLoss(), (x,), compute_loss, 8)
        live = [p for p in model.parameters() if p.grad is not None]
        assert len(live) > 100 and all(p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all() for p in live)
        assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
        model.eval()
        with torch.no_grad():
            full = model(x)
            one = model(x[3:4].contiguous())
        assert ((full[3:4] - one).norm() / one.norm()).item() < (1e-3 if dtype == torch.float32 else 3e-2)


def test_sampling_order_prefixes_are_fps_states(ops, co):
    """models/completion.py's geometry contract: after sampling_order() the S-point prefix of a cloud is the reference's
    FPS sample of S points (same first point), for every S -- checked against the C oracle on a 16,384-point cloud
    (the generic FPS kernel: beyond the register-resident kernel's 12,288 points)."""
    from mpa_amd.models.completion import sampling_order
    B, N = 2, 16384
    xyz = unit_cloud(B, N, seed=41)
    start = torch.tensor([5, 16000])
    got = sampling_order(xyz.cuda(), start_idx=start).cpu().numpy()
    want_idx = co.farthest_point_sample(xyz.numpy(), 4096, start.numpy())
    assert np.array_equal(got[:, :4096], np.take_along_axis(xyz.numpy(), want_idx[..., None], axis=1))
    idx_full = ops.farthest_point_sample(xyz.cuda(), N, start_idx=start).cpu().numpy()
    assert all(len(set(idx_full[b].tolist())) == N for b in range(B))          # a permutation: every point once


@pytest.mark.parametrize("N", [8192, 16384])
def test_completion_localmerge_vs_oracle(ops, N):
    """The completion chain's transition block at its two largest states -- LocalMerge(64, 64, 8) of an N-point state
    in itself (coordinate search, feature-space search, three attention streams, fc2; reference
    modules/pointnet2_utils.py:427-477) -- on one cloud against oracle/ref_cpu.py: neighbour indices bit-exact,
    features <= 1e-4 in fp32, gradient w.r.t. the input features <= 1e-4 x its scale; on bf16 features against the
    same fp32 oracle at the bf16 storage tolerance (relative L2 <= 2e-2)."""
    from mpa_amd.modules import pointnet2_utils as P2
    from oracle import ref_cpu as R
    from param_fill import fill_state
    xyz = unit_cloud(1, N, seed=N)
    feat = randn((1, N, 64), seed=N + 1)
    w = randn((1, N, 64), seed=N + 2)
    cpu = fill_state(R.LocalMergeSeg(64, 64, 8, residual=False), seed=3).train()
    fc = feat.clone().requires_grad_(True)
    oc, _, oidx, odist = cpu(xyz=xyz, base_xyz=xyz, normal=None, feature=fc)
    (oc * w).sum().backward()
    gpu = fill_state(P2.LocalMerge(64, 64, 8, residual=False), seed=3).cuda().train()
    fg = feat.cuda().requires_grad_(True)
    gdist, gidx = ops.knn_point(8, xyz.cuda(), xyz.cuda())

    class OracleNeighbourhoods:          # the block under test with the oracle's coordinate neighbourhoods (ties: below)
        chain = None
        dist, idx = odist.cuda(), oidx.cuda()

        def search(self, k, feature, query):
            return (self.dist, self.idx), ops.knn_point(k, feature, query)[1]

    og = gpu(xyz=xyz.cuda(), base_xyz=xyz.cuda(), normal=None, feature=fg, geometry=OracleNeighbourhoods())[0]
    # distances bit for bit; indices equal wherever the distance is not shared with a neighbouring rank (at these
    # densities a few of the N x 8 distances tie exactly in fp32, and torch.topk -- the oracle -- orders ties
    # arbitrarily; the lowest-index-first rule itself is pinned against the C oracle above)
    assert np.array_equal(bits(gdist.cpu().numpy()), bits(odist.numpy()))
    od = odist.numpy()
    tie = np.zeros(od.shape, bool)
    tie[..., 1:] |= od[..., 1:] == od[..., :-1]
    tie[..., :-1] |= od[..., 1:] == od[..., :-1]
    same = gidx.cpu().numpy() == oidx.numpy()
    assert (same | tie).all() and same.mean() > 0.999
    assert (og.detach().cpu() - oc.detach()).abs().max().item() < 1e-4 * max(1.0, oc.detach().abs().max().item())
    (og * w.cuda()).sum().backward()
    # gradient: 1e-4 x its scale on all but a handful of entries -- with N x 64 x 3 max-over-K selections a few sit on
    # near-ties that a 1e-6 forward difference flips, which re-routes an O(1) gradient -- and 1e-3
    # relative L2 overall
    gerr = (fg.grad.cpu() - fc.grad).abs()
    frac, grel = (gerr > 1e-4 * max(1.0, fc.grad.abs().max().item())).float().mean().item(), (gerr.norm() / fc.grad.norm()).item()
    print("LocalMerge N=%d: gradient entries beyond 1e-4 x scale %.2e, relative L2 %.2e" % (N, frac, grel))
    # (tools/lm_probe.py over several seeds: 0 ... 280 of the N rows carry such entries, 18-60 channels each -- whole
    # re-routed rows, seed dependent -- while every other entry agrees to 1e-5: selection flips, not arithmetic)
    assert frac < 2e-2 and grel < 1e-2, (frac, grel)
    # bf16 features: the same block on bf16-rounded inputs against the fp32 block (just checked against the oracle) on
    # those rounded inputs -- equal neighbourhoods on both sides (the searches run on the rounded features' exact values)
    f32r = feat.to(torch.bfloat16).float().cuda().requires_grad_(True)
    o32r = gpu(xyz=xyz.cuda(), base_xyz=xyz.cuda(), normal=None, feature=f32r, geometry=OracleNeighbourhoods())[0]
    (o32r * w.cuda()).sum().backward()
    with ops.feature_dtype(torch.bfloat16):
        gpu16 = fill_state(P2.LocalMerge(64, 64, 8, residual=False), seed=3).cuda().train()
        f16 = feat.to(torch.bfloat16).cuda().requires_grad_(True)
        o16 = gpu16(xyz=xyz.cuda(), base_xyz=xyz.cuda(), normal=None, feature=f16, geometry=OracleNeighbourhoods())[0]
        assert o16.dtype == torch.bfloat16
        rel = ((o16.detach().float() - o32r.detach()).norm() / o32r.detach().norm()).item()
        assert rel < 2e-2, rel
        (o16.float() * w.cuda()).sum().backward()
        grel = ((f16.grad.float() - f32r.grad).norm() / f32r.grad.norm()).item()
        assert grel < 0.15, grel              # (max-over-K selections flipped by the storage rounding: measured 5-9e-2)
