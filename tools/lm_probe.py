"""LocalMerge(64,64,8) self-level block vs oracle/ref_cpu.py at large N, several seeds: how many gradient entries differ."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")]
import torch
import mpa_amd  # noqa
from mpa_amd import ops
from mpa_amd.modules import pointnet2_utils as P2
from oracle import ref_cpu as Rf
from param_fill import fill_state, unit_cloud, randn

for N, seed in [(4096, 1), (8192, 8192), (8192, 2), (8192, 3), (16384, 16384), (16384, 5)]:
    xyz = unit_cloud(1, N, seed=seed)
    feat = randn((1, N, 64), seed=seed + 1)
    w = randn((1, N, 64), seed=seed + 2)
    cpu = fill_state(Rf.LocalMergeSeg(64, 64, 8, residual=False), seed=3).train()
    fc = feat.clone().requires_grad_(True)
    oc, _, oidx, odist = cpu(xyz=xyz, base_xyz=xyz, normal=None, feature=fc)
    (oc * w).sum().backward()
    gpu = fill_state(P2.LocalMerge(64, 64, 8, residual=False), seed=3).cuda().train()
    fg = feat.cuda().requires_grad_(True)

    class O:
        chain = None
        dist, idx = odist.cuda(), oidx.cuda()

        def search(self, k, feature, query):
            return (self.dist, self.idx), ops.knn_point(k, feature, query)[1]

    og = gpu(xyz=xyz.cuda(), base_xyz=xyz.cuda(), normal=None, feature=fg, geometry=O())[0]
    (og * w.cuda()).sum().backward()
    ferr = (og.detach().cpu() - oc.detach()).abs().max().item()
    gerr = (fg.grad.cpu() - fc.grad).abs()
    scale = fc.grad.abs().max().item()
    bad = gerr > 1e-4 * scale
    rows = bad.any(-1)[0]
    print("N=%d seed=%d: fwd max err %.2e (|out| max %.2f) | grad: frac bad %.2e, rel L2 %.2e, max err %.2e (scale %.2f), bad rows %d, bad cols/row mean %.1f"
          % (N, seed, ferr, oc.abs().max().item(), bad.float().mean().item(), (gerr.norm() / fc.grad.norm()).item(), gerr.max().item(), scale,
             int(rows.sum()), float(bad[0][rows].float().sum(-1).mean()) if rows.any() else 0), flush=True)
