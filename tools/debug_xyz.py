import os, sys, math
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import mpa_amd
from mpa_amd import ops
from param_fill import randn
g = dict(np.load(os.path.join(ROOT, "tests/golden/blocks.npz")))
dev = "cuda"
xyz = torch.from_numpy(g["geo/xyz"]).to(dev); fps = torch.from_numpy(g["geo/fps"]).to(dev); idx = torch.from_numpy(g["geo/idx"]).to(dev)
center = ops.index_points(xyz, fps)
C = 64
torch.manual_seed(0)
Ws = [torch.randn(C, 3, device=dev) * 0.5, torch.randn(C, device=dev) * 0.1, torch.randn(C, 3, device=dev) * 0.5, torch.randn(C, device=dev) * 0.1,
      torch.randn(C, 3, device=dev) * 0.5, torch.randn(C, device=dev) * 0.1]
gout = torch.randn(2, 128, C, device=dev)
def ref(dt):
    W = [w.to(dt).requires_grad_(True) for w in Ws]
    x = xyz.to(dt); c = center.to(dt)
    rel = ops.index_points(xyz, idx).to(dt) - c.unsqueeze(2)
    q = torch.nn.functional.linear(c, W[0], W[1]).unsqueeze(2)
    k = torch.nn.functional.linear(rel, W[2], W[3]); v = torch.nn.functional.linear(rel, W[4], W[5])
    a = torch.softmax((q - k) / math.sqrt(C), dim=2)
    a = a - a.sum(2, keepdim=True)
    t = a * v
    ctx, am = t.max(2)
    ctx.backward(gout.to(dt))
    return ctx.detach(), am, [w.grad for w in W], t.detach()
W = [w.clone().requires_grad_(True) for w in Ws]
out = ops.diffattn_xyz(xyz, center, idx, *W)
out.backward(gout)
for dt in (torch.float32, torch.float64):
    ctx, am, gr, t = ref(dt)
    print(dt, "fwd err", (out.detach().double() - ctx.double()).abs().max().item())
    for n, a_, b_ in zip("Wq bq Wk bk Wv bv".split(), W, gr):
        print("   ", n, "grad err", (a_.grad.double() - b_.double()).abs().max().item(), "max", b_.abs().max().item())
    # near ties
    top2 = t.double().topk(2, dim=2)[0]
    gap = (top2[:, :, 0] - top2[:, :, 1])
    print("   min gap", gap.min().item(), "count gap<1e-6", int((gap < 1e-6).sum()), "of", gap.numel())
