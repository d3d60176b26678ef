import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo/tests/golden")
import mpa_amd
from mpa_amd import ops
from param_fill import unit_cloud
B, N, C = 64, 1024, 64
dev = torch.device("cuda")
xyz = unit_cloud(B, N, seed=1).to(dev)
idx = ops.knn_point(8, xyz, xyz)[1]
W = [torch.randn(C, 3, device=dev, requires_grad=True) for _ in range(3)]
b = [torch.randn(C, device=dev, requires_grad=True) for _ in range(3)]
for _ in range(20):
    out = ops.diffattn_xyz(xyz, xyz, idx, W[0], b[0], W[1], b[1], W[2], b[2])
    g = torch.randn_like(out)
    torch.autograd.grad(out, W + b, g)
torch.cuda.synchronize()
