"""diffattn backward: inverted-table (csr) path vs fp64 at large S with skewed neighbour lists (hub rows)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import mpa_amd  # noqa
from mpa_amd import ops
from test_gpu_ops import _diffattn_torch

for (B, N, S, C, K, skew) in [(1, 8192, 8192, 64, 8, 1), (1, 8192, 8192, 64, 8, 3), (2, 4096, 4096, 64, 8, 3), (1, 2048, 2048, 64, 8, 3),
                              (1, 8192, 8192, 64, 8, 6), (1, 12288, 12288, 64, 8, 3), (4, 1024, 512, 128, 8, 4)]:
    g = torch.Generator().manual_seed(N + S + skew)
    q = torch.randn(B, S, C, generator=g).cuda().requires_grad_()
    kv = torch.randn(B, N, 2 * C, generator=g).cuda().requires_grad_()
    idx = (torch.rand(B, S, K, generator=g) ** skew * N).long().clamp(max=N - 1).cuda()
    deg = torch.bincount(idx[0].reshape(-1), minlength=N)
    go = torch.randn(B, S, C, generator=g).cuda()
    out = ops.diffattn(q, kv, idx)
    out.backward(go)
    q64 = q.detach().double().requires_grad_()
    kv64 = kv.detach().double().requires_grad_()
    ref = _diffattn_torch(q64, kv64, idx)
    ref.backward(go.double())
    scale = kv64.grad.abs().max().item()
    err = (kv.grad.double() - kv64.grad).abs()
    bad_rows = (err > 1e-4 * scale).any(-1)[0]
    print("B%d N%d S%d C%d skew%d: max degree %d, rows>48: %d | fwd err %.2e | gkv frac bad %.2e, max err/scale %.2e, bad rows %d, their degrees %s"
          % (B, N, S, C, skew, int(deg.max()), int((deg > 48).sum()), (out.double() - ref).abs().max().item(),
             (err > 1e-4 * scale).float().mean().item(), err.max().item() / scale, int(bad_rows.sum()),
             deg[bad_rows][:12].tolist()))
