import sys, torch
sys.path.insert(0, "/root/repo")
import mpa_amd
from mpa_amd import ops
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
    want.append(gy.double().t() @ x.double())
ops.defer_weight_grads(True)
try:
    ops._DW_QUEUE.extend(queue)
    ops.flush_weight_grads()
finally:
    ops.defer_weight_grads(False)
torch.cuda.synchronize()
for (M, N, K), q, w in zip(shapes, queue, want):
    d = (q[4].double() - w).abs()
    d = torch.nan_to_num(d, nan=1e9)
    bad = (d > 0).nonzero()
    print((M, N, K), "max err", d.max().item(), "bad", bad.shape[0], "of", M * N, "first", bad[:3].tolist(),
          "rows", sorted(set((bad[:, 0] % 64).tolist()))[:12], "cols", sorted(set((bad[:, 1] % 64).tolist()))[:12],
          "tiles", sorted(set(((bad[:, 0] // 64) * 100 + bad[:, 1] // 64).tolist()))[:10])
