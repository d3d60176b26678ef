import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
import mpa_amd
from mpa_amd import ops
g = np.load("/root/repo/tests/golden/umbrella.npz")
xyz = torch.from_numpy(g["xyz"]).cuda()
f = ops.umbrella_features(xyz, 9, return_dist=True).cpu().numpy()
ref = g["features"]
print("nan in ref", np.isnan(ref).sum(), "nan in got", np.isnan(f).sum())
err = np.abs(f - ref)
for c in range(10):
    print("channel", c, "max err", np.nanmax(err[..., c]))
bad = np.argwhere(err > 1e-4)
print("count > 1e-4:", len(bad), bad[:10])
for b, n, gi, c in bad[:3]:
    print("point", b, n, "got", f[b, n, gi], "ref", ref[b, n, gi])
idx = ops.knn_point(9, xyz, xyz)[1]
print("idx of point 6:", idx[0, 6].tolist(), "dup equal:", torch.equal(xyz[0, 3], xyz[0, 7]))
rel = (xyz[0, idx[0, 6, 1:]] - xyz[0, 6]).cpu()
key = torch.atan2(rel[:, 1], rel[:, 0]) / 6.283185307179586 + 0.5
print("rel:", rel.tolist()); print("keys:", key.tolist())
order = key.argsort(stable=True); print("order", order.tolist())
srt = rel[order]
for i in range(8):
    a, b = srt[i], srt[(i + 1) % 8]
    print(i, "cross", torch.linalg.cross(a, b).tolist())
