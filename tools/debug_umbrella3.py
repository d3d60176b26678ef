import sys, torch, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import mpa_amd
from mpa_amd import ops
from mpa_amd.modules import repsurface_utils as RS
from oracle import ref_cpu as R
from param_fill import fill_state, unit_cloud
xyz = unit_cloud(2, 1024, seed=4242)
fa = ops.umbrella_features(xyz.cuda(), 9, return_dist=True).cpu()
fb = fill_state(R.UmbrellaSurfaceConstructor(9, 10, return_dist=True, random_inv=False), seed=21).features(xyz)
bad = ((fa - fb).abs() > 1e-4).any(-1).any(-1)
print("points differing:", int(bad.sum()), np.argwhere(bad.numpy())[:5].tolist())
ta = RS.group_by_umbrella(xyz.cuda(), xyz.cuda(), 9).cpu(); tb = R.group_by_umbrella(xyz, xyz, 9)
print("triangles differ at points:", int(((ta - tb).abs() > 0).flatten(2).any(-1).sum()))
b, n = np.argwhere(bad.numpy())[0]
idx = R.knn_point(9, xyz, xyz)[1]
rel = xyz[b, idx[b, n, 1:]] - xyz[b, n]
print("point", b, n, "keys cpu:", (torch.atan2(rel[:, 1], rel[:, 0]) / (2 * np.pi) + .5).tolist())
print("got group chan0..2", fa[b, n, :, :3].tolist()); print("ref", fb[b, n, :, :3].tolist())
print("got normal", fa[b, n, :, 6:9].tolist()); print("ref normal", fb[b, n, :, 6:9].tolist())
