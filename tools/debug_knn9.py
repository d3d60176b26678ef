import sys, torch, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import mpa_amd
from mpa_amd import ops
from oracle import c_oracle as co
from param_fill import unit_cloud
for N in (512, 1024, 2048):
    for K in (8, 9, 16):
        xyz = unit_cloud(2, N, seed=4242)
        d, i = ops.knn_point(K, xyz.cuda(), xyz.cuda())
        rd, ri = co.knn_point(K, xyz.numpy(), xyz.numpy())
        bad = (i.cpu().numpy() != ri).any(-1)
        print(N, K, "rows differing:", int(bad.sum()), "first", np.argwhere(bad)[:3].tolist())
        if bad.any():
            b, s = np.argwhere(bad)[0]
            print("   got", i[b, s].tolist(), "\n   ref", ri[b, s].tolist())
