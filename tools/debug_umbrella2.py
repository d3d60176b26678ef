import sys, torch, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import mpa_amd
from mpa_amd import ops
from mpa_amd.modules import repsurface_utils as RS
from oracle import ref_cpu as R
from param_fill import fill_state, unit_cloud
for N in (512, 1024):
    xyz = unit_cloud(2, N, seed=4242)
    for mode in ("eval", "train"):
        for rinv in (False, True):
            a = fill_state(RS.UmbrellaSurfaceConstructor(9, 10, return_dist=True, random_inv=rinv), seed=21).cuda()
            b = fill_state(R.UmbrellaSurfaceConstructor(9, 10, return_dist=True, random_inv=rinv), seed=21)
            getattr(a, mode)(); getattr(b, mode)()
            torch.manual_seed(5); oa = a(xyz.transpose(1, 2).contiguous().cuda())
            torch.manual_seed(5); ob = b(xyz.transpose(1, 2).contiguous())
            fa = ops.umbrella_features(xyz.cuda(), 9, return_dist=True).cpu(); fb = fill_state(R.UmbrellaSurfaceConstructor(9, 10, return_dist=True, random_inv=False), seed=21).features(xyz)
            print(N, mode, rinv, "out diff %.3e (max %.2f)  feature diff %.3e" % (float((oa.detach().cpu() - ob.detach()).abs().max()), float(ob.abs().max()), float((fa - fb).abs().max())))
