"""Stand-alone timing of the geometry riders against the plain entry points (cls shapes: B=64, N=1024)."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "tests", "golden")]
import torch
import mpa_amd  # noqa
from mpa_amd import ops
from mpa_amd._lib import lib, GeoRider
from param_fill import unit_cloud

B, N, npoints, k = 64, 1024, (512, 256, 128, 64, 32), 8
xyz = unit_cloud(B, N, seed=1).cuda()
pf = ops.GeometryPrefetch()
pf.spec = ((B, N, 3), npoints, k)
pf.allocate(xyz.device)
starts = [torch.zeros(B, dtype=torch.int64, device="cuda") for _ in npoints]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def rider(which, fps=True, search=True):
    arr = pf._riders(xyz, starts)
    r = arr[which]
    if not fps:
        r.nlev = 0
    if not search:
        r.base = None
    return lambda: ops._launch("mpa_geo_rider_f32", ctypes.byref(r), ops._stream())


print("fps 1024->512 (mpa_fps_f32)            %.1f us" % timed(lambda: ops.farthest_point_sample(xyz, 512, start_idx=starts[0])))
x1 = pf.fps_xyz[0]
print("fps 512->256                           %.1f us" % timed(lambda: ops.farthest_point_sample(x1, 256, start_idx=starts[0])))
print("knn 1024 in 1024 (mpa_knn_f32)         %.1f us" % timed(lambda: (ops.clear_knn_memo(), ops.knn_point(8, xyz, xyz))))
print("knn 512 in 1024                        %.1f us" % timed(lambda: (ops.clear_knn_memo(), ops.knn_point(8, xyz, x1))))
print("rider0 alone: fps only                 %.1f us" % timed(rider(0, True, False)))
print("rider0 alone: search only              %.1f us" % timed(rider(0, False, True)))
print("rider0 alone: both                     %.1f us" % timed(rider(0, True, True)))
print("rider1 alone: chain only               %.1f us" % timed(rider(1, True, False)))
print("rider1 alone: search only              %.1f us" % timed(rider(1, False, True)))
print("rider1 alone: both                     %.1f us" % timed(rider(1, True, True)))
