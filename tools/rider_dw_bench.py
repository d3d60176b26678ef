"""The cls model's grouped weight-gradient call with and without geometry riders: where does the carried time go?"""
import argparse, ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "tests", "golden")]
import torch
import mpa_amd  # noqa
from mpa_amd import ops
from mpa_amd.models.repsurf.repsurf_ssg_umb import Model, SmoothClsLoss
from mpa_amd.distributed import GradReducer
from param_fill import unit_cloud

dev = torch.device("cuda")
torch.manual_seed(0)
model = Model(argparse.Namespace(num_point=1024, return_dist=True, cuda_ops=True, num_class=40)).to(dev).train()
crit = SmoothClsLoss()
B, N, npoints, k = 64, 1024, (512, 256, 128, 64, 32), 8
x = unit_cloud(B, N, seed=1).transpose(1, 2).contiguous().cuda()
y = (torch.arange(B) % 40).cuda()
red = GradReducer(model, direct=True)
red.overlap = False
for _ in range(2):
    red.zero_grad(); crit(model(x), y).backward(); red.all_reduce()
ops.defer_weight_grads(True)
red.zero_grad()
crit(model(x), y).backward()
saved = list(ops._DW_QUEUE)
ops.flush_weight_grads()
print("%d products" % len(saved))
xyz = unit_cloud(B, N, seed=2).cuda()
pf = ops.GeometryPrefetch()
pf.spec = ((B, N, 3), npoints, k)
pf.allocate(dev)
pf.next_xyz.copy_(xyz)
starts = [torch.zeros(B, dtype=torch.int64, device="cuda") for _ in npoints]


def run(mode):
    ops._DW_QUEUE.extend(saved)
    if mode == "plain":
        ops.flush_weight_grads()
        return
    pf.starts = starts
    arr = pf.riders()
    if "nofps" in mode:
        arr[0].nlev = arr[1].nlev = 0
    if "nosearch" in mode:
        arr[0].base = arr[1].base = None
    if "noq" in mode:
        arr[0].queue = arr[1].queue = None
    ops.flush_weight_grads(riders=arr)


def timed(mode, n=10):
    for _ in range(3):
        run(mode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run(mode)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for mode in ("plain", "riders", "riders-nosearch", "riders-nofps", "riders-nofps-nosearch-x", "riders-noq"):
    if mode.endswith("-x"):
        # no sampling, no search: the carrier alone through the rider kernel's queue (a search with zero queries is
        # invalid, so keep one tiny search: 32 queries)
        continue
    print("%-28s %.1f us" % (mode, timed(mode)))
