"""ATen ops (not libmpa kernels) of one eager part-seg train step after the flat parameter buffers
exist, with counts and device time (development tool)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpa_amd
from mpa_amd.models.repsurf.pointnet2_part_seg_msg import get_model, get_loss
from mpa_amd.runtime import GraphedTrainStep
B, N = 8, 2048
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1234)
x = (torch.rand(B, 3, N, generator=g) * 2 - 1).to(dev)
label = torch.zeros(B, 1, 16); label[:, 0, 3] = 1; label = label.to(dev)
target = torch.randint(0, 50, (B, N), generator=g).to(dev)
torch.manual_seed(0)
model = get_model(50).to(dev).train()
def compute_loss(model, crit, x, label, target):
    pred, _ = model(x, label)
    return crit(pred.reshape(-1, 50), target.reshape(-1))
step = GraphedTrainStep(model, get_loss(), (x, label, target), lr=1e-3, compute_loss=compute_loss)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step._fwd_bwd(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "device_time_total", None) or getattr(e, "cuda_time_total", 0)
    if dt > 0 and e.key.startswith("aten::"):
        rows.append((dt, e.count, e.key, str(e.input_shapes)[:100]))
rows.sort(reverse=True)
import collections
by = collections.defaultdict(lambda: [0, 0.0])
for dt, cnt, key, shp in rows:
    by[key][0] += cnt; by[key][1] += dt
for k, (c, d) in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-40s %5d calls %9.1f us" % (k, c, d))
print()
for dt, cnt, key, shp in rows[:45]:
    print("%8.1f us %4d  %-22s %s" % (dt, cnt, key, shp))
