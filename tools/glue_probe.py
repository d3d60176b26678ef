"""Which host lines launch the ATen (non-libmpa) device kernels of one training pass?  Runs a GraphedTrainStep's
eager pass (same code the graph captures) under torch.profiler with Python stacks and prints, per ATen op and
calling line, the launch count and device time.   python tools/glue_probe.py [cls|seg] [f32|bf16]"""
import os
import sys
from argparse import Namespace

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpa_amd  # noqa: E402
from mpa_amd.runtime import GraphedTrainStep  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cls"
DT = sys.argv[2] if len(sys.argv) > 2 else "f32"
mpa_amd.ops.set_feature_dtype(torch.bfloat16 if DT == "bf16" else torch.float32)
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
if which == "cls":
    from mpa_amd.models.repsurf.repsurf_ssg_umb import Model, SmoothClsLoss
    B, N = 64, 1024
    x = (torch.rand(B, 3, N, generator=g) * 2 - 1).to(dev)
    y = torch.randint(0, 40, (B,), generator=g).to(dev)
    model = Model(Namespace(num_point=N, return_dist=True, cuda_ops=True, num_class=40)).to(dev).train()
    step = GraphedTrainStep(model, SmoothClsLoss(), (x, y), lr=1e-3)
else:
    from mpa_amd.models.repsurf.pointnet2_part_seg_msg import get_loss, get_model
    B, N = 32, 2048
    x = (torch.rand(B, 3, N, generator=g) * 2 - 1).to(dev)
    label = torch.zeros(B, 1, 16)
    label[torch.arange(B), 0, torch.randint(0, 16, (B,), generator=g)] = 1
    label = label.to(dev)
    target = torch.randint(0, 50, (B, N), generator=g).to(dev)
    model = get_model(50).to(dev).train()

    def compute_loss(model, crit, x, label, target):
        pred, _ = model(x, label)
        return crit(pred.reshape(-1, 50), target.reshape(-1))
    step = GraphedTrainStep(model, get_loss(), (x, label, target), lr=1e-3, compute_loss=compute_loss)

torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step._fwd_bwd()
    torch.cuda.synchronize()

print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_device_time_total", row_limit=90,
                                                          max_name_column_width=60, max_shapes_column_width=70))

# per calling line: the ATen ops that launch device work
import collections
agg = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.self_device_time_total <= 0:
        continue
    stack = [s for s in (ev.stack or []) if "mpa_amd" in s or "markov" in s or "bench" in s]
    agg[(ev.name, " <- ".join(s.split("/")[-1] for s in stack[:3]))] += 1
print("\nATen launches by calling line (count per pass):")
for (name, where), n in sorted(agg.items(), key=lambda kv: -kv[1]):
    print("%3d  %-28s %s" % (n, name, where))
