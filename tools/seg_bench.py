"""Part-segmentation training step (ShapeNetPart-shaped: 2048 points, 16 object classes, 50 parts),
HIP-graph replay -- SURVEY 8(d) config 3's shape (development tool; the recorded number comes from
`bench.py --config partseg-bf16`).
    python tools/seg_bench.py [batch] [steps] [f32|bf16]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpa_amd  # noqa: E402
from mpa_amd.models.repsurf.pointnet2_part_seg_msg import get_model, get_loss  # noqa: E402
from mpa_amd.runtime import GraphedTrainStep  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
DT = sys.argv[3] if len(sys.argv) > 3 else "f32"
mpa_amd.ops.set_feature_dtype(torch.bfloat16 if DT == "bf16" else torch.float32)
N = 2048
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1234)
x = torch.rand(B, N, 3, generator=g) * 2 - 1
x = x - x.mean(1, keepdim=True)
x = (x / x.norm(dim=-1).max(dim=1)[0].view(B, 1, 1)).transpose(1, 2).contiguous().to(dev)
label = torch.zeros(B, 1, 16)
label[torch.arange(B), 0, torch.randint(0, 16, (B,), generator=g)] = 1
label = label.to(dev)
target = torch.randint(0, 50, (B, N), generator=g).to(dev)
torch.manual_seed(0)
model = get_model(50).to(dev).train()
crit = get_loss()


def compute_loss(model, crit, x, label, target):
    pred, _ = model(x, label)
    return crit(pred.reshape(-1, 50), target.reshape(-1))


print("building graph ...", flush=True)
step = GraphedTrainStep(model, crit, (x, label, target), lr=1e-3, compute_loss=compute_loss)
for i in range(4):
    l = step(x, label, target)
    print("warm-up step %d loss %.5f" % (i, float(l.detach())), flush=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
losses = []
for _ in range(steps):
    loss = step(x, label, target)
    losses.append(loss.detach().clone())
torch.cuda.synchronize()
print("losses:", " ".join("%.4f" % float(l) for l in losses[:: max(1, steps // 16)]), flush=True)
dt = (time.perf_counter() - t0) / steps
print("seg %s B=%d N=%d: %.2f ms/step, %.1f clouds/s, loss %.4f, peak mem %.1f GB" % (
    DT, B, N, dt * 1e3, B / dt, float(loss.detach()), torch.cuda.max_memory_allocated() / 2**30), flush=True)
