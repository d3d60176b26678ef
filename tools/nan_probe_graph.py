"""Development probe: the captured bf16 part-seg step; after each replay lists which parameter gradients are
non-finite (graph replay has no hooks: the pattern over the layers locates the source)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpa_amd  # noqa: E402
from mpa_amd import ops  # noqa: E402
from mpa_amd.models.repsurf.pointnet2_part_seg_msg import get_model, get_loss  # noqa: E402
from mpa_amd.runtime import GraphedTrainStep  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
DT = sys.argv[2] if len(sys.argv) > 2 else "bf16"
ops.set_feature_dtype(torch.bfloat16 if DT == "bf16" else torch.float32)
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


def compute_loss(model, crit, x, label, target):
    pred, _ = model(x, label)
    return crit(pred.reshape(-1, 50), target.reshape(-1))


step = GraphedTrainStep(model, get_loss(), (x, label, target), lr=1e-3, compute_loss=compute_loss)
import mpa_amd.runtime as rt
rt._SKIP_OPT = True          # gradients only: parameters stay put, every replay should give the same numbers
prev = None
for it in range(4):
    loss = step(x, label, target)
    torch.cuda.synchronize()
    bad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    flat = step.reducer.buckets[0]["flat"].clone()
    same = None if prev is None else float((flat - prev).abs().max())
    print("replay %d loss %.5f non-finite grads: %d %s ; max |grad - previous replay| %s" % (
        it, float(loss), len(bad), bad[:12], same), flush=True)
    if bad and it == 1:
        good = [n for n, p in model.named_parameters() if p.grad is not None and torch.isfinite(p.grad).all()]
        print("finite:", good, flush=True)
    prev = flat
step.close()
