"""Part-seg training through GraphedTrainStep with a finite check after every replay; stops at the
first non-finite value and names the parameters whose gradients are affected (development tool)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpa_amd
from mpa_amd.models.repsurf.pointnet2_part_seg_msg import get_model, get_loss
from mpa_amd.runtime import GraphedTrainStep
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N = 2048
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1234)
x = torch.rand(B, N, 3, generator=g) * 2 - 1
x = x - x.mean(1, keepdim=True)
x = (x / x.norm(dim=-1).max(dim=1)[0].view(B, 1, 1)).transpose(1, 2).contiguous().to(dev)
label = torch.zeros(B, 1, 16); label[torch.arange(B), 0, torch.randint(0, 16, (B,), generator=g)] = 1
label = label.to(dev)
target = torch.randint(0, 50, (B, N), generator=g).to(dev)
torch.manual_seed(0)
model = get_model(50).to(dev).train()
crit = get_loss()
def compute_loss(model, crit, x, label, target):
    pred, _ = model(x, label)
    return crit(pred.reshape(-1, 50), target.reshape(-1))
step = GraphedTrainStep(model, crit, (x, label, target), lr=1e-3, compute_loss=compute_loss)
if os.environ.get("FREEZE_FPS") == "1":
    step.feeder.frozen = True
    step.feeder.refill = lambda: None
for it in range(steps):
    loss = step(x, label, target)
    torch.cuda.synchronize()
    gbad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    pbad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    print("step %d loss %.5f bad grads %d %s bad params %d %s" % (it, float(loss.detach()), len(gbad), gbad[:6], len(pbad), pbad[:3]), flush=True)
    if gbad or pbad or not torch.isfinite(loss.detach()):
        sys.exit(3)
