"""Part-seg training with the graph-mode components run eagerly (direct gradients, deferred grouped
dW, FlatAdam), checking for non-finite values after every step (development tool)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpa_amd
from mpa_amd import ops
from mpa_amd.models.repsurf.pointnet2_part_seg_msg import get_model, get_loss
from mpa_amd.distributed import GradReducer
from mpa_amd.optim import FlatAdam
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
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
red = GradReducer(model, direct=True); red.overlap = False
opt = None
defer = os.environ.get("DEFER", "1") == "1"
for it in range(steps):
    red.zero_grad()
    pred, _ = model(x, label)
    loss = crit(pred.reshape(-1, 50), target.reshape(-1))
    if defer and it > 0:
        ops.defer_weight_grads(True)
    loss.backward()
    if defer and it > 0:
        ops.flush_weight_grads(); ops.defer_weight_grads(False)
    red.all_reduce()
    if opt is None:
        opt = FlatAdam(red, 1e-3)
    torch.cuda.synchronize()
    gbad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    pbad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    gmax = max(float(b["flat"].abs().max()) for b in red.buckets)
    print("step %d loss %.5f  |grad|max %.3e  bad grads %d %s  bad params %d" % (it, loss.item(), gmax, len(gbad), gbad[:4], len(pbad)), flush=True)
    if gbad or pbad or not torch.isfinite(loss):
        sys.exit(3)
    opt.step()
