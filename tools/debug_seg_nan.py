"""Finds the first non-finite tensor in a part-seg training step (development tool)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpa_amd
from mpa_amd.models.repsurf.pointnet2_part_seg_msg import get_model, get_loss
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
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
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
bad = []
def hook(name):
    def f(mod, inp, out):
        outs = out if isinstance(out, (tuple, list)) else (out,)
        for o in outs:
            if torch.is_tensor(o) and o.is_floating_point() and not torch.isfinite(o).all():
                bad.append(name)
    return f
for n, m in model.named_modules():
    m.register_forward_hook(hook(n))
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    opt.zero_grad(set_to_none=True)
    pred, _ = model(x, label)
    loss = crit(pred.reshape(-1, 50), target.reshape(-1))
    loss.backward()
    gbad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print("step %d loss %.5f  first bad fwd module: %s  bad grads: %d %s" % (it, loss.item(), bad[:1], len(gbad), gbad[:3]), flush=True)
    if bad or gbad:
        break
    opt.step()
