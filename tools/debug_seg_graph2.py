"""Part-seg HIP-graph replay determinism probe (development tool): every module output produced
during capture is kept alive; after each replay the tensors are checksummed and compared with the
previous replay (FPS starts frozen, optimizer skipped: everything before the dropout must repeat)."""
import os, sys
import torch
os.environ["MPA_DEBUG_SKIP_OPT"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpa_amd
from mpa_amd.models.repsurf.pointnet2_part_seg_msg import get_model, get_loss
from mpa_amd.runtime import GraphedTrainStep
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
PIN = os.environ.get("PIN", "1") == "1"
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
kept = []
def hook(name):
    def f(mod, inp, out):
        if torch.cuda.is_current_stream_capturing() and PIN:
            outs = out if isinstance(out, (tuple, list)) else (out,)
            for i, o in enumerate(outs):
                if torch.is_tensor(o):
                    kept.append(("%s[%d]" % (name, i), o.detach()))
    return f
for n, m in model.named_modules():
    m.register_forward_hook(hook(n))
def compute_loss(model, crit, x, label, target):
    pred, _ = model(x, label)
    return crit(pred.reshape(-1, 50), target.reshape(-1))
step = GraphedTrainStep(model, crit, (x, label, target), lr=1e-3, compute_loss=compute_loss)
step.feeder.frozen = True
step.feeder.refill = lambda: None
print("kept %d tensors" % len(kept), flush=True)
prev = None
for it in range(steps):
    loss = step(x, label, target)
    torch.cuda.synchronize()
    sums = [(n, float(t.double().sum()) if t.is_floating_point() else int(t.sum())) for n, t in kept]
    nonfin = [n for n, t in kept if t.is_floating_point() and not torch.isfinite(t).all()]
    gbad = sum(1 for p in model.parameters() if p.grad is not None and not torch.isfinite(p.grad).all())
    good = [n for n, p in model.named_parameters() if p.grad is not None and torch.isfinite(p.grad).all()]
    if gbad:
        print("   parameters with finite gradients (%d): %s" % (len(good), good), flush=True)
    print("replay %d loss %.5f non-finite kept outputs %d %s bad grads %d" % (it, float(loss.detach()), len(nonfin), nonfin[:3], gbad), flush=True)
    if prev is not None:
        diff = [n for (n, a), (_, b) in zip(sums, prev) if a != b and not (a != a and b != b)]
        print("   outputs whose checksum changed vs previous replay: %d, first: %s" % (len(diff), diff[:5]), flush=True)
    prev = sums
    if nonfin or gbad:
        sys.exit(3)
