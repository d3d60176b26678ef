"""Development probe: eager bf16 part-seg training steps with finiteness checks on every module output,
every gradient and every parameter; prints the first offender.  python tools/nan_probe.py [B] [steps] [f32|bf16]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpa_amd  # noqa: E402
from mpa_amd import ops  # noqa: E402
from mpa_amd.models.repsurf.pointnet2_part_seg_msg import get_model, get_loss  # noqa: E402
from mpa_amd.distributed import GradReducer  # noqa: E402
from mpa_amd.optim import FlatAdam  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
DT = sys.argv[3] if len(sys.argv) > 3 else "bf16"
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
crit = get_loss()
names = {m: n for n, m in model.named_modules()}
bad = []


def fwd_hook(m, inp, out):
    outs = out if isinstance(out, (tuple, list)) else (out,)
    for i, t in enumerate(outs):
        if torch.is_tensor(t) and t.is_floating_point() and not torch.isfinite(t.float()).all() and not bad:
            bad.append("forward output %d of %s (%s): %d non-finite of %d, absmax of finite %.3e" % (
                i, names[m], type(m).__name__, int((~torch.isfinite(t.float())).sum()), t.numel(),
                float(t.float()[torch.isfinite(t.float())].abs().max())))


def bwd_hook(m, gin, gout):
    for i, t in enumerate(gout):
        if torch.is_tensor(t) and not torch.isfinite(t.float()).all() and not bad:
            bad.append("grad_output %d of %s (%s)" % (i, names[m], type(m).__name__))
    for i, t in enumerate(gin):
        if torch.is_tensor(t) and not torch.isfinite(t.float()).all() and not bad:
            bad.append("grad_input %d of %s (%s)" % (i, names[m], type(m).__name__))


for m in model.modules():
    m.register_forward_hook(fwd_hook)
    m.register_full_backward_hook(bwd_hook)

red = GradReducer(model, direct=False)
red.overlap = False
opt = None
for it in range(steps):
    red.zero_grad()
    pred, _ = model(x, label)
    loss = crit(pred.reshape(-1, 50), target.reshape(-1))
    loss.backward()
    red.all_reduce()
    if opt is None:
        opt = FlatAdam(red, lr=1e-3)
    gmax = max(float(p.grad.abs().max()) for p in model.parameters() if p.grad is not None)
    for n, p in model.named_parameters():
        if p.grad is not None and not torch.isfinite(p.grad).all() and not bad:
            bad.append("parameter gradient %s" % n)
    opt.step()
    pmax = max(float(p.abs().max()) for p in model.parameters())
    print("step %d loss %.5f  max|grad| %.3e  max|param| %.3e" % (it, float(loss), gmax, pmax), flush=True)
    if bad:
        print("FIRST NON-FINITE:", bad[0], flush=True)
        break
