"""Is the forward pass of the part-seg wiring at 4096 points reproducible run to run (deterministic statistics)?"""
import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
import mpa_amd
from mpa_amd import ops
from param_fill import unit_cloud, fill_state
from mpa_amd.models.repsurf.pointnet2_part_seg_msg import get_model, get_loss
B, N, NC = 2, 4096, 13
ops._fps_start = lambda B_, N_, device, start_idx=None: ((torch.arange(B_, device=device) * 5 + 1) % N_ if start_idx is None else start_idx.to(device))
ops.set_deterministic(True)
label = torch.zeros(B, 1, 16); label[:, 0, 2] = 1
x = unit_cloud(B, N, seed=22).transpose(1, 2).contiguous().cuda()
model = fill_state(get_model(NC), seed=2).cuda().train()
model.drop1.p = 0.0
outs = {}
names = {}
def hook(name):
    def f(mod, inp, out):
        o = out[0] if isinstance(out, tuple) else out
        if isinstance(o, torch.Tensor):
            outs.setdefault(name, []).append(o.detach().float().clone())
    return f
for n, m in model.named_modules():
    if n:
        m.register_forward_hook(hook(n))
for it in range(6):
    with torch.no_grad():
        model(x, label.cuda())
torch.cuda.synchronize()
bad = 0
for n, v in outs.items():
    d = max(float((v[0] - u).abs().max()) for u in v[1:])
    d1 = max(float((v[1] - u).abs().max()) for u in v[2:])
    if d > 0:
        bad += 1
        if bad < 25:
            print("%-50s run 0 vs later %.3e, runs 1..5 among themselves %.3e  (shape %s)" % (n, d, d1, tuple(v[0].shape)))
print("modules with run-to-run differences:", bad, "of", len(outs))
