"""Runs the same forward twice (same weights, same FPS starts) and reports the first block whose
output is not reproducible."""
import os, sys, argparse
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpa_amd
from mpa_amd import ops
from mpa_amd.models.repsurf.repsurf_ssg_umb import Model, SmoothClsLoss
from mpa_amd.runtime import FpsStartFeeder
sys.argv = [sys.argv[0]]
from bench import synthetic_batch

B = int(os.environ.get("B", 64))
dev = torch.device("cuda")
torch.manual_seed(0)
args = argparse.Namespace(num_point=1024, return_dist=True, cuda_ops=True, num_class=40)
model = Model(args).to(dev).train()
model.drop1.p = model.drop2.p = 0.0
x, y = synthetic_batch(B, 1234, dev)
feeder = FpsStartFeeder()
ops.set_fps_start_hook(feeder)
outs = []
def hook(name):
    def f(mod, inp, out):
        o = out[0] if isinstance(out, tuple) else out
        outs[-1][name] = o.detach().clone()
    return f
for n, m in model.named_modules():
    if n and n.count(".") <= 3:
        m.register_forward_hook(hook(n))
for it in range(3):
    outs.append({})
    feeder.begin_pass()
    loss = SmoothClsLoss()(model(x), y)
    feeder.frozen = True
    print("pass", it, "loss", float(loss))
for it in (1, 2):
    bad = [(n, float((outs[it][n] - outs[0][n]).abs().max())) for n in outs[0] if outs[it][n].shape == outs[0][n].shape and not torch.equal(outs[it][n], outs[0][n])]
    print("pass", it, "differing modules:", len(bad), "of", len(outs[0]))
    for n, d in bad[:12]:
        print("   %-50s max diff %.3e" % (n, d))
