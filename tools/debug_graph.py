"""Compares one HIP-graph replayed fwd+bwd with the same step run eagerly (same weights, same FPS starts)."""
import os, sys, argparse
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpa_amd
from mpa_amd import ops
from mpa_amd.models.repsurf.repsurf_ssg_umb import Model, SmoothClsLoss
from mpa_amd.runtime import FpsStartFeeder
from mpa_amd.distributed import GradReducer
sys.argv = [sys.argv[0]]
from bench import synthetic_batch

B = int(os.environ.get("B", 64))
dev = torch.device("cuda")
torch.manual_seed(0)
args = argparse.Namespace(num_point=1024, return_dist=True, cuda_ops=True, num_class=40)
model = Model(args).to(dev).train()
model.drop1.p = model.drop2.p = 0.0
crit = SmoothClsLoss()
x, y = synthetic_batch(B, 1234, dev)
feeder = FpsStartFeeder()
ops.set_fps_start_hook(feeder)
red = GradReducer(model)
red.overlap = False

def fwd_bwd():
    feeder.begin_pass()
    red.zero_grad()
    loss = crit(model(x), y)
    loss.backward()
    return loss

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        l = fwd_bwd(); red.all_reduce()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("eager loss", float(l))
# freeze BN running-stat drift: irrelevant for train-mode outputs
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    gl = fwd_bwd()
torch.cuda.synchronize()
# same FPS starts for both: do not refill
feeder.frozen = True
g.replay(); torch.cuda.synchronize()
gloss = float(gl)
ggrads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
el = fwd_bwd(); torch.cuda.synchronize()
print("graph loss", gloss, "eager loss", float(el))
bad = 0
for n, p in model.named_parameters():
    if p.grad is None: continue
    a, b = ggrads[n], p.grad
    if not torch.isfinite(a).all() or (a - b).abs().max() > 1e-3 * (b.abs().max() + 1e-6):
        bad += 1
        if bad < 15: print("MISMATCH %-50s graph max %.3e eager max %.3e diff %.3e finite=%s" % (n, a.abs().max(), b.abs().max(), (a - b).abs().max(), bool(torch.isfinite(a).all())))
print("mismatching grads:", bad)
