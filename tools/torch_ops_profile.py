"""Which ATen ops (not libmpa kernels) still launch kernels in one eager train step, and from where."""
import os, sys, argparse, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpa_amd
from mpa_amd import ops
from mpa_amd.models.repsurf.repsurf_ssg_umb import Model, SmoothClsLoss
from mpa_amd.distributed import GradReducer
sys.argv = [sys.argv[0]]
from bench import synthetic_batch
dev = torch.device("cuda")
torch.manual_seed(0)
args = argparse.Namespace(num_point=1024, return_dist=True, cuda_ops=True, num_class=40)
model = Model(args).to(dev).train()
crit = SmoothClsLoss()
x, y = synthetic_batch(64, 1234, dev)
red = GradReducer(model, direct=True); red.overlap = False
def step():
    red.zero_grad(); loss = crit(model(x), y); loss.backward(); red.all_reduce()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
    dt = getattr(e, "device_time_total", None) or getattr(e, "cuda_time_total", 0)
    if dt > 0 and e.key.startswith("aten::"):
        where = [f for f in e.stack if "mpa" in f or "markov" in f]
        rows.append((dt, e.count, e.key, str(e.input_shapes)[:70], " <- ".join(w.split("/")[-1][:60] for w in where[:3])))
rows.sort(reverse=True)
for dt, cnt, key, shp, where in rows[:60]:
    print("%8.1f us %4d  %-18s %-70s %s" % (dt, cnt, key, shp, where))
