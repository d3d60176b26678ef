"""Host-side cost of a GraphedTrainStep replay against its GPU time (development tool): is the CPU
ahead of the GPU?  (It is: 0.17 ms of host time per 4.2 ms step.)"""
import os, sys, time, argparse, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpa_amd
from mpa_amd.models.repsurf.repsurf_ssg_umb import Model, SmoothClsLoss
from mpa_amd.runtime import GraphedTrainStep
sys.argv = [sys.argv[0]]
from bench import synthetic_batch
dev = torch.device("cuda")
torch.manual_seed(0)
args = argparse.Namespace(num_point=1024, return_dist=True, cuda_ops=True, num_class=40)
model = Model(args).to(dev).train()
x, y = synthetic_batch(64, 1234, dev)
step = GraphedTrainStep(model, SmoothClsLoss(), (x, y), lr=1e-3)
for _ in range(5): step(x, y)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n): step(x, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host loop %.3f ms/step, with final sync %.3f ms/step" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
