"""Replays the cls model's grouped weight-gradient launch (all dW = dY^T X products of one
backward) and times it (development tool; knobs: MPA_TN_STREAM, MPA_TN_WGS, MPA_TN_KCHUNK)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpa_amd  # noqa: E402
from mpa_amd import ops  # noqa: E402
from mpa_amd.models.repsurf.repsurf_ssg_umb import Model, SmoothClsLoss  # noqa: E402
from mpa_amd.distributed import GradReducer  # noqa: E402
sys.argv = [sys.argv[0]]
from bench import synthetic_batch  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
args = argparse.Namespace(num_point=1024, return_dist=True, cuda_ops=True, num_class=40)
model = Model(args).to(dev).train()
crit = SmoothClsLoss()
x, y = synthetic_batch(64, 1234, dev)
red = GradReducer(model, direct=True)
red.overlap = False


def step():
    red.zero_grad()
    loss = crit(model(x), y)
    loss.backward()
    red.all_reduce()


for _ in range(2):
    step()
ops.defer_weight_grads(True)
red.zero_grad()
crit(model(x), y).backward()
saved = list(ops._DW_QUEUE)
print("%d products" % len(saved))
tot_b = tot_f = 0
for (gy, lda, xx, ldb, out, M, N, K, acs) in saved:
    tot_b += 4 * K * (M + N)
    tot_f += 2 * M * N * K
    if "-v" in os.environ.get("DW_ARGS", ""):
        print("  M=%5d N=%5d K=%6d" % (M, N, K))
print("unique operand bytes %.1f MB, %.2f GFLOP" % (tot_b / 1e6, tot_f / 1e9))
ops.flush_weight_grads()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 10
for _ in range(3):
    ops._DW_QUEUE.extend(saved)
    ops.flush_weight_grads()
torch.cuda.synchronize()
e0.record()
for _ in range(n):
    ops._DW_QUEUE.extend(saved)
    ops.flush_weight_grads()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
print("grouped dW: %.1f us  (%.2f TB/s unique, %.1f TFLOP/s)  stream=%s wgs=%s kchunk=%s" % (
    us, tot_b / us / 1e6, tot_f / us / 1e6, os.environ.get("MPA_TN_STREAM", "1"), os.environ.get("MPA_TN_WGS", "512"),
    os.environ.get("MPA_TN_KCHUNK", "256")))
