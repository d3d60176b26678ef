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
x, y = synthetic_batch("cls", 64, 1024, 1234, dev)
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
if "-p" in os.environ.get("DW_ARGS", ""):
    # every product as a grouped launch of its own: where does the launch's time go?
    import collections
    agg = collections.OrderedDict()
    for q in saved:
        key = (q[5], q[6], q[7])
        if key in agg:
            agg[key][0] += 1
            continue
        for _ in range(2):
            ops._DW_QUEUE.append(q)
            ops.flush_weight_grads()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            ops._DW_QUEUE.append(q)
            ops.flush_weight_grads()
        e1.record()
        torch.cuda.synchronize()
        agg[key] = [1, e0.elapsed_time(e1) / 10 * 1e3]
    tot = 0.0
    for (M, N, K), (cnt, t) in sorted(agg.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        fl, by = 2.0 * M * N * K, 4.0 * K * (M + N)
        tot += cnt * t
        print("  M=%5d N=%5d K=%6d x%d : %7.1f us alone  %6.1f TFLOP/s %5.2f TB/s   (%.0f us if run one by one)" % (
            M, N, K, cnt, t, fl / t / 1e6, by / t / 1e6, cnt * t))
    print("sum of the products run one by one: %.0f us" % tot)
