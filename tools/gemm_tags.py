"""Per-launch shapes and event times of the GEMM entry points in one eager cls-fp32 training pass."""
import argparse, collections, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R]
import torch
import mpa_amd  # noqa
from mpa_amd import ops
from mpa_amd.models.repsurf.repsurf_ssg_umb import Model, SmoothClsLoss
from mpa_amd.runtime import GraphedTrainStep
sys.argv = [sys.argv[0]]
from bench import synthetic_batch

dev = torch.device("cuda")
torch.manual_seed(0)
model = Model(argparse.Namespace(num_point=1024, return_dist=True, cuda_ops=True, num_class=40)).to(dev).train()
data = synthetic_batch("cls", 64, 1024, 1234, dev)
step = GraphedTrainStep(model, SmoothClsLoss(), data, lr=1e-3)
for _ in range(3):
    step(*data)
ops.enable_kernel_timing(["mpa_gemm_f32/tiled", "mpa_gemm_f32/shortk", "mpa_gemm_grouped_f32"])
agg = collections.OrderedDict()
for it in range(5):
    ops._TAGS = []
    step._fwd_bwd()
    torch.cuda.synchronize()
    for name, tag, e0, e1 in ops._TAGS:
        if tag is None:
            continue
        a = agg.setdefault((name,) + tuple(tag), [0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1) * 1e3
ops._TAGS = None
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print("%-22s %6s %6s %6s tA tB st  calls/pass  us/launch  GFLOP  TFLOP/s" % ("entry", "M", "N", "K"))
for (name, M, N, K, tA, tB, st), (n, us) in rows:
    fl = 2.0 * M * N * K
    print("%-22s %6d %6d %6d %2d %2d %2d  %8.1f  %9.1f  %5.2f  %6.1f" % (name, M, N, K, tA, tB, int(st), n / 5, us / n, fl / 1e9, fl / (us / n) / 1e6))
