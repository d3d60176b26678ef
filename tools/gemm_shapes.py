"""Per-shape GEMM time in one eager train step (development tool)."""
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
ops.enable_kernel_timing(["mpa_gemm_f32"]); ops._TAGS = []
for _ in range(5): step()
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for name, tag, e0, e1 in ops._TAGS:
    agg[tag][0] += 1; agg[tag][1] += e0.elapsed_time(e1) * 1e3
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows) / 5
print("total gemm us/step %.0f" % tot)
for tag, (n, us) in rows[:40]:
    M, N, K, tA, tB, st = tag
    fl = 2.0 * M * N * K
    print("M=%6d N=%5d K=%6d tA=%d tB=%d stats=%d : %3d calls/step %7.1f us each %7.1f us/step %6.1f TF %5.2f TB/s" % (
        M, N, K, tA, tB, st, n // 5, us / n, us / 5, fl / (us / n) / 1e6, 4.0 * (M * K + N * K + M * N) / (us / n) / 1e6))
