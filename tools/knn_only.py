import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpa_amd
from mpa_amd import ops
B, S, N, C = [int(v) for v in os.environ.get("SHAPE", "64,512,1024,64").split(",")]
base = torch.randn(B, N, C, device="cuda")
q = base[:, :S].contiguous()
for _ in range(int(os.environ.get("ITERS", 10))):
    ops.knn_point(8, base, q)
torch.cuda.synchronize()
