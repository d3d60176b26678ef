"""Achievable chip peaks (HBM copy, fp32 MFMA GEMM through mpa_gemm_f32) and forward-only
throughput of the cls model ('measure achievable HBM BW with a copy kernel and MFMA
peak with a GEMM on the box'; the >= 20x forward target).  Development / DESIGN.md numbers."""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import mpa_amd
from mpa_amd import ops
from mpa_amd.models.repsurf.repsurf_ssg_umb import Model
from kbench import timeit
sys.argv = [sys.argv[0]]
from bench import synthetic_batch

dev = torch.device("cuda")
# HBM copy: 1 GiB device-to-device (read + write bytes counted)
a = torch.empty(1 << 28, dtype=torch.float32, device=dev)
b = torch.empty_like(a)
us = timeit(lambda: b.copy_(a), n=10)
print("HBM copy 1 GiB: %.1f us -> %.2f TB/s (read+write)" % (us, 2 * a.numel() * 4 / us / 1e6))
del a, b
# fp32 MFMA GEMM peak through the library: 8192 x 8192 x 4096 (NT)
M, N, K = 8192, 8192, 4096
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.empty(M, N, device=dev)
us = timeit(lambda: ops._gemm(A, K, 0, B, K, 1, None, C, N, M, N, K), n=5)
print("mpa_gemm_f32 %dx%dx%d: %.1f us -> %.1f TFLOP/s" % (M, N, K, us, 2.0 * M * N * K / us / 1e6))
us = timeit(lambda: torch.matmul(A, B.t()), n=5)
print("torch.matmul (hipBLASLt) same shape: %.1f us -> %.1f TFLOP/s" % (us, 2.0 * M * N * K / us / 1e6))
del A, B, C
# forward-only throughput, eval-free (train-mode BN statistics), HIP graph
torch.manual_seed(0)
args = argparse.Namespace(num_point=1024, return_dist=True, cuda_ops=True, num_class=40)
model = Model(args).to(dev).train()
x, y = synthetic_batch(64, 1234, dev)
start = torch.zeros(64, dtype=torch.long)
from mpa_amd.runtime import FpsStartFeeder
feeder = FpsStartFeeder(); ops.set_fps_start_hook(feeder)
with torch.no_grad():
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            feeder.begin_pass(); model(x)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        feeder.begin_pass(); out = model(x)
    def fwd():
        feeder.refill(); g.replay()
    for _ in range(5): fwd()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fwd()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print("cls forward only (B=64, N=1024, fp32, HIP graph): %.3f ms -> %.0f clouds/s" % (dt * 1e3, 64 / dt))
ops.set_fps_start_hook(None)
# (the CPU forward baseline is measured by bench.py's cpu_baseline leg: "forward_only_value")
