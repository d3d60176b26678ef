"""Per-op timing of the hot-path kernels at the BASELINE config shapes (cls, B=64, N=1024).
Development tool (run on the GPU box):  python tools/kbench.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import mpa_amd  # noqa: E402
from mpa_amd import ops  # noqa: E402
from param_fill import unit_cloud  # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


def main():
    B = int(os.environ.get("B", 64))
    dev = torch.device("cuda")
    xyz = unit_cloud(B, 1024, seed=1).to(dev)
    start = torch.zeros(B, dtype=torch.long)
    print("== FPS (B=%d)" % B)
    cur = xyz
    for S in (512, 256, 128, 64, 32):
        N = cur.shape[1]
        t = timeit(lambda: ops.farthest_point_sample(cur, S, start_idx=start))
        print("fps N=%4d S=%4d : %8.1f us  (%.3f us/iter)" % (N, S, t, t / S))
        cur = ops.farthest_point_sample(cur, S, start_idx=start, return_xyz=True)[1]
    for N, S in ((2048, 1024), (4096, 2048)):
        x = unit_cloud(8, N, seed=2).to(dev)
        t = timeit(lambda: ops.farthest_point_sample(x, S, start_idx=start[:8]), n=5)
        print("fps N=%4d S=%4d B=8: %8.1f us  (%.3f us/iter)" % (N, S, t, t / S))
    print("== kNN")
    for (S, N, C) in ((1024, 1024, 3), (512, 1024, 3), (256, 512, 3), (512, 1024, 64), (256, 512, 64), (128, 256, 64),
                      (64, 128, 128), (32, 64, 256)):
        base = torch.randn(B, N, C, device=dev)
        q = base[:, :S].contiguous()
        t = timeit(lambda: ops.knn_point(8, base, q))
        pairs = B * S * N
        print("knn S=%4d N=%4d C=%3d : %8.1f us  %.1f Gpair/s  %.2f TFLOP/s" % (S, N, C, t, pairs / t / 1e3,
                                                                          pairs * 2 * C / t / 1e6))
    print("== diffattn (feature branch)")
    for (N, S, C) in ((1024, 512, 64), (512, 256, 64), (256, 128, 128), (128, 64, 256), (64, 32, 512)):
        q = torch.randn(B, S, C, device=dev, requires_grad=True)
        kv = torch.randn(B, N, 2 * C, device=dev, requires_grad=True)
        idx = torch.randint(0, N, (B, S, 8), device=dev)
        t = timeit(lambda: ops.diffattn(q, kv, idx))
        byts = B * S * (4 * (2 * C + 2 * 8 * C) + 8 * 8 + C)
        out = ops.diffattn(q, kv, idx)
        g = torch.randn_like(out)
        tb = timeit(lambda: torch.autograd.grad(out, (q, kv), g, retain_graph=True))
        print("diffattn N=%4d S=%4d C=%3d : fwd %7.1f us (%.2f TB/s algorithmic)  bwd %7.1f us" % (
            N, S, C, t, byts / t / 1e6, tb))
    print("== diffattn xyz (la0)")
    W = [torch.randn(64, 3, device=dev, requires_grad=True) if i % 2 == 0 else torch.randn(64, device=dev, requires_grad=True)
         for i in range(6)]
    idx = torch.randint(0, 1024, (B, 1024, 8), device=dev)
    t = timeit(lambda: ops.diffattn_xyz(xyz, xyz, idx, *W))
    out = ops.diffattn_xyz(xyz, xyz, idx, *W)
    g = torch.randn_like(out)
    tb = timeit(lambda: torch.autograd.grad(out, W, g, retain_graph=True))
    print("diffattn_xyz S=1024 C=64: fwd %7.1f us  bwd %7.1f us" % (t, tb))
    print("== gather")
    f = torch.randn(B, 1024, 64, device=dev)
    fi = torch.randint(0, 1024, (B, 512), device=dev)
    t = timeit(lambda: ops.index_points(f, fi))
    print("index_points [B,1024,64] x [B,512]: %7.1f us" % t)


if __name__ == "__main__":
    main()


def linear_bench():
    from mpa_amd import ops
    dev = torch.device("cuda")
    print("== Linear+BN+LReLU unit (train) fwd / fwd+bwd, and plain GEMM")
    for (M, K, N) in ((65536, 64, 64), (65536, 64, 256), (32768, 64, 64), (32768, 128, 64), (8192, 256, 128),
                      (4096, 512, 256), (2048, 1024, 512), (2048, 512, 1024), (65536, 3, 64)):
        x = torch.randn(M, K, device=dev, requires_grad=True)
        lin = torch.nn.Linear(K, N).to(dev)
        bn = torch.nn.BatchNorm1d(N).to(dev).train()
        t1 = timeit(lambda: ops.linear_bn_act(x, lin.weight, lin.bias, bn, 0.2))
        out = ops.linear_bn_act(x, lin.weight, lin.bias, bn, 0.2)
        g = torch.randn_like(out)
        t2 = timeit(lambda: torch.autograd.grad(out, (x, lin.weight, bn.weight), g, retain_graph=True))
        y = torch.empty(M, N, device=dev)
        t3 = timeit(lambda: ops._gemm(x, K, 0, lin.weight, K, 1, lin.bias, y, N, M, N, K))
        fl = 2.0 * M * K * N
        by = 4.0 * (M * K + M * N)
        print("M=%6d K=%4d N=%4d : unit fwd %7.1f us  bwd %7.1f us | gemm %7.1f us = %6.1f TFLOP/s %5.2f TB/s" % (
            M, K, N, t1, t2, t3, fl / t3 / 1e6, by / t3 / 1e6))


if __name__ == "__main__" and os.environ.get("LINEAR", "1") == "1":
    linear_bench()
