"""Timing of the difference-wise attention backward at the cls model's shapes (development tool)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import mpa_amd  # noqa: E402
from mpa_amd import ops  # noqa: E402
from mpa_amd._lib import lib  # noqa: E402
from param_fill import unit_cloud  # noqa: E402
from kbench import timeit  # noqa: E402

B = 64
dev = torch.device("cuda")
for (N, S, C) in ((1024, 1024, 64), (1024, 512, 64), (512, 256, 64), (256, 128, 128), (128, 64, 256), (64, 32, 512)):
    xyz = unit_cloud(B, N, seed=N).to(dev)
    idx = ops.knn_point(8, xyz, xyz[:, :S].contiguous())[1]
    q = torch.randn(B, S, C, device=dev)
    kv = torch.randn(B, N, 2 * C, device=dev)
    go = torch.randn(B, S, C, device=dev)
    out = torch.empty_like(q)
    argk = torch.empty(B, S, C, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    lib.mpa_diffattn_fwd_f32(q.data_ptr(), kv.data_ptr(), kv.data_ptr() + 4 * C, 2 * C, idx.data_ptr(), B, N, S, 8, C,
                             out.data_ptr(), argk.data_ptr(), st)
    gq = torch.empty_like(q)
    gkv = torch.empty_like(kv)
    need = int(lib.mpa_diffattn_bwd_workspace_bytes(B, N, S, 8, C))
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)

    def bwd():
        lib.mpa_diffattn_bwd_f32(q.data_ptr(), kv.data_ptr(), kv.data_ptr() + 4 * C, 2 * C, idx.data_ptr(),
                                 argk.data_ptr(), go.data_ptr(), B, N, S, 8, C, gq.data_ptr(), gkv.data_ptr(),
                                 gkv.data_ptr() + 4 * C, 2 * C, ws.data_ptr(), need, st)
    print("N=%4d S=%4d C=%3d : bwd %8.1f us" % (N, S, C, timeit(bwd)), flush=True)
