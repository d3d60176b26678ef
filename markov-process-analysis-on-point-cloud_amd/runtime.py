"""HIP-graph execution of a training step of the Markov path.

The path is hundreds of short kernels per step (992 FPS iterations aside, most launches are
5-50 us), so an eagerly launched step is bounded by host launch overhead, not by the GPU.  All
shapes are static per configuration, so the whole forward + backward is captured once into a
HIP graph (torch.cuda.CUDAGraph = hipGraph on ROCm) and replayed: one launch per step, no
Python between kernels.  The libmpa_hip.so entry points only enqueue work on the caller's
stream (no allocation, no synchronisation), which is what makes them capturable.

One thing in the reference path is host-side state: farthest_point_sample draws its first index
per cloud from the global CPU generator (modules/pointnet2_utils.py:96).  FpsStartFeeder keeps
that behaviour under replay: each FPS call site owns a pinned host buffer + a device buffer, the
captured graph contains the pinned->device copy, and refill() draws fresh indices from the CPU
generator (same order, same distribution) before every replay.
"""
import torch

from . import ops
from .distributed import GradReducer, is_dist, world_size


import os

_SKIP_OPT = bool(os.environ.get("MPA_DEBUG_SKIP_OPT"))      # profiling aid: replay fwd+bwd only


class FpsStartFeeder:
    def __init__(self):
        self.slots = []
        self.cursor = 0
        self.frozen = False      # True: keep the current start indices (debugging / parity runs)

    def __call__(self, B, N, device):
        if self.cursor == len(self.slots):
            self.slots.append({"B": B, "N": N, "host": torch.empty(B, dtype=torch.int64).pin_memory(),
                               "dev": torch.empty(B, dtype=torch.int64, device=device)})
        slot = self.slots[self.cursor]
        if slot["B"] != B or slot["N"] != N:
            raise RuntimeError("FPS call sequence changed shape under a captured step")
        self.cursor += 1
        if not self.frozen and not torch.cuda.is_current_stream_capturing():
            slot["host"].copy_(torch.randint(0, N, (B,), dtype=torch.long))
        slot["dev"].copy_(slot["host"], non_blocking=True)
        return slot["dev"]

    def begin_pass(self):
        self.cursor = 0

    def refill(self):
        for slot in self.slots:
            slot["host"].copy_(torch.randint(0, slot["N"], (slot["B"],), dtype=torch.long))


class GraphedTrainStep:
    """forward + loss + backward captured as one HIP graph, gradient all-reduce (if distributed)
    and the optimizer step (a second graph) after it.

        step = GraphedTrainStep(model, loss_fn, (points, labels), lr=1e-3)
        loss = step(points, labels)        # copies the batch into the static buffers, replays

    Gradients live in the GradReducer's flat buckets, written there directly by the backward
    kernels; the default optimizer is optim.FlatAdam (one launch per bucket).  A torch optimizer
    can be passed instead (`optimizer=`; it must be capturable).  BatchNorm statistics stay per
    rank.  The gradients are reduced after the replayed backward (nothing to overlap with: the
    grouped weight-gradient launch closes the backward), so they travel as ONE flat bucket (cls:
    25.7 MB) -- a single ring all-reduce instead of two."""

    def __init__(self, model, loss_fn, example_batch, optimizer=None, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, warmup=2, bucket_bytes=64 << 20, compute_loss=None):
        # compute_loss(model, loss_fn, *batch) -> scalar loss; default: loss_fn(model(batch[0]), *batch[1:])
        from .optim import FlatAdam
        self.model, self.loss_fn = model, loss_fn
        self.compute_loss = compute_loss
        self.static = [t.clone() for t in example_batch]
        self.feeder = FpsStartFeeder()
        self.reducer = GradReducer(model, bucket_bytes=bucket_bytes, direct=True)
        self.reducer.overlap = False
        ops.set_fps_start_hook(self.feeder)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._fwd_bwd()                          # discovers the live parameters, builds buckets
            self.reducer.all_reduce()
            self.opt = optimizer if optimizer is not None else FlatAdam(self.reducer, lr, betas, eps, weight_decay)
            self.opt.step()
            for _ in range(max(0, warmup - 1)):      # eager passes: warm allocator / workspaces
                self._fwd_bwd()
                self.reducer.all_reduce()
                self.opt.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._fwd_bwd()
        self.opt_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.opt_graph):
            self.opt.step()

    def _fwd_bwd(self):
        self.feeder.begin_pass()
        self.reducer.zero_grad()
        if self.compute_loss is not None:
            loss = self.compute_loss(self.model, self.loss_fn, *self.static)
        else:
            loss = self.loss_fn(self.model(self.static[0]), *self.static[1:])
        ops.defer_weight_grads(True)          # dW products are queued during backward ...
        try:
            loss.backward()
            ops.flush_weight_grads()          # ... and issued as one grouped launch
        finally:
            ops.defer_weight_grads(False)
        return loss

    def __call__(self, *batch):
        for dst, src in zip(self.static, batch):
            if src is not dst:
                dst.copy_(src, non_blocking=True)
        self.feeder.refill()
        self.graph.replay()
        if is_dist() and world_size() > 1:
            self.reducer.all_reduce()
        if not _SKIP_OPT:
            self.opt_graph.replay()
        return self.loss

    def close(self):
        ops.set_fps_start_hook(None)
