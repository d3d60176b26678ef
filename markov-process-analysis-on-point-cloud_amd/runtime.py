"""HIP-graph execution of a training step of the Markov path.

The path is hundreds of short kernels per step (992 FPS iterations aside, most launches are
5-50 us), so an eagerly launched step is bounded by host launch overhead, not by the GPU.  All
shapes are static per configuration, so the whole forward + backward is captured once into a
HIP graph (torch.cuda.CUDAGraph = hipGraph on ROCm) and replayed: one launch per step, no
Python between kernels.  The libmpa_hip.so entry points only enqueue work on the caller's
stream (no allocation, no synchronisation), which is what makes them capturable.

One thing in the reference path is host-side state: farthest_point_sample draws its first index
per cloud from the global CPU generator (modules/pointnet2_utils.py:96).  FpsStartFeeder keeps
that behaviour under replay: each FPS call site owns a slice of one device buffer that the captured
kernels read in place, and refill() draws fresh indices from the CPU generator (same calls, same
order) and ships them with one stream-ordered copy before every replay.
"""
import torch

from . import ops
from .distributed import GradReducer, is_dist, world_size


import os

_SKIP_OPT = bool(os.environ.get("MPA_DEBUG_SKIP_OPT"))      # profiling aid: replay fwd+bwd only


class FpsStartFeeder:
    """Graph-safe source of farthest_point_sample's first indices (ops.set_fps_start_hook).

    Every FPS call site of a pass owns a slice of ONE int64 device buffer; a captured graph's FPS kernels
    read their slices in place (no copy node in the graph).  refill() draws all slices from the global CPU
    generator -- same calls, same order as the reference's per-call torch.randint -- into a FRESH pinned
    tensor and issues one stream-ordered H2D copy; torch's pinned-memory allocator does not hand that
    block out again before the copy has run, so a host that runs many replays ahead never overwrites
    draws an earlier replay has yet to read (each replay sees exactly its own draws).

    The hook only answers between begin_pass() and end_pass(): any other forward (an evaluation between
    training steps, vote_classification, ...) falls through to the plain reference path and cannot grow
    the slot list."""

    CAPACITY = 1 << 16           # start indices per pass (all FPS call sites x clouds)

    def __init__(self):
        self.prefilled = False
        self.slots = []          # (B, N, offset) per FPS call site, in call order
        self.total = 0
        self.dev = None          # int64 [total] device buffer the kernels read
        self.cursor = 0
        self.active = False
        self.sealed = False      # True once a graph has been captured over the slots: the layout is fixed
        self.frozen = False      # True: keep the current start indices (debugging / parity runs)

    def __call__(self, B, N, device):
        if not self.active:
            return None          # not inside a fed pass: ops falls back to the reference's CPU draw
        capturing = torch.cuda.is_current_stream_capturing()
        if self.cursor == len(self.slots):
            if self.sealed or capturing:
                raise RuntimeError("FPS call sequence grew after the step was captured")
            if self.dev is None:
                self.dev = torch.zeros(self.CAPACITY, dtype=torch.int64, device=device)
            if self.total + B > self.CAPACITY:
                raise RuntimeError("FpsStartFeeder: more than %d start indices per pass" % self.CAPACITY)
            self.slots.append((B, N, self.total))
            self.total += B
        sB, sN, off = self.slots[self.cursor]
        if sB != B or sN != N:
            raise RuntimeError("FPS call sequence changed shape under a captured step")
        self.cursor += 1
        view = self.dev[off:off + B]
        if not self.frozen and not capturing and not self.prefilled:
            src = torch.randint(0, N, (B,), dtype=torch.long).pin_memory()
            view.copy_(src, non_blocking=True)
        return view

    def begin_pass(self, prefilled=False):
        """prefilled: the caller has just refill()ed every slot (a replayed or re-run pass)."""
        self.cursor = 0
        self.active = True
        self.prefilled = prefilled

    def end_pass(self):
        self.active = False

    def seal(self):
        self.sealed = True

    def refill(self):
        """Fresh draws for every slot: one pinned tensor, one stream-ordered copy."""
        if self.frozen or not self.slots:
            return
        src = torch.cat([torch.randint(0, N, (B,), dtype=torch.long) for B, N, _ in self.slots]).pin_memory()
        self.dev[:self.total].copy_(src, non_blocking=True)


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
                 weight_decay=0.0, warmup=2, bucket_bytes=64 << 20, compute_loss=None, split_after=None,
                 capture_reduce=False, prefetch_geometry=False, coords_of=None):
        # compute_loss(model, loss_fn, *batch) -> scalar loss; default: loss_fn(model(batch[0]), *batch[1:])
        from .optim import FlatAdam
        self.model, self.loss_fn = model, loss_fn
        self.compute_loss = compute_loss
        self.static = [t.clone() for t in example_batch]
        self.feeder = FpsStartFeeder()
        self.arena = ops.ZeroArena(self.static[0].device)       # the pass's pre-zeroed accumulators (one fill per step)
        # split_after (a sub-module, e.g. model.keepHigh.la4): when its backward has run, the weight gradients queued so
        # far (everything downstream of it in the forward pass: most of the gradient bytes) are flushed as an early
        # grouped launch and sit complete in their own flat bucket(s).  capture_reduce=True additionally starts
        # that bucket's all-reduce right there, INSIDE the captured graph (RCCL collectives are capturable), so it
        # overlaps the rest of backward instead of running after the replay.
        self.reducer = GradReducer(model, bucket_bytes=bucket_bytes, direct=True, split_after=split_after)
        self.reducer.overlap = False
        # (MPA_CAPTURE_REDUCE_SINGLE_RANK=1: also on a one-rank group -- the captured RCCL collective then runs on a
        # single GPU, which is how tests/test_gpu_rccl_capture.py exercises this path on a one-GPU box)
        self.capture_reduce = bool(capture_reduce) and is_dist() and (
            world_size() > 1 or os.environ.get("MPA_CAPTURE_REDUCE_SINGLE_RANK") == "1")
        if split_after is not None:
            self.reducer.on_split = self._on_split
        ops.set_fps_start_hook(self.feeder)
        # prefetch_geometry: the sampling chain + the first two coordinate searches of the NEXT batch are computed by
        # geometry riders inside this step's closing weight-gradient launches (ops.GeometryPrefetch) instead of at the
        # head of the next step's forward pass.  step(*batch, next_batch=...) announces the next batch; a batch that was
        # not announced gets its geometry computed on the spot (correct, not hidden).  coords_of(batch) -> [B,3,N]
        # channel-first coordinates of a batch (default: the first three channels of batch[0], as both models read).
        # prefetch_geometry = True / "searches": the next batch's chain rides in THIS batch's search launches
        # (ops.GeometryPipeline); "riders": in the closing weight-gradient launches (ops.GeometryPrefetch, slower)
        self.prefetch = None
        if prefetch_geometry:
            self.prefetch = ops.GeometryPrefetch() if prefetch_geometry == "riders" else ops.GeometryPipeline()
        self.coords_of = coords_of or (lambda batch: batch[0][:, :3])
        self._announced = None                   # (tensor, version) of the batch whose geometry the buffers hold
        # (the prefetch object is installed only while one of THIS step's passes runs, _fwd_bwd: any other forward of
        # the model -- an evaluation between steps -- must run its own chain, not read another batch's geometry)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._fwd_bwd()                          # discovers the live parameters, builds buckets
            if self.prefetch is not None:
                pf = self.prefetch
                if pf.spec is None or not pf.supported():
                    self.prefetch = None                                     # no chain in this model / unsupported shape
                else:
                    pf.allocate(self.static[0].device)
                    pf.next_xyz.copy_(self.coords_of(self.static).transpose(1, 2))
                    pf.compute_now(pf.next_xyz)      # geometry of the batch the next (eager) pass sees
            self.reducer.all_reduce()
            self.opt = optimizer if optimizer is not None else FlatAdam(self.reducer, lr, betas, eps, weight_decay)
            self.opt.step()
            for _ in range(max(0, warmup - 1)):      # eager passes: warm allocator / workspaces
                self._fwd_bwd()
                if not self.capture_reduce:          # (with capture_reduce the pass itself ends with the reduction)
                    self.reducer.all_reduce()
                self.opt.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

        self.feeder.refill()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._fwd_bwd()
        self.feeder.seal()
        self.opt_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.opt_graph):
            self.opt.step()

    def _on_split(self):
        ops.flush_weight_grads()                 # early grouped launch: the gradients of the early buckets are now complete
        if self.capture_reduce:
            self.reducer.start(self.reducer.early)

    def _fwd_bwd(self):
        ops.clear_knn_memo()
        self.feeder.begin_pass()
        self.reducer.zero_grad()
        self.arena.begin()
        ops.set_arena(self.arena)
        saved_prefetch = ops.set_geometry_prefetch(self.prefetch)
        try:
            if self.compute_loss is not None:
                loss = self.compute_loss(self.model, self.loss_fn, *self.static)
            else:
                loss = self.loss_fn(self.model(self.static[0]), *self.static[1:])
        finally:
            self.feeder.end_pass()
            ops.set_arena(None)
            self.arena.end()
            ops.set_geometry_prefetch(saved_prefetch)
        ops.defer_weight_grads(True)          # dW products are queued during backward ...
        try:
            loss.backward()
            # ... and issued as one grouped launch, which also carries the next batch's geometry (prefetch mode)
            ops.flush_weight_grads(riders=self.prefetch.riders() if self.prefetch is not None else None)
        finally:
            ops.defer_weight_grads(False)
        if self.capture_reduce and self.reducer.buckets is not None:
            self.reducer.all_reduce()         # the remaining buckets + completion of the early ones, inside the graph
        return loss

    def __call__(self, *batch, next_batch=None):
        for dst, src in zip(self.static, batch):
            if src is not dst:
                dst.copy_(src, non_blocking=True)
        self.feeder.refill()
        if self.prefetch is not None:
            pf = self.prefetch
            if self._announced is None or self._announced[0] is not batch[0] or self._announced[1] != batch[0]._version:
                # this batch was not announced a step ahead: its geometry now, as launches of their own
                pf.next_xyz.copy_(self.coords_of(batch).transpose(1, 2))
                pf.compute_now(pf.next_xyz)
            nxt = next_batch if next_batch is not None else batch      # (no announcement: the same batch again)
            pf.next_xyz.copy_(self.coords_of(nxt).transpose(1, 2))
            self._announced = (nxt[0], nxt[0]._version)
        self.graph.replay()
        if is_dist() and world_size() > 1 and not self.capture_reduce:
            if self.reducer.early:                                   # the early bucket(s) first: they are the bulk
                self.reducer.start(self.reducer.early)
            self.reducer.all_reduce()
        if not _SKIP_OPT:
            sync = getattr(self.opt, "sync_hyper", None)
            if sync is not None:
                sync()                       # a scheduler may have moved the learning rate since the last step
            self.opt_graph.replay()
        return self.loss

    def close(self):
        ops.set_fps_start_hook(None)
