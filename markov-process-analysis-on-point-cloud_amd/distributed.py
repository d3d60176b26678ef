"""One-process-per-GPU data parallelism for the Markov set-abstraction path.

The path shards by batch: FPS, kNN, grouping, attention and upsample are all per-cloud, so each
rank owns B/world clouds and there is NO data-path collective.  The only exchange is one
all-reduce (mean) of the gradients per step, over RCCL (`backend="nccl"` on ROCm) on xGMI.
The reference has no distributed code at all (single GPU selected by a hard-coded env var,
tool/train_cls_scanobjectnn.py:135); this is the "scripts/ train loop DDP init" named by
BASELINE.json's north_star.

Design for xGMI (point-to-point links, per-link-bound rings): gradients live in a few large flat
buckets (default 16 MiB; cls has 25.7 MB of live gradients -> 2 buckets), parameters' .grad are
views into them, and each bucket's all-reduce is launched from a post-accumulate hook as soon as
its last gradient lands, overlapping the rest of backward.  Parameters that never receive a
gradient (2.1 M in cls: normal_Trans, fc1, start, final, every norm1) are left out of the
buckets but stay in the state dict.

BatchNorm statistics stay per-rank (as plain DDP does); see DESIGN.md for the caveat.
"""
import os

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def init_process_group(backend=None):
    """Reads RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment (torchrun)."""
    if is_dist():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    kw = {}
    if backend == "nccl":
        kw["device_id"] = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend, **kw)


def shutdown():
    if is_dist():
        dist.destroy_process_group()


def world_size():
    return dist.get_world_size() if is_dist() else 1


def rank():
    return dist.get_rank() if is_dist() else 0


def barrier():
    if is_dist():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    if not is_dist():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_batch(n_items, r=None, w=None):
    """Contiguous [start, stop) slice of a global batch owned by rank r of w."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    base, rem = divmod(n_items, w)
    start = r * base + min(r, rem)
    return start, start + base + (1 if r < rem else 0)


class GradReducer:
    """Bucketed, backward-overlapped gradient all-reduce (mean) over the default process group.

    Usage per step:   reducer.zero_grad(); loss.backward(); reducer.all_reduce(); opt.step()
    The first backward discovers which parameters receive gradients; from then on their .grad
    tensors are views into flat buckets and reductions start from autograd hooks."""

    def __init__(self, module, bucket_bytes=16 << 20, direct=False, split_after=None):
        # direct=True: the libmpa backward kernels write parameter gradients straight into the flat
        # buckets (ops._direct) instead of handing tensors to autograd's AccumulateGrad -- valid when
        # every parameter is used once per step (true for the cls / part-seg models) and
        # zero_grad() is called every step.
        self.direct = direct
        self.module = module
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.bucket_bytes = bucket_bytes
        self.buckets = None          # list of dicts: flat, params, pending, handle
        self.overlap = True          # False: no hook-launched collectives (HIP-graph replayed backward)
        self._where = {}
        self._handles = []
        # split_after: a sub-module whose backward marks the point where the gradients of everything that ran
        # AFTER it in the forward pass (itself included) are complete -- for the cls model `keepHigh.la4`: head,
        # la5 and la4 hold 96 % of the gradient bytes and are done after the first quarter of the backward pass.
        # Those parameters get their own bucket(s) (self.early), reduced while the rest of backward runs
        # (on_split(), called from the module's backward hook).
        self.split_after = split_after
        self.early = []              # indices of the buckets that are complete at the split point
        self.on_split = None         # callable run by the split module's backward hook (after the first build)
        self._order, self._split_pos = [], None
        self._order_hooks = []
        if split_after is not None:
            self._order_hooks = [p.register_post_accumulate_grad_hook(self._record_order) for p in self.params]
            split_after.register_full_backward_hook(self._split_hook)

    def _record_order(self, p):
        if self.buckets is None:
            self._order.append(p)

    def _split_hook(self, module, grad_input, grad_output):
        if self.buckets is None:
            if self._split_pos is None:
                self._split_pos = len(self._order)      # discovery pass: everything accumulated so far is "early"
        elif self.on_split is not None:
            self.on_split()

    def start(self, which):
        """Launch the all-reduce (sum) of the given buckets now (asynchronous; all_reduce() completes them)."""
        if not is_dist():
            return
        for i in which:
            b = self.buckets[i]
            if not b.get("started"):
                b["started"] = True
                self._handles.append(dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, async_op=True))

    # -- bucket construction (after the first backward) --------------------------------------
    def _build(self):
        for h in self._order_hooks:  # (discovery only: hooks keep AccumulateGrad nodes alive, see below)
            h.remove()
        self._order_hooks = []
        live = [p for p in self.params if p.grad is not None]
        live.reverse()               # roughly the order gradients become ready
        early_set = None
        if self.split_after is not None and self._split_pos is not None:
            early_set = set(self._order[:self._split_pos])
        # parameter groups that a module wants back to back in the flat buffers (LocalTrans' k|v
        # projections are read as one stacked weight): a group is placed whole, in its own order
        group_of = {}
        for m in self.module.modules():
            for grp in getattr(m, "mpa_adjacent_params", lambda: ())():
                grp = [p for p in grp if p.grad is not None]
                if len(grp) > 1:
                    for p in grp:               # a parameter follows the largest group that names it
                        if p not in group_of or len(group_of[p]) < len(grp):
                            group_of[p] = grp
        units, seen = [], set()
        for p in live:
            if p in seen:
                continue
            unit = group_of.get(p, [p])
            units.append(unit)
            seen.update(unit)
        self.buckets = []
        # a unit (parameters kept adjacent) is early only if all of it is
        groups = [units] if early_set is None else [[u for u in units if all(p in early_set for p in u)],
                                                    [u for u in units if not all(p in early_set for p in u)]]
        for gi, group in enumerate(groups):
            cur, cur_bytes = [], 0
            first = len(self.buckets)
            for unit in group:
                nbytes = sum(p.numel() * p.element_size() for p in unit)
                if cur and cur_bytes + nbytes > self.bucket_bytes:
                    self._make_bucket(cur)
                    cur, cur_bytes = [], 0
                cur.extend(unit)
                cur_bytes += nbytes
            if cur:
                self._make_bucket(cur)
            if early_set is not None and gi == 0:
                self.early = list(range(first, len(self.buckets)))
        # Hooks keep the AccumulateGrad nodes (and the stream they were created on) alive across
        # iterations; a HIP-graph captured backward must not inherit them (the accumulation would
        # fork onto the stale stream inside the capture), so they are only installed for overlap.
        if self.overlap:
            for p in live:
                p.register_post_accumulate_grad_hook(self._hook)

    def _make_bucket(self, params):
        # every view starts on a 16-byte boundary (a 50-float bias would otherwise push all later
        # parameters off the float4 paths of the kernels); the pad elements stay zero
        offs, off = [], 0
        for p in params:
            offs.append(off)
            off += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(off, dtype=params[0].dtype, device=params[0].device)
        b = {"flat": flat, "params": params, "pending": len(params), "views": [], "offsets": offs}
        for p, o in zip(params, offs):
            v = flat[o:o + p.numel()].view_as(p)
            v.copy_(p.grad)
            p.grad = v
            b["views"].append(v)
            if self.direct:
                p._mpa_grad_buf = v
            self._where[p] = b
        self.buckets.append(b)

    def _hook(self, p):
        b = self._where.get(p)
        if b is None or not is_dist() or not self.overlap:
            return
        b["pending"] -= 1
        if b["pending"] == 0:
            self._handles.append(dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, async_op=True))

    # -- per-step API ------------------------------------------------------------------------
    def zero_grad(self):
        if self.buckets is None:
            for p in self.params:
                p.grad = None
            return
        for b in self.buckets:
            b["flat"].zero_()
            b["pending"] = len(b["params"])
            b["started"] = False
            for p, v in zip(b["params"], b["views"]):
                p.grad = v

    def all_reduce(self):
        """Completes the step's reduction: gradients become the mean over ranks."""
        w = world_size()
        if self.buckets is None:
            self._build()
            if is_dist():
                for b in self.buckets:
                    dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM)
        else:
            for b in self.buckets:      # a bucket whose hook did not fire (or overlap is off)
                if (b["pending"] != 0 or not self.overlap) and is_dist() and not b.get("started"):
                    self._handles.append(dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, async_op=True))
            for h in self._handles:
                h.wait()
            self._handles = []
            # the step's reductions are complete: re-arm every bucket.  zero_grad() also does this, but under HIP-graph
            # replay zero_grad() runs inside the captured pass only -- Python never executes it again -- so a bucket
            # left "started" would be skipped by start() AND by this method from the second replayed step on (its
            # gradients divided by the world size but never summed).
            for b in self.buckets:
                b["started"] = False
        if w > 1:
            for b in self.buckets:
                b["flat"].div_(w)
