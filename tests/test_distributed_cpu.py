"""world_size-2 gloo tests of the data-parallel helpers (CPU; the model is a plain torch
stand-in -- the reducer is host logic and does not depend on the HIP kernels)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.body = torch.nn.Sequential(torch.nn.Linear(8, 32), torch.nn.LeakyReLU(0.2), torch.nn.Linear(32, 32),
                                        torch.nn.LeakyReLU(0.2), torch.nn.Linear(32, 4))
        self.unused = torch.nn.Linear(5, 5)   # never receives a gradient (like normal_Trans / fc1)

    def forward(self, x):
        return self.body(x)


def _make_model():
    torch.manual_seed(0)
    return _Net()


def _data():
    g = torch.Generator().manual_seed(1)
    return torch.randn(12, 8, generator=g), torch.randn(12, 4, generator=g)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import mpa_amd  # noqa: F401
    from mpa_amd import distributed as md
    md.init_process_group("gloo")
    assert md.world_size() == world and md.rank() == rank
    model = _make_model()
    red = md.GradReducer(model, bucket_bytes=2048)      # small buckets: several per model
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    x, y = _data()
    lo, hi = md.shard_batch(x.shape[0])
    for _ in range(3):
        red.zero_grad()
        # mean over the GLOBAL batch = mean over ranks of (local sum / local count) when shards are equal
        loss = ((model(x[lo:hi]) - y[lo:hi]) ** 2).mean()
        loss.backward()
        red.all_reduce()
        opt.step()
    assert len(red.buckets) > 1
    assert model.unused.weight.grad is None
    t = md.max_over_ranks(float(rank))
    assert t == world - 1
    if rank == 0:
        torch.save({k: v.clone() for k, v in model.state_dict().items()}, out)
    md.barrier()
    md.shutdown()


def test_grad_reducer_matches_single_process(tmp_path):
    world, port, out = 2, _free_port(), str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    dp = torch.load(out, weights_only=True)
    model = _make_model()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    x, y = _data()
    for _ in range(3):
        opt.zero_grad()
        ((model(x) - y) ** 2).mean().backward()
        opt.step()
    for k, v in model.state_dict().items():
        assert torch.allclose(v, dp[k], atol=1e-6), k


def test_shard_batch_partitions():
    import mpa_amd  # noqa: F401
    from mpa_amd.distributed import shard_batch
    for n in (1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            parts = [shard_batch(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def _worker_split(rank, world, port, out):
    """GradReducer(split_after=...): the buckets that are complete when the split module's backward has run are
    reduced FROM ITS BACKWARD HOOK, overlapping the rest of backward; same result as one process."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import mpa_amd  # noqa: F401
    from mpa_amd import distributed as md
    md.init_process_group("gloo")
    model = _make_model()
    red = md.GradReducer(model, bucket_bytes=1 << 20, split_after=model.body[2])
    red.overlap = False
    fired = []
    red.on_split = lambda: (fired.append(1), red.start(red.early))
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    x, y = _data()
    lo, hi = md.shard_batch(x.shape[0])
    for it in range(3):
        red.zero_grad()
        ((model(x[lo:hi]) - y[lo:hi]) ** 2).mean().backward()
        red.all_reduce()
        opt.step()
    # body[2] and body[4] (weights + biases) are complete when body[2]'s backward has run; body[0] is not
    early = {id(p) for i in red.early for p in red.buckets[i]["params"]}
    assert early == {id(p) for m in (model.body[2], model.body[4]) for p in m.parameters()}
    assert len(red.buckets) == 2 and len(fired) == 2        # the hook launches reductions from the second pass on
    if rank == 0:
        torch.save({k: v.clone() for k, v in model.state_dict().items()}, out)
    md.barrier()
    md.shutdown()


def test_split_reducer_overlaps_and_matches_single_process(tmp_path):
    world, port, out = 2, _free_port(), str(tmp_path / "dp_split.pt")
    mp.spawn(_worker_split, args=(world, port, out), nprocs=world, join=True)
    dp = torch.load(out, weights_only=True)
    model = _make_model()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    x, y = _data()
    for _ in range(3):
        opt.zero_grad()
        ((model(x) - y) ** 2).mean().backward()
        opt.step()
    for k, v in model.state_dict().items():
        assert torch.allclose(v, dp[k], atol=1e-6), k


def _worker_replay(rank, world, port, out):
    """What GraphedTrainStep.__call__ does after a replayed pass with split_after and capture_reduce=False:
    start(early) + all_reduce(), step after step, WITHOUT zero_grad() in between (it runs inside the captured
    graph, not in Python).  Every bucket must be summed over the ranks in every round."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import mpa_amd  # noqa: F401
    from mpa_amd import distributed as md
    md.init_process_group("gloo")
    model = _make_model()
    red = md.GradReducer(model, bucket_bytes=1 << 20, split_after=model.body[2])
    red.overlap = False
    x, y = _data()
    lo, hi = md.shard_batch(x.shape[0])
    red.zero_grad()
    ((model(x[lo:hi]) - y[lo:hi]) ** 2).mean().backward()
    red.all_reduce()                                   # discovery pass: builds the buckets
    assert red.early and len(red.buckets) == 2
    for it in range(4):                                # "replays": the flat buffers are rewritten, no zero_grad()
        for i, b in enumerate(red.buckets):
            b["flat"].fill_(float((rank + 1) * (it + 1) * (i + 1)))
        red.start(red.early)
        red.all_reduce()
        for i, b in enumerate(red.buckets):
            want = sum((r + 1) * (it + 1) * (i + 1) for r in range(world)) / world
            assert torch.allclose(b["flat"], torch.full_like(b["flat"], want)), (it, i, float(b["flat"][0]), want)
    if rank == 0:
        open(out, "w").write("ok")
    md.barrier()
    md.shutdown()


def test_split_reducer_rearms_without_zero_grad(tmp_path):
    world, port, out = 2, _free_port(), str(tmp_path / "replay.txt")
    mp.spawn(_worker_replay, args=(world, port, out), nprocs=world, join=True)
    assert open(out).read() == "ok"


class _PairNet(torch.nn.Module):
    """two projections a module wants back to back in the flat buffers (as LocalTrans' k | v)"""

    def __init__(self):
        super().__init__()
        self.k = torch.nn.Linear(8, 6)
        self.mid = torch.nn.Linear(8, 8)
        self.v = torch.nn.Linear(8, 6)

    def mpa_adjacent_params(self):
        return ((self.k.weight, self.v.weight), (self.k.bias, self.v.bias))

    def forward(self, x):
        h = self.mid(x)
        return (self.k(h) * self.v(h)).sum(-1)


def _worker_direct(rank, world, port, out):
    """GradReducer(direct=True) on a module with mpa_adjacent_params: the grouped parameters' flat views are
    adjacent, `_mpa_grad_buf` is installed (what the libmpa backward kernels write through), writes through it
    land in .grad, and the reduced result equals one process."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import mpa_amd  # noqa: F401
    from mpa_amd import distributed as md
    md.init_process_group("gloo")
    torch.manual_seed(0)
    model = _PairNet()
    red = md.GradReducer(model, bucket_bytes=1 << 20, direct=True)
    red.overlap = False
    g = torch.Generator().manual_seed(2)
    x = torch.randn(10, 8, generator=g)
    lo, hi = md.shard_batch(10)
    red.zero_grad()
    model(x[lo:hi]).sum().backward()
    red.all_reduce()
    kw, vw = model.k.weight.grad, model.v.weight.grad
    assert vw.data_ptr() == kw.data_ptr() + 4 * ((kw.numel() + 3) // 4 * 4)          # back to back (16-byte aligned)
    assert all(getattr(p, "_mpa_grad_buf", None) is p.grad for p in model.parameters())
    red.zero_grad()
    for p in model.parameters():                       # a "kernel" writing straight into the flat gradients
        p._mpa_grad_buf.fill_(float(rank + 1))
    red.all_reduce()
    for p in model.parameters():
        assert torch.allclose(p.grad, torch.full_like(p.grad, (1 + world) / 2))
    if rank == 0:
        open(out, "w").write("ok")
    md.barrier()
    md.shutdown()


def test_direct_reducer_with_adjacent_params(tmp_path):
    world, port, out = 2, _free_port(), str(tmp_path / "direct.txt")
    mp.spawn(_worker_direct, args=(world, port, out), nprocs=world, join=True)
    assert open(out).read() == "ok"
