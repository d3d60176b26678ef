"""bench.py -- point-clouds/sec, forward+backward, ModelNet40-shaped 1024-point classification
(BASELINE.json metric) on N MI355X GPUs of one node.

A "step" is one training pass of the hot path over one batch of synthetic clouds:
forward (FPS -> kNN grouping -> difference-wise attention -> transition MLPs -> head), the
label-smoothed loss, backward, gradient all-reduce over RCCL when N > 1, and an Adam step.
Workload at N=1: BASELINE configs[1] -- 1024 points, batch 64 per GPU, fp32 (weak scaling:
per-GPU batch fixed).  Inputs are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Extra objects: "roofline" for the dominant kernel (HIP events on
the launch stream around that kernel's launches inside the timed region), "cpu_baseline" (the
CPU oracle oracle/ref_cpu.py timed on this box's host cores on a bounded sample; rank 0, N=1;
its "forward_only_value" is the same model's forward pass alone) and, at N=1, "forward_only" (the
forward pass alone as a HIP graph, measured after the timed region: a secondary figure).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NUM_POINT, NUM_CLASS = 1024, 40
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3     # dense fp32 MFMA peak (MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32)
# kernels timed with HIP events for the roofline leg: name -> bound
# (mpa_gemm_f32 launches are priced by the device kernel they pick: the 64x64-tile kernel against the
#  MFMA peak, the short-K kernel of the K <= 128 layers -- 16 FLOP/B -- against HBM)
TIMED = {"mpa_gemm_f32/tiled": "mfma", "mpa_gemm_f32/shortk": "hbm", "mpa_gemm_tn_grouped_f32": "mfma",
         "mpa_knn_f32": "mfma", "mpa_diffattn_fwd_f32": "hbm", "mpa_diffattn_bwd_f32": "hbm"}


def synthetic_batch(B, seed, device):
    """uniform(-1,1) clouds, centred and scaled into the unit sphere; random labels."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, NUM_POINT, 3, generator=g) * 2 - 1
    x = x - x.mean(1, keepdim=True)
    x = x / x.norm(dim=-1).max(dim=1)[0].view(B, 1, 1)
    y = torch.randint(0, NUM_CLASS, (B,), generator=g)
    return x.transpose(1, 2).contiguous().to(device), y.to(device)


def host_cores():
    """CPU cores this process may actually use: affinity mask, capped by the cgroup CPU quota
    (os.cpu_count() reports the whole host and oversubscribes a container badly)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(float(quota) / float(period))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, q // per))
    except (OSError, ValueError):
        pass
    return n


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def forward_only(model, x, feeder, iters=50):
    """Forward pass alone (train-mode BatchNorm statistics, no autograd), captured as a HIP graph: a
    secondary figure next to the headline fwd+bwd metric (SURVEY 8d states the >= 20x target on it)."""
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                feeder.begin_pass()
                model(x)
                feeder.end_pass()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            feeder.begin_pass()
            model(x)
            feeder.end_pass()
        for _ in range(5):
            feeder.refill()
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            feeder.refill()
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
    return {"value": x.shape[0] / dt, "unit": "point-clouds/s", "ms_per_step": dt * 1e3, "steps": iters}


def cpu_baseline(batch, steps=3):
    """The oracle's plain-PyTorch restatement of the reference path, fwd+bwd+Adam on host cores."""
    from oracle import ref_cpu as R
    threads = int(os.environ.get("MPA_CPU_THREADS", host_cores()))
    log("cpu baseline on %d threads (os.cpu_count()=%s)" % (threads, os.cpu_count()))
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    args = argparse.Namespace(num_point=NUM_POINT, return_dist=True, cuda_ops=False, num_class=NUM_CLASS)
    model = R.ClsModel(args).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    x, y = synthetic_batch(batch, 1234, "cpu")
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss = R.smooth_cls_loss(model(x), y)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        log("cpu baseline step %d: %.2f s" % (i, times[-1]))
    times = sorted(times[1:])
    med = times[len(times) // 2]
    ftimes = []
    with torch.no_grad():                                  # forward only (the >= 20x target is on forward)
        for i in range(steps + 1):
            t0 = time.perf_counter()
            model(x)
            ftimes.append(time.perf_counter() - t0)
    fmed = sorted(ftimes[1:])[len(ftimes[1:]) // 2]
    log("cpu baseline forward only: %.2f s" % fmed)
    return {"value": batch / med, "unit": "point-clouds/s", "cores": threads, "kind": "port",
            "sample": "oracle/ref_cpu.py ClsModel fwd+bwd+Adam, batch %d x %d pts, 1 warm-up + %d timed steps, median"
                      % (batch, NUM_POINT, steps),
            "forward_only_value": batch / fmed}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="clouds per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python (no HIP graph)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    local = local % torch.cuda.device_count()      # (rehearsals may stack ranks on one card)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import mpa_amd  # noqa: F401
    from mpa_amd import ops
    from mpa_amd import distributed as mdist
    from mpa_amd.models.repsurf.repsurf_ssg_umb import Model, SmoothClsLoss

    if world > 1:
        mdist.init_process_group(os.environ.get("MPA_DIST_BACKEND"))
    torch.manual_seed(0)
    args = argparse.Namespace(num_point=NUM_POINT, return_dist=True, cuda_ops=True, num_class=NUM_CLASS)
    model = Model(args).to(dev).train()
    crit = SmoothClsLoss()
    x, y = synthetic_batch(a.batch, 1234 + rank, dev)
    if a.eager:
        reducer = mdist.GradReducer(model) if world > 1 else None
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)

        def step():
            if reducer is not None:
                reducer.zero_grad()
            else:
                opt.zero_grad(set_to_none=True)
            loss = crit(model(x), y)
            loss.backward()
            if reducer is not None:
                reducer.all_reduce()
            opt.step()
            return loss
    else:
        # the whole forward+backward is one HIP graph; all-reduce + (graphed) Adam follow it
        from mpa_amd.runtime import GraphedTrainStep
        graphed = GraphedTrainStep(model, crit, (x, y), lr=1e-3)      # optim.FlatAdam on the flat buckets

        def step():
            return graphed(x, y)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    log("rank %d: warm-up done" % rank)
    if a.eager:
        ops.enable_kernel_timing(list(TIMED))
    mdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    mdist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = mdist.max_over_ranks(elapsed, dev)
    log("rank %d: %d steps in %.3f s" % (rank, a.steps, elapsed))
    assert torch.isfinite(loss).item(), "loss is not finite"
    if not a.eager and rank == 0:
        # A replayed HIP graph has no per-kernel event hooks: the kernels are timed live, with HIP
        # events on their launch stream, in an eagerly launched pass over the same step right
        # after the timed region (same shapes, same data).
        ops.enable_kernel_timing(list(TIMED))
        for _ in range(min(a.steps, 10)):
            graphed._fwd_bwd()
    kt = ops.kernel_timing_results()
    ops.disable_kernel_timing()
    pair_us = None
    if rank == 0:       # what an (empty) HIP event pair itself costs on this stream: the bracket's overhead
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
        for e0, e1 in evs:
            e0.record()
            e1.record()
        torch.cuda.synchronize()
        pair_us = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)[50]

    if rank == 0:
        clouds = a.batch * world * a.steps
        traffic = {}
        try:        # HBM bytes per launch from the committed rocprofv3 PMC passes (see the file's "how")
            with open(os.path.join(ROOT, "profiles", "r01f_pmc_traffic.json")) as fh:
                traffic = {k: v.get("hbm_bytes_per_launch") for k, v in json.load(fh)["kernels"].items()}
        except (OSError, ValueError, KeyError):
            pass
        kernels = []
        for name, bound in TIMED.items():
            r = kt.get(name)
            if not r or not r["launches"]:
                continue
            sec = r["ms"] / 1e3
            if bound == "hbm":
                ach, peak, unit = r["algo_bytes"] / sec / 1e9, HBM_PEAK_GBS, "GB/s"
            else:
                ach, peak, unit = r["algo_flops"] / sec / 1e12, MFMA_F32_PEAK_TFLOPS, "TFLOP/s"
            kernels.append({"kernel": name, "bound": bound, "achieved": ach, "peak": peak, "unit": unit,
                            "frac": ach / peak, "traffic": traffic.get(name), "launches": r["launches"],
                            "avg_launch_us": r["ms"] * 1e3 / r["launches"], "total_ms": r["ms"],
                            "algo_bytes_per_launch": r["algo_bytes"] / r["launches"],
                            "algo_flops_per_launch": r["algo_flops"] / r["launches"]})
        kernels.sort(key=lambda k: -k["total_ms"])
        roof = dict(kernels[0]) if kernels else None       # the dominant kernel by measured time
        if roof:
            roof["empty_event_pair_us"] = pair_us     # included in avg_launch_us (not subtracted): frac is a lower bound
            roof["measured"] = ("HIP events around every launch of the kernel, " +
                                ("inside the timed region" if a.eager else
                                 "eager pass over the same step right after the timed (graph-replayed) region"))
        line = {
            "metric": "point-clouds/sec fwd+bwd, ModelNet40 1024pt cls", "value": clouds / elapsed,
            "unit": "point-clouds/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ModelNet40-shaped classification, 1024 points, batch %d per GPU, fp32, "
                                   "fwd+loss+bwd+Adam (BASELINE configs[1])" % a.batch,
                       "points": NUM_POINT, "batch_per_gpu": a.batch, "global_batch": a.batch * world,
                       "parallelism": "dp%d" % world, "launch": "eager" if a.eager else "hipgraph"},
            "roofline": roof,
            "roofline_other_kernels": kernels[1:],
        }
        if world == 1 and not a.eager:
            line["forward_only"] = forward_only(model, x, graphed.feeder)
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.batch)
        print(json.dumps(line), flush=True)
    mdist.shutdown()


if __name__ == "__main__":
    main()
