"""bench.py -- point-clouds/sec, forward+backward, on N MI355X GPUs of one node.

A "step" is one training pass of the hot path over one batch of synthetic clouds: forward (FPS -> kNN
grouping -> difference-wise attention -> transition MLPs -> head / decoder), the label-smoothed loss,
backward, gradient all-reduce over RCCL when N > 1, and an Adam step.  Inputs are resident in HBM before
the timed region; weak scaling (per-GPU batch fixed).

    python bench.py --gpus 1 --steps 20 --warmup 5                       # default: BASELINE configs[1]
    python bench.py --config partseg-bf16                                # BASELINE configs[2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

--config   cls-fp32         ModelNet40-shaped classification, 1024 points, batch 64/GPU, fp32   (BASELINE configs[1]; default,
                            the configuration BASELINE.json's metric is quoted on)
           cls-bf16         the same model on bf16 features
           partseg-fp32     ShapeNetPart-shaped part segmentation, 2048 points, batch 32/GPU, fp32
           partseg-bf16     the same on bf16 features                                            (BASELINE configs[2])
           s3dis-fp32       S3DIS-shaped semantic segmentation: the part-seg encoder-decoder wiring on 4096-point
                            blocks (4096 -> 2048 -> 1024 -> 512 -> 256), 13 classes, batch 16/GPU, fp32  (configs[3])
           completion-bf16  completion decoder: upsample + LocalMerge, 1024 -> ... -> 16,384 output points,
                            batch 8/GPU, bf16 features                                           (configs[4])
           completion-fp32  the same at fp32

Rank 0 prints ONE JSON line.  Extra objects: "roofline" for the dominant kernel (HIP events on the launch
stream around that kernel's launches), "cpu_baseline" (the CPU oracle oracle/ref_cpu.py timed on this box's
host cores on a bounded sample; rank 0, N=1, cls configs), at N=1 "forward_only" (the forward pass alone
as a HIP graph, measured after the timed region: a secondary figure) and -- in the default run (N=1, default
config) -- "soak" (400 more replayed steps of the headline workload, same process, same graph) and
"other_configs": the other BASELINE configurations timed in the same process after the headline, each with
its own throughput, ms/step and dominant-kernel roofline (--no-others skips them).
"""
import argparse
import gc
import glob
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3     # dense fp32 MFMA peak (MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32)
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak (MI355X_MICROARCH.md; the 5 PF figure is 2:1 sparse)

CONFIGS = {
    # name: (task, dtype, points, default batch, BASELINE.json config)
    "cls-fp32": ("cls", "f32", 1024, 64, "BASELINE configs[1]"),
    "cls-bf16": ("cls", "bf16", 1024, 64, "configs[1]'s model on bf16 features"),
    "partseg-fp32": ("partseg", "f32", 2048, 32, "configs[2]'s model at fp32"),
    "partseg-bf16": ("partseg", "bf16", 2048, 32, "BASELINE configs[2]"),
    "s3dis-fp32": ("s3dis", "f32", 4096, 16, "BASELINE configs[3]"),
    "completion-bf16": ("completion", "bf16", 16384, 8, "BASELINE configs[4]"),
    "completion-fp32": ("completion", "f32", 16384, 8, "configs[4]'s chain at fp32"),
}
OTHER_CONFIGS = ("partseg-bf16", "s3dis-fp32", "completion-bf16", "cls-bf16", "partseg-fp32")   # after the headline, N=1
# kernels timed with HIP events for the roofline leg: name -> bound.  fp32: the 64x64-tile GEMM is priced
# against the fp32 MFMA peak, the short-K kernel (K <= 128 layers, 16 FLOP/B) against HBM.  bf16: every
# product of the path is below the machine balance (32..200 FLOP/B against ~310), so all are priced on HBM.
TIMED = {
    "f32": {"mpa_gemm_f32/tiled": "mfma", "mpa_gemm_f32/shortk": "hbm", "mpa_gemm_tn_grouped_f32": "mfma",
            "mpa_gemm_grouped_f32": "mfma", "mpa_knn_f32": "mfma", "mpa_diffattn_fwd_f32": "hbm", "mpa_diffattn_bwd_f32": "hbm"},
    "bf16": {"mpa_gemm_bf16": "hbm", "mpa_gemm_tn_grouped_bf16": "hbm", "mpa_gemm_grouped_bf16": "hbm",
             "mpa_knn_f32": "mfma",
             "mpa_diffattn_fwd_bf16": "hbm", "mpa_diffattn_bwd_bf16": "hbm"},
}
NUM_CLASS, NUM_PART, NUM_OBJ, NUM_SEM = 40, 50, 16, 13


def unit_clouds(B, N, g):
    """uniform(-1,1) clouds, centred and scaled into the unit sphere."""
    x = torch.rand(B, N, 3, generator=g) * 2 - 1
    x = x - x.mean(1, keepdim=True)
    return (x / x.norm(dim=-1).max(dim=1)[0].view(B, 1, 1)).transpose(1, 2).contiguous()


def synthetic_batch(task, B, N, seed, device):
    g = torch.Generator().manual_seed(seed)
    x = unit_clouds(B, N, g)
    if task == "cls":
        return (x.to(device), torch.randint(0, NUM_CLASS, (B,), generator=g).to(device))
    if task == "completion":
        # dense clouds in sampling order (every prefix = the FPS state of that size): the data-preparation step of
        # models/completion.py, done once here like the offline FPS of dataset/ModelNetDataLoader.py
        if str(device) == "cpu":
            return (x,)
        from mpa_amd.models.completion import sampling_order
        start = torch.randint(0, N, (B,), generator=g)
        return (sampling_order(x.transpose(1, 2).contiguous().to(device), start_idx=start).transpose(1, 2).contiguous(),)
    label = torch.zeros(B, 1, NUM_OBJ)
    label[torch.arange(B), 0, torch.randint(0, NUM_OBJ, (B,), generator=g)] = 1
    target = torch.randint(0, NUM_SEM if task == "s3dis" else NUM_PART, (B, N), generator=g)
    return (x.to(device), label.to(device), target.to(device))


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


def kernel_sources_sha():
    """sha256 (16 hex digits) over the kernel sources: ties an offline PMC profile to the code it measured."""
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "markov-process-analysis-on-point-cloud_amd", "csrc", "*.h*"))):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def forward_only(run_forward, feeder, arena, batch, iters=50):
    """Forward pass alone (train-mode BatchNorm statistics, no autograd), captured as a HIP graph: a
    secondary figure next to the headline fwd+bwd metric (SURVEY 8d states the >= 20x target on it)."""
    from mpa_amd import ops

    def one_pass():             # (the sampling chain runs inside the pass: no step's prefetch object is installed here)
        feeder.begin_pass()
        arena.begin()
        ops.set_arena(arena)
        try:
            run_forward()
        finally:
            ops.set_arena(None)
            arena.end()
            feeder.end_pass()

    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                one_pass()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            one_pass()
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
    return {"value": batch / dt, "unit": "point-clouds/s", "ms_per_step": dt * 1e3, "steps": iters}


def cpu_baseline(task, batch, npoint, warm=1, steps=3):
    """The oracle's plain-PyTorch restatement of the reference path, fwd+bwd+Adam on host cores (fp32: the
    reference has no reduced-precision mode)."""
    from oracle import ref_cpu as R
    threads = int(os.environ.get("MPA_CPU_THREADS", host_cores()))
    log("cpu baseline (%s, batch %d) on %d threads (os.cpu_count()=%s)" % (task, batch, threads, os.cpu_count()))
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    if task == "cls":
        args = argparse.Namespace(num_point=npoint, return_dist=True, cuda_ops=False, num_class=NUM_CLASS)
        model = R.ClsModel(args).train()
        x, y = synthetic_batch(task, batch, npoint, 1234, "cpu")
        fwd = lambda: model(x)                                   # noqa: E731
        loss_of = lambda out: R.smooth_cls_loss(out, y)          # noqa: E731
        name = "ClsModel"
    else:
        nc = NUM_SEM if task == "s3dis" else NUM_PART
        model = R.PartSegModel(nc).train()
        x, lab, tgt = synthetic_batch(task, batch, npoint, 1234, "cpu")
        fwd = lambda: model(x, lab)[0]                           # noqa: E731
        loss_of = lambda out: R.partseg_loss(out.reshape(-1, nc), tgt.reshape(-1))        # noqa: E731
        name = "PartSegModel"
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    times = []
    for i in range(warm + steps):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss = loss_of(fwd())
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
        log("cpu baseline step %d: %.2f s" % (i, times[-1]))
    times = sorted(times[warm:])
    med = times[len(times) // 2]
    ftimes = []
    with torch.no_grad():                                  # forward only (the >= 20x target is on forward)
        for i in range(1 + max(2, steps // 2)):
            t0 = time.perf_counter()
            fwd()
            ftimes.append(time.perf_counter() - t0)
    fmed = sorted(ftimes[1:])[len(ftimes[1:]) // 2]
    log("cpu baseline forward only: %.2f s" % fmed)
    return {"value": batch / med, "unit": "point-clouds/s", "cores": threads, "kind": "port",
            "sample": "oracle/ref_cpu.py %s fp32 fwd+bwd+Adam, batch %d x %d pts, %d warm-up + %d timed steps, median"
                      % (name, batch, npoint, warm, steps),
            "forward_only_value": batch / fmed}


def build_workload(name, batch, rank, dev):
    """model, loss, resident synthetic batch and the callables of one configuration."""
    task, dt, npoint, dbatch, which = CONFIGS[name]
    torch.manual_seed(0)
    data = synthetic_batch(task, batch, npoint, 1234 + rank, dev)
    # a second, distinct resident batch: steps alternate between the two, and each step announces the other as the next
    # one (its sampling chain is computed a step ahead, inside this step's weight-gradient launches)
    data2 = synthetic_batch(task, batch, npoint, 4321 + rank, dev)
    w = {"task": task, "dt": dt, "npoint": npoint, "batch": batch, "data": data, "data2": data2, "compute_loss": None,
         "split": None, "has_chain": task != "completion"}
    if task == "cls":
        from mpa_amd.models.repsurf.repsurf_ssg_umb import Model, SmoothClsLoss
        args = argparse.Namespace(num_point=npoint, return_dist=True, cuda_ops=True, num_class=NUM_CLASS)
        model = Model(args).to(dev).train()
        crit = SmoothClsLoss()
        w["run_forward"] = lambda: model(data[0])
        w["loss_eager"] = lambda: crit(model(data[0]), data[1])
        w["split"] = model.keepHigh.la4
        w["metric"] = "point-clouds/sec fwd+bwd, ModelNet40 1024pt cls"
        workload = "ModelNet40-shaped classification, %d points, batch %d per GPU" % (npoint, batch)
    elif task == "completion":
        from mpa_amd.models.completion import CompletionDecoder, CoordinateLoss
        model = CompletionDecoder().to(dev).train()
        crit = CoordinateLoss()

        def compute_loss(model, crit, x):
            return crit(model(x), x)

        w["compute_loss"] = compute_loss
        w["run_forward"] = lambda: model(data[0])
        w["loss_eager"] = lambda: compute_loss(model, crit, *data)
        w["metric"] = "point-clouds/sec fwd+bwd, completion decoder to 16384 points"
        workload = ("completion decoder (upsample + LocalMerge chain 1024 -> 2048 -> 4096 -> 8192 -> %d output points, "
                    "clouds resident in sampling order), batch %d per GPU" % (npoint, batch))
    else:
        from mpa_amd.models.repsurf.pointnet2_part_seg_msg import get_model, get_loss
        nc = NUM_SEM if task == "s3dis" else NUM_PART
        model = get_model(nc).to(dev).train()
        crit = get_loss()

        def compute_loss(model, crit, x, label, target):
            pred, _ = model(x, label)
            return crit(pred.reshape(-1, nc), target.reshape(-1))

        w["compute_loss"] = compute_loss
        w["run_forward"] = lambda: model(data[0], data[1])
        w["loss_eager"] = lambda: compute_loss(model, crit, *data)
        if task == "s3dis":
            w["metric"] = "point-clouds/sec fwd+bwd, S3DIS-shaped 4096pt blocks sem-seg"
            workload = ("S3DIS-shaped semantic segmentation (part-seg encoder-decoder wiring, states 4096 -> 2048 -> 1024 -> "
                        "512 -> 256, %d classes), %d-point blocks, batch %d per GPU" % (nc, npoint, batch))
        else:
            w["metric"] = "point-clouds/sec fwd+bwd, ShapeNetPart 2048pt part-seg"
            workload = "ShapeNetPart-shaped part segmentation, %d points, batch %d per GPU" % (npoint, batch)
    w["model"], w["crit"] = model, crit
    w["workload"] = workload + ", %s, fwd+loss+bwd+Adam (%s)" % (
        "fp32" if dt == "f32" else "bf16 features / fp32 accumulate, statistics, coordinates and parameters", which)
    return w


def run_config(name, a, world, rank, dev, steps, warmup, headline):
    """Time `steps` steps of one configuration (after `warmup` untimed ones) and price its kernels.  headline: the
    line the driver reads (barriers, max over ranks, forward-only / soak / CPU legs); otherwise a compact record
    for "other_configs" (N=1 only)."""
    from mpa_amd import ops
    from mpa_amd import distributed as mdist
    task, dt, npoint, dbatch, which = CONFIGS[name]
    batch = (a.batch or dbatch) if headline else dbatch
    old_dtype = ops.set_feature_dtype(torch.bfloat16 if dt == "bf16" else torch.float32)
    w = build_workload(name, batch, rank, dev)
    model, crit, data = w["model"], w["crit"], w["data"]
    graphed = None
    cap = False
    if a.eager and headline:
        reducer = mdist.GradReducer(model) if world > 1 else None
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)

        def step():
            if reducer is not None:
                reducer.zero_grad()
            else:
                opt.zero_grad(set_to_none=True)
            loss = w["loss_eager"]()
            loss.backward()
            if reducer is not None:
                reducer.all_reduce()
            opt.step()
            return loss
    else:
        # the whole forward+backward is one HIP graph; all-reduce + (graphed) Adam follow it
        from mpa_amd.runtime import GraphedTrainStep
        # MPA_CAPTURE_REDUCE=1 (N > 1): flush the weight gradients of head / la5 / la4 (96 % of the bytes) early and
        # all-reduce them INSIDE the captured graph, overlapped with the rest of backward.  Off by default: RCCL under
        # graph capture has run on one rank only (tests/test_gpu_rccl_capture.py), never across GPUs.
        cap = world > 1 and os.environ.get("MPA_CAPTURE_REDUCE") == "1" and w["split"] is not None
        # Cross-step geometry (ops.GeometryPipeline): the NEXT batch's sampling chain and state-0 search ride in this
        # batch's search launches (the step announces its next batch).  On by default where the shapes allow (clouds of
        # <= 2048 points): 3.51 against 3.56 ms per cls-fp32 step.  MPA_PREFETCH=0: the chain inside the pass;
        # MPA_PREFETCH=riders: the first form (riders in the weight-gradient launches: slower).
        mode = os.environ.get("MPA_PREFETCH", "1")
        prefetch = False
        if w["has_chain"] and not cap and mode != "0":
            prefetch = "riders" if mode == "riders" else True
        graphed = GraphedTrainStep(model, crit, data, lr=1e-3, compute_loss=w["compute_loss"],
                                   split_after=w["split"] if cap else None, capture_reduce=cap,
                                   prefetch_geometry=prefetch)                                    # optim.FlatAdam
        pair = (data, w["data2"])
        turn = [0]

        def step(announce=True):
            cur, nxt = pair[turn[0] & 1], pair[(turn[0] + 1) & 1]
            turn[0] += 1
            return graphed(*cur, next_batch=nxt if announce else None)

    eager = a.eager and headline
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    log("rank %d: %s warm-up done" % (rank, name))
    timed = dict(TIMED[dt])
    timed.update({"mpa_fps_knn_xyz_f32": None, "mpa_fps_f32": None})      # the sampling chain: reported as `fps`, not priced
    if eager:
        ops.enable_kernel_timing(list(timed))
    if headline:
        mdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    if headline:
        mdist.barrier()
    elapsed = time.perf_counter() - t0
    if headline:
        elapsed = mdist.max_over_ranks(elapsed, dev)
    log("rank %d: %s: %d steps in %.3f s" % (rank, name, steps, elapsed))
    assert torch.isfinite(loss).item(), "loss is not finite"
    soak = None
    if headline and world == 1 and not eager and a.soak > 0:
        # a longer run of the same replayed step in the same process (the timed region above is ~0.1 s)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(a.soak):
            loss = step()
        torch.cuda.synchronize()
        dts = time.perf_counter() - t1
        assert torch.isfinite(loss).item(), "loss is not finite after the soak"
        soak = {"steps": a.soak, "ms_per_step": dts / a.soak * 1e3, "value": batch * a.soak / dts, "unit": "point-clouds/s"}
        log("soak: %d steps, %.3f ms/step" % (a.soak, soak["ms_per_step"]))
    n_timed_passes = steps if eager else min(steps, 10 if headline else 3)
    if not eager:
        step(announce=False)        # untimed: leaves the static batch and the buffered geometry on the SAME batch
        torch.cuda.synchronize()
    if not eager and rank == 0 and not cap:     # (with the all-reduce captured in the pass, rank 0 cannot run it alone)
        # A replayed HIP graph has no per-kernel event hooks: the kernels are timed live, with HIP
        # events on their launch stream, in an eagerly launched pass over the same step right
        # after the timed region (same shapes, same data).
        ops.enable_kernel_timing(list(timed))
        fused = ops.set_fps_feature_fusion(False)     # one entry point per event bracket: FPS and the searches apart
        for _ in range(n_timed_passes):
            graphed._fwd_bwd()
        ops.set_fps_feature_fusion(fused)
    kt = ops.kernel_timing_results()
    ops.disable_kernel_timing()
    pair_us = None
    line = None
    if rank == 0:       # what an (empty) HIP event pair itself costs on this stream: the bracket's overhead
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
        for e0, e1 in evs:
            e0.record()
            e1.record()
        torch.cuda.synchronize()
        pair_us = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)[50]

    if rank == 0:
        clouds = batch * world * steps
        # HBM bytes per launch come from committed rocprofv3 PMC passes (collected offline exactly as
        # MI355X_MICROARCH.md prescribes: separate --pmc FETCH_SIZE / WRITE_SIZE runs).  The file records the
        # sha of the kernel sources it measured; if the sources have changed since, traffic is reported as null.
        traffic, traffic_source = {}, None
        src_sha = kernel_sources_sha()
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_pmc_traffic.json" % name.replace("-", "_"))))
        if cands:
            try:
                with open(cands[-1]) as fh:
                    pmc = json.load(fh)
                fresh = pmc.get("kernel_sources_sha") == src_sha
                traffic_source = "%s (offline rocprofv3 --pmc passes; kernel sources %s: %s)" % (
                    os.path.relpath(cands[-1], ROOT), pmc.get("kernel_sources_sha"),
                    "current" if fresh else "STALE, sources are now %s -> traffic withheld" % src_sha)
                if fresh:
                    traffic = {k: v.get("hbm_bytes_per_launch") for k, v in pmc["kernels"].items()}
            except (OSError, ValueError, KeyError):
                pass
        peak_mfma = MFMA_F32_PEAK_TFLOPS if dt == "f32" else MFMA_BF16_PEAK_TFLOPS
        kernels = []
        for kname, bound in timed.items():
            r = kt.get(kname)
            if bound is None or not r or not r["launches"]:
                continue
            sec = r["ms"] / 1e3
            if bound == "hbm":
                ach, peak, unit = r["algo_bytes"] / sec / 1e9, HBM_PEAK_GBS, "GB/s"
            else:
                pk = MFMA_F32_PEAK_TFLOPS if kname.endswith("_f32") or "_f32/" in kname else peak_mfma
                ach, peak, unit = r["algo_flops"] / sec / 1e12, pk, "TFLOP/s"
            kernels.append({"kernel": kname, "bound": bound, "achieved": ach, "peak": peak, "unit": unit,
                            "frac": ach / peak, "traffic": traffic.get(kname), "launches": r["launches"],
                            "avg_launch_us": r["ms"] * 1e3 / r["launches"], "total_ms": r["ms"],
                            "algo_bytes_per_launch": r["algo_bytes"] / r["launches"],
                            "algo_flops_per_launch": r["algo_flops"] / r["launches"]})
        kernels.sort(key=lambda k: -k["total_ms"])
        roof = dict(kernels[0]) if kernels else None       # the dominant kernel by measured time
        if roof:
            roof["traffic_source"] = traffic_source
            roof["empty_event_pair_us"] = pair_us     # included in avg_launch_us (not subtracted): frac is a lower bound
            roof["measured"] = ("HIP events around every launch of the kernel, " +
                                ("inside the timed region" if eager else
                                 "eager pass over the same step right after the timed (graph-replayed) region, with the "
                                 "FPS + search launches of the replayed step issued as their separate entry points"))
        if not headline:
            line = {"metric": w["metric"], "value": clouds / elapsed, "unit": "point-clouds/s", "steps": steps,
                    "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "dtype": dt,
                    "config": {"workload": w["workload"], "name": name, "points": npoint, "batch_per_gpu": batch,
                               "launch": "hipgraph"},
                    "roofline": roof,
                    "roofline_other_kernels": [{k: r[k] for k in ("kernel", "bound", "achieved", "unit", "frac", "launches",
                                                                  "avg_launch_us")} for r in kernels[1:]]}
        else:
            line = {
                "metric": w["metric"], "value": clouds / elapsed,
                "unit": "point-clouds/s", "n_gpus": world, "steps": steps, "warmup": warmup,
                "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": dt, "data": "synthetic",
                "config": {"workload": w["workload"], "name": name, "points": npoint, "batch_per_gpu": batch,
                           "global_batch": batch * world, "parallelism": "dp%d" % world,
                           "launch": "eager" if eager else "hipgraph",
                           "batches": "two distinct resident synthetic batches per rank, alternating every step (copied "
                                      "into the graph's static buffers inside the timed region)" + (
                               "; each step announces the next batch, whose FPS chain and state-0 coordinate search "
                               "ride in this step's %s" % ("search launches (ops.GeometryPipeline)" if type(graphed.prefetch).__name__
                                                           == "GeometryPipeline" else "weight-gradient launches (ops.GeometryPrefetch)")
                               if (graphed is not None and graphed.prefetch is not None) else "")},
                "roofline": roof,
                "roofline_other_kernels": kernels[1:],
            }
            if soak:
                line["soak"] = soak
        fps_recs = [kt[n] for n in ("mpa_fps_knn_xyz_f32", "mpa_fps_f32") if kt.get(n) and kt[n]["launches"]]
        if fps_recs:
            # FPS is a chain of S dependent iterations per cloud (latency bound): iterations per second
            # against the per-iteration floor of its structure (one s_barrier + four dependent LDS round trips, ~650
            # clocks at 2.4 GHz).  The fused launches also carry the xyz search of the previous state.
            its = sum(r["algo_units"] for r in fps_recs)
            sec = sum(r["ms"] for r in fps_recs) / 1e3
            floor_us = 0.27
            line["fps"] = {"iterations_per_s_per_cloud": its / sec, "us_per_iteration": sec / its * 1e6,
                           "floor_us_per_iteration": floor_us, "floor_over_measured": floor_us / (sec / its * 1e6),
                           "iterations_per_step": its / n_timed_passes, "ms_per_step": sec * 1e3 / n_timed_passes}
        kn = kt.get("mpa_knn_f32")
        if kn and kn["launches"]:
            # SURVEY 8(d)'s unit for the grouping kernel: query x base-point distance evaluations (each over the
            # launch's C channels: xyz searches have C = 3, feature-space searches C = 64..256)
            line["knn"] = {"distance_evals_per_s": kn["algo_units"] / (kn["ms"] / 1e3),
                           "distance_evals_per_step": kn["algo_units"] / n_timed_passes,
                           "launches_per_step": kn["launches"] / n_timed_passes,
                           "ms_per_step": kn["ms"] / n_timed_passes}
        if world == 1 and not eager and not a.no_forward_only:
            line["forward_only"] = forward_only(w["run_forward"], graphed.feeder, graphed.arena, batch,
                                                iters=50 if headline else 10)
    # release the configuration: graphs, pools and the FPS hook (the next configuration builds its own)
    if graphed is not None:
        graphed.close()
    ops.clear_knn_memo()
    ops.set_feature_dtype(old_dtype)
    del graphed, model, crit, data, w, step
    gc.collect()
    torch.cuda.empty_cache()
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cls-fp32")
    ap.add_argument("--batch", type=int, default=0, help="clouds per GPU (default: the configuration's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-forward-only", action="store_true", help="skip the secondary forward-only leg (profiling runs)")
    ap.add_argument("--no-others", action="store_true", help="skip the other configurations after the headline")
    ap.add_argument("--others", action="store_true", help="time the other configurations even with a non-default --config")
    ap.add_argument("--soak", type=int, default=400, help="extra replayed steps of the headline after the timed region (0: none)")
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
    from mpa_amd import distributed as mdist

    if world > 1:
        mdist.init_process_group(os.environ.get("MPA_DIST_BACKEND"))
    default_run = a.config == "cls-fp32" and not a.eager and not a.batch
    if not default_run:
        a.soak = 0 if "--soak" not in sys.argv else a.soak
    line = run_config(a.config, a, world, rank, dev, a.steps, a.warmup, headline=True)
    if rank == 0:
        task = CONFIGS[a.config][0]
        if world == 1 and not a.no_cpu_baseline and task in ("cls", "partseg", "s3dis"):
            if task == "cls":
                batch = a.batch or CONFIGS[a.config][3]
                line["cpu_baseline"] = cpu_baseline(task, batch, CONFIGS[a.config][2], warm=2, steps=8)
                # BASELINE configs[0]: the reference's own CPU-runnable case, batch 16 (3 warm-up + 10 timed)
                c0 = cpu_baseline(task, 16, CONFIGS[a.config][2], warm=3, steps=10)
                line["cpu_baseline"]["configs0_batch16"] = {k: c0[k] for k in ("value", "unit", "forward_only_value", "sample")}
            else:
                line["cpu_baseline"] = cpu_baseline(task, 2 if task == "s3dis" else 4, CONFIGS[a.config][2], warm=1, steps=2)
        if world == 1 and ((default_run and not a.no_others) or a.others):
            others = {}
            for name in OTHER_CONFIGS:
                if name == a.config:
                    continue
                try:
                    others[name] = run_config(name, a, 1, 0, dev, steps=10, warmup=3, headline=False)
                except Exception as e:          # a secondary configuration must never cost the headline line
                    log("other config %s failed: %r" % (name, e))
                    others[name] = {"error": repr(e)}
            line["other_configs"] = others
        print(json.dumps(line), flush=True)
    mdist.shutdown()


if __name__ == "__main__":
    main()
