"""Worker of tests/test_gpu_rccl_capture.py (a fresh process: the process group must exist before anything else
touches the GPU).  One-rank `nccl` (= RCCL) group; GraphedTrainStep(split_after=la4, capture_reduce=True) captures the
early bucket's all-reduce INSIDE the step's HIP graph (and the remaining buckets' at its end); three replays; the
gradients must equal those of the plain (unsplit, reduce-after-replay) step on the same batch and sampling starts.
Exits non-zero on any failure."""
import os
import sys
from argparse import Namespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1], RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                  MPA_DEBUG_SKIP_OPT="1", MPA_CAPTURE_REDUCE_SINGLE_RANK="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

import mpa_amd  # noqa: E402,F401
from mpa_amd import distributed as md, ops  # noqa: E402
from mpa_amd.models.repsurf.repsurf_ssg_umb import Model, SmoothClsLoss  # noqa: E402
from mpa_amd.runtime import GraphedTrainStep  # noqa: E402
from param_fill import fill_state, unit_cloud  # noqa: E402


def grads(split, capture):
    # fixed parameters (lr = 0: the constructor's warm-up Adam steps must not move them -- Adam's first steps are
    # sign(g) * lr, i.e. rounding noise in near-zero gradients would be amplified into different parameters) and
    # bit-reproducible BatchNorm statistics, so both runs see the same forward pass
    ops.set_deterministic(True)
    torch.manual_seed(0)
    model = fill_state(Model(Namespace(num_point=1024, return_dist=True, cuda_ops=True, num_class=40)), seed=1).cuda().train()
    model.drop1.p = model.drop2.p = 0.0
    x = unit_cloud(8, 1024, seed=3).transpose(1, 2).contiguous().cuda()
    y = (torch.arange(8) % 40).cuda()
    step = GraphedTrainStep(model, SmoothClsLoss(), (x, y), lr=0.0, split_after=model.keepHigh.la4 if split else None,
                            capture_reduce=capture)
    assert step.capture_reduce == capture
    step.feeder.frozen = True                    # every replay samples from the same first points
    try:
        losses = [float(step(x, y)) for _ in range(3)]
        torch.cuda.synchronize()
        out = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        early = len(step.reducer.early)
    finally:
        step.close()
    return losses, out, early


def main():
    torch.cuda.set_device(0)
    md.init_process_group("nccl")
    assert md.is_dist() and md.world_size() == 1
    l_cap, g_cap, early = grads(True, True)
    l_ref, g_ref, _ = grads(False, False)
    assert early >= 1, "no early bucket was formed at the split point"
    assert all(abs(a - b) < 1e-5 for a, b in zip(l_cap, l_ref)), (l_cap, l_ref)
    assert g_cap.keys() == g_ref.keys()
    worst = 0.0
    for n in g_ref:
        assert torch.isfinite(g_cap[n]).all(), n
        # (atomically accumulated BatchNorm statistics: two runs of one step agree to fp32 noise, not bit for bit)
        err = float((g_cap[n] - g_ref[n]).norm() / g_ref[n].norm().clamp_min(1e-12))
        worst = max(worst, err)
        assert err < 1e-3 or float((g_cap[n] - g_ref[n]).abs().max()) < 1e-6, (n, err)
    print("rccl-capture ok: %d parameters, early buckets %d, worst relative gradient difference %.2e, losses %s"
          % (len(g_ref), early, worst, l_cap))
    md.shutdown()


if __name__ == "__main__":
    main()
