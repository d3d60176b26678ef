"""The evaluation harness (mpa_amd.eval) on the device with the real models: voting classification
(reference tool/test_classification.py:114-162) and part-segmentation voting + accuracy / mIoU
(tool/test_partseg.py:134-199) run on CUDA tensors through the HIP path, against (a) the CPU oracle models driven by a
loop-by-loop restatement of the reference's test scripts and (b) the numpy restatement of the metric loops applied to
the very predictions the device produced (metrics must be equal, not close)."""
from argparse import Namespace

import numpy as np
import pytest
import torch

from param_fill import fill_state, unit_cloud
from test_eval_cpu import _ref_partseg_metrics, _ref_pointscale

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ev():
    import mpa_amd  # noqa: F401
    from mpa_amd import eval as ev
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    return ev


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def test_vote_classification_and_meter_on_device(ev):
    from mpa_amd.models.repsurf.repsurf_ssg_umb import Model
    from oracle import ref_cpu as R
    B, N, NC, votes = 6, 1024, 40, 3
    args = Namespace(num_point=N, return_dist=True, cuda_ops=True, num_class=NC)
    gpu = fill_state(Model(args), seed=2).cuda()
    cpu = fill_state(R.ClsModel(Namespace(num_point=N, return_dist=True, cuda_ops=False, num_class=NC)), seed=2).eval()
    pts = unit_cloud(B, N, seed=9).transpose(1, 2).contiguous()
    target = torch.arange(B) % 5
    # the device run: same CPU-generator draws (FPS starts: torch; vote scales: numpy) as the restated loop below
    torch.manual_seed(3)
    np.random.seed(7)
    got = ev.vote_classification(gpu, pts.clone().cuda(), vote_num=votes)
    assert got.is_cuda and got.shape == (B, NC)
    torch.manual_seed(3)
    np.random.seed(7)
    p = pts.clone()
    pool = torch.zeros(B, NC)
    with torch.no_grad():
        for v in range(votes):              # tool/test_classification.py:123-131
            if v > 0:
                p = _ref_pointscale(p, 0.95, 1.05)
            pool += cpu(p)
    want = pool / votes
    assert _rel(got.cpu(), want) < 1e-3 and torch.equal(got.argmax(1).cpu(), want.argmax(1))
    # the meter on device tensors == the reference's accumulation on the same predictions
    meter = ev.ClassificationMeter(NC)
    meter.update(got, target.cuda())
    choice = got.cpu().max(1)[1]
    class_acc = np.zeros((NC, 2))
    for cat in np.unique(target.numpy()):
        class_acc[cat, 0] += choice[target == cat].eq(target[target == cat]).sum().item() / float((target == cat).sum().item())
        class_acc[cat, 1] += 1
    ins, cls = meter.result()
    assert ins == choice.eq(target).sum().item() / float(B)
    with np.errstate(invalid="ignore"):
        ref_cls = np.mean(class_acc[:, 0] / class_acc[:, 1])
    assert (np.isnan(cls) and np.isnan(ref_cls)) or cls == ref_cls


def test_vote_partseg_and_meter_on_device(ev):
    from mpa_amd.models.repsurf.pointnet2_part_seg_msg import get_model
    from oracle import ref_cpu as R
    B, N, votes = 2, 2048, 2
    gpu = fill_state(get_model(50), seed=1).cuda()
    cpu = fill_state(R.PartSegModel(50), seed=1).eval()
    pts = unit_cloud(B, N, seed=19).transpose(1, 2).contiguous()
    obj = torch.tensor([12, 3])                                    # Airplane (parts 0-3), Car (parts 8-11) in SEG_CLASSES order
    cats = list(ev.SEG_CLASSES)
    g = torch.Generator().manual_seed(5)
    target = torch.stack([torch.tensor(ev.SEG_CLASSES[cats[int(o)]])[torch.randint(0, len(ev.SEG_CLASSES[cats[int(o)]]), (N,), generator=g)]
                          for o in obj])
    torch.manual_seed(4)
    np.random.seed(8)
    got = ev.vote_partseg(gpu, pts.clone().cuda(), obj.view(B, 1).cuda(), num_classes=16, num_votes=votes)
    assert got.is_cuda and got.shape == (B, N, 50)
    torch.manual_seed(4)
    np.random.seed(8)
    p = pts.clone()
    pool = torch.zeros(B, N, 50)
    with torch.no_grad():
        for v in range(votes):              # tool/test_partseg.py:148-154
            if v > 0:
                p = _ref_pointscale(p, 0.95, 1.05)
            pool += cpu(p, torch.eye(16)[obj].view(B, 1, 16))[0]
    want = pool / votes
    assert _rel(got.cpu(), want) < 2e-3
    assert float((got.argmax(-1).cpu() == want.argmax(-1)).float().mean()) > 0.995
    for quirk in (True, False):
        meter = ev.PartSegMeter(reference_quirk=quirk)
        meter.update(got, target.cuda())
        res, ref = meter.result(), _ref_partseg_metrics(ev, [(got.cpu(), target)], quirk)
        for k, v in ref.items():
            assert res[k] == v or (np.isnan(res[k]) and np.isnan(v)), (k, res[k], v)
