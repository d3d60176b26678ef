"""The evaluation harness (mpa_amd.eval) against a loop-by-loop numpy restatement of
the reference's test scripts (tool/test_classification.py:114-162, tool/test_partseg.py:118-199).
Device-agnostic tensor code, exercised here on CPU tensors; the metrics must be equal, not close."""
import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def ev():
    import mpa_amd  # noqa: F401
    from mpa_amd import eval as ev
    return ev


def _ref_pointscale(pc, low, high):
    """test_classification.py:73-79 on a [B, C, N] batch (scales the first three points)."""
    for i in range(pc.shape[0]):
        xyz = np.random.uniform(low=low, high=high, size=[3])
        pc[i, :, 0:3] = pc[i, :, 0:3] * torch.from_numpy(xyz).float()
    return pc


class _Toy(torch.nn.Module):
    def __init__(self, n_out):
        super().__init__()
        self.w = torch.nn.Parameter(torch.randn(3, n_out))

    def forward(self, pts, label=None):
        f = pts.transpose(1, 2) @ self.w                   # [B, N, n_out]
        if label is None:
            return f.mean(1)
        return f + label.sum(-1).view(-1, 1, 1), None


def test_vote_classification_matches_reference_loop(ev):
    torch.manual_seed(0)
    model = _Toy(15)
    pts = torch.randn(6, 3, 64)
    np.random.seed(7)
    got = ev.vote_classification(model, pts.clone(), vote_num=4)
    np.random.seed(7)
    p = pts.clone()
    pool = torch.zeros(6, 15)
    with torch.no_grad():
        for v in range(4):
            if v > 0:
                p = _ref_pointscale(p, 0.95, 1.05)
            pool += model.eval()(p)
    assert torch.equal(got, pool / 4)


def test_classification_meter_equals_reference_accumulation(ev):
    g = torch.Generator().manual_seed(3)
    nc = 15
    meter = ev.ClassificationMeter(nc)
    class_acc = np.zeros((nc, 3))
    mean_correct = []
    for _ in range(5):
        pred = torch.randn(40, nc, generator=g)
        target = torch.randint(0, nc - 2, (40,), generator=g)          # two classes never occur -> nan, as in the reference
        meter.update(pred, target)
        choice = pred.max(1)[1]
        for cat in np.unique(target.numpy()):
            acc = choice[target == cat].eq(target[target == cat]).sum()
            class_acc[cat, 0] += acc.item() / float((target == cat).sum().item())
            class_acc[cat, 1] += 1
        mean_correct.append(choice.eq(target).sum().item() / float(40))
    with np.errstate(invalid="ignore"):
        class_acc[:, 2] = class_acc[:, 0] / class_acc[:, 1]
    ins, cls = meter.result()
    assert ins == np.mean(mean_correct)
    assert (np.isnan(cls) and np.isnan(np.mean(class_acc[:, 2]))) or cls == np.mean(class_acc[:, 2])
    # with every class present the class accuracy is a number and equal
    meter2 = ev.ClassificationMeter(4)
    pred = torch.randn(64, 4, generator=g)
    target = torch.arange(64) % 4
    meter2.update(pred, target)
    ch = pred.max(1)[1]
    want = np.mean([(ch[target == c] == c).float().mean().item() for c in range(4)])
    assert abs(meter2.result()[1] - want) < 1e-7


def _ref_partseg_metrics(ev, batches, quirk=True):
    seg_classes = ev.SEG_CLASSES
    num_part = 50
    total_correct = total_seen = 0
    seen = [0] * num_part
    corr = [0] * num_part
    shape_ious = {cat: [] for cat in seg_classes}
    l2c = {l: c for c, ls in seg_classes.items() for l in ls}
    for logits, target in batches:
        logits, target = logits.numpy(), target.numpy()
        B, N, _ = logits.shape
        pred = np.zeros((B, N), dtype=np.int32)
        for i in range(B):
            cat = l2c[target[i, 0]]
            pred[i] = np.argmax(logits[i][:, seg_classes[cat]], 1) + (0 if quirk else seg_classes[cat][0])
        total_correct += np.sum(pred == target)
        total_seen += B * N
        for l in range(num_part):
            seen[l] += np.sum(target == l)
            corr[l] += np.sum((pred == l) & (target == l))
        for i in range(B):
            cat = l2c[target[i, 0]]
            ious = []
            for l in seg_classes[cat]:
                if np.sum(target[i] == l) == 0 and np.sum(pred[i] == l) == 0:
                    ious.append(1.0)
                else:
                    ious.append(np.sum((target[i] == l) & (pred[i] == l)) / float(np.sum((target[i] == l) | (pred[i] == l))))
            shape_ious[cat].append(np.mean(ious))
    all_ious = [x for c in shape_ious for x in shape_ious[c]]
    with np.errstate(invalid="ignore", divide="ignore"):
        cat_iou = {c: np.mean(v) for c, v in shape_ious.items()}
        return {"accuracy": total_correct / float(total_seen),
                "class_avg_accuracy": np.mean(np.array(corr) / np.array(seen, dtype=np.float64)),
                "class_avg_iou": np.mean(list(cat_iou.values())), "inctance_avg_iou": np.mean(all_ious)}


@pytest.mark.parametrize("quirk", [True, False])
def test_partseg_meter_equals_reference_loops(ev, quirk):
    g = torch.Generator().manual_seed(11)
    cats = list(ev.SEG_CLASSES)
    batches = []
    for b in range(3):
        B, N = 2 * len(cats), 96
        logits = torch.randn(B, N, 50, generator=g)
        target = torch.empty(B, N, dtype=torch.long)
        for i in range(B):
            parts = ev.SEG_CLASSES[cats[i % len(cats)]]
            # every cloud's labels lie in its category; the last part is sometimes absent
            hi = len(parts) if i % 3 else max(1, len(parts) - 1)
            target[i] = torch.tensor(parts)[torch.randint(0, hi, (N,), generator=g)]
            # make the prediction agree with the target on about half of the points
            agree = torch.rand(N, generator=g) < 0.5
            logits[i, agree, target[i, agree]] += 10.0
        batches.append((logits, target))
    meter = ev.PartSegMeter(reference_quirk=quirk)
    for logits, target in batches:
        meter.update(logits, target)
    got, want = meter.result(), _ref_partseg_metrics(ev, batches, quirk)
    for k, v in want.items():
        assert got[k] == v or (np.isnan(got[k]) and np.isnan(v)), (k, got[k], v)
    if not quirk:
        assert got["accuracy"] > 0.45 and got["inctance_avg_iou"] > 0.2
