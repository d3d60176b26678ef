"""Evaluation harness of the reference, device-side -- SURVEY 8(f) rank 2, the step right after the
hot path: voting classification (tool/test_classification.py:114-162) and part-segmentation
accuracy / mIoU (tool/test_partseg.py:118-199).

The reference loops over clouds and parts in numpy on the host; here the per-point work (restricted
arg-max, per-part intersection / union counts, per-class hit counts) is integer tensor arithmetic on
the device the predictions live on, and only the final handful of divisions happens on the host, in
float64 like the reference's, so the reported numbers are equal to the reference's for equal inputs.

Two quirks of the reference are kept (both on by default, both switchable):
  * `PointcloudScale` is applied to a channel-first batch as `pc[i, :, 0:3] *= scales`, i.e. it
    scales the first three POINTS of every channel, not the xyz channels (test_classification.py:78,
    test_partseg.py:66), and the scaling accumulates over the votes (`points.data` is overwritten);
  * part-seg predictions are the arg-max over the category's parts WITHOUT the category's label
    offset (`# + seg_classes[cat][0]` is commented out, test_partseg.py:158), so only categories
    whose first label is 0 can score.
"""
import numpy as np
import torch

# ShapeNetPart: category -> part labels (tool/test_partseg.py:17-20), in the reference's dict order
SEG_CLASSES = {'Earphone': [16, 17, 18], 'Motorbike': [30, 31, 32, 33, 34, 35], 'Rocket': [41, 42, 43],
               'Car': [8, 9, 10, 11], 'Laptop': [28, 29], 'Cap': [6, 7], 'Skateboard': [44, 45, 46], 'Mug': [36, 37],
               'Guitar': [19, 20, 21], 'Bag': [4, 5], 'Lamp': [24, 25, 26, 27], 'Table': [47, 48, 49],
               'Airplane': [0, 1, 2, 3], 'Pistol': [38, 39, 40], 'Chair': [12, 13, 14, 15], 'Knife': [22, 23]}
SEG_LABEL_TO_CAT = {label: cat for cat, labels in SEG_CLASSES.items() for label in labels}


class PointcloudScale:
    """Random per-cloud scaling as the reference's test scripts do it (see the module docstring);
    scales come from numpy's global generator in the reference's order (one draw of 3 per cloud)."""

    def __init__(self, scale_low=2. / 3., scale_high=3. / 2., reference_quirk=True):
        self.scale_low, self.scale_high, self.reference_quirk = scale_low, scale_high, reference_quirk

    def __call__(self, pc):
        scales = np.random.uniform(low=self.scale_low, high=self.scale_high, size=(pc.size(0), 3))
        s = torch.from_numpy(scales).float().to(pc.device)
        if self.reference_quirk:
            pc[:, :, 0:3] *= s[:, None, :]          # pc[i, :, 0:3] * scales, all clouds at once
        else:
            pc[:, 0:3, :] *= s[:, :, None]          # what was meant: the xyz channels of [B, C, N]
        return pc


def to_categorical(y, num_classes):
    """1-hot encodes a tensor (tool/test_partseg.py:36-41), on y's device."""
    return torch.eye(num_classes, device=y.device)[y]


@torch.no_grad()
def vote_classification(model, points, vote_num=10, pointscale=None):
    """points [B, 3, N] channel-first (modified in place, like the reference's `points.data`);
    returns the mean of `vote_num` predictions (votes after the first see rescaled clouds)."""
    model.eval()
    pointscale = pointscale or PointcloudScale(scale_low=0.95, scale_high=1.05)
    pool = None
    for v in range(vote_num):
        if v > 0:
            points = pointscale(points)
        pred = model(points)
        pool = pred.clone() if pool is None else pool + pred
    return pool / vote_num


class ClassificationMeter:
    """Instance / class accuracy exactly as test_classification.py:141-151 accumulates them (per
    batch: per-class accuracy of the classes present, and the batch's instance accuracy)."""

    def __init__(self, num_class):
        self.class_acc = np.zeros((num_class, 3))
        self.mean_correct = []

    def update(self, pred, target):
        choice = pred.max(1)[1]
        hit = choice.eq(target.long())
        nc = self.class_acc.shape[0]
        per_class_n = torch.bincount(target.long(), minlength=nc).cpu().numpy()
        per_class_hit = torch.bincount(target.long(), weights=hit.double(), minlength=nc).cpu().numpy()
        for cat in np.nonzero(per_class_n)[0]:
            self.class_acc[cat, 0] += float(int(per_class_hit[cat])) / float(per_class_n[cat])
            self.class_acc[cat, 1] += 1
        self.mean_correct.append(int(hit.sum().item()) / float(target.shape[0]))

    def result(self):
        """-> (instance_acc, class_acc)"""
        with np.errstate(invalid="ignore", divide="ignore"):
            per = self.class_acc[:, 0] / self.class_acc[:, 1]
        return float(np.mean(self.mean_correct)), float(np.mean(per))


@torch.no_grad()
def vote_partseg(model, points, label, num_classes=16, num_votes=3, pointscale=None):
    """points [B, C, N] channel-first (modified in place), label [B] or [B,1] int64 object class;
    returns the mean of `num_votes` part-logit predictions [B, N, num_part]."""
    model.eval()
    pointscale = pointscale or PointcloudScale(scale_low=0.95, scale_high=1.05)
    pool = None
    for v in range(num_votes):
        if v > 0:
            points = pointscale(points)
        seg_pred, _ = model(points, to_categorical(label.long(), num_classes))
        pool = seg_pred.clone() if pool is None else pool + seg_pred
    return pool / num_votes


class PartSegMeter:
    """accuracy / class-average accuracy / class-average IoU / instance-average IoU of
    test_partseg.py:146-195, the per-point work done with tensor ops on the predictions' device."""

    def __init__(self, num_part=50, seg_classes=None, reference_quirk=True):
        self.num_part = num_part
        self.seg_classes = seg_classes or SEG_CLASSES
        self.label_to_cat = {l: c for c, ls in self.seg_classes.items() for l in ls}
        self.quirk = reference_quirk
        self.total_correct = 0
        self.total_seen = 0
        self.seen_class = np.zeros(num_part, dtype=np.int64)
        self.correct_class = np.zeros(num_part, dtype=np.int64)
        self.shape_ious = {cat: [] for cat in self.seg_classes}
        # first label / number of parts of the category each part label belongs to
        first = torch.zeros(num_part, dtype=torch.long)
        count = torch.ones(num_part, dtype=torch.long)
        for ls in self.seg_classes.values():
            for l in ls:
                first[l], count[l] = ls[0], len(ls)
        self._first, self._count = first, count

    def update(self, seg_pred, target):
        """seg_pred [B, N, num_part] (averaged votes), target [B, N] int64 part labels."""
        B, N, P = seg_pred.shape
        dev = seg_pred.device
        target = target.long()
        first = self._first.to(dev)[target[:, 0]]                     # [B] the cloud's category, by its first point
        count = self._count.to(dev)[target[:, 0]]
        part = torch.arange(P, device=dev)[None, :]
        inside = (part >= first[:, None]) & (part < (first + count)[:, None])       # [B,P] the category's parts
        masked = seg_pred.masked_fill(~inside[:, None, :], float("-inf"))
        pred = masked.argmax(dim=2) - first[:, None]                  # np.argmax(logits[:, parts], 1): first maximum
        if not self.quirk:
            pred = pred + first[:, None]
        self.total_correct += int((pred == target).sum().item())
        self.total_seen += B * N
        self.seen_class += torch.bincount(target.reshape(-1), minlength=P).cpu().numpy()
        self.correct_class += torch.bincount(target.reshape(-1), weights=(pred == target).reshape(-1).double(),
                                             minlength=P).long().cpu().numpy()
        # per cloud and part label: |pred == l & tgt == l| and |pred == l or tgt == l|
        oh_t = torch.zeros(B, P, dtype=torch.long, device=dev).scatter_add_(1, target, torch.ones_like(target))
        oh_p = torch.zeros(B, P, dtype=torch.long, device=dev).scatter_add_(1, pred.clamp(0, P - 1), torch.ones_like(pred))
        both = torch.zeros(B, P, dtype=torch.long, device=dev).scatter_add_(1, target, (pred == target).long())
        union = (oh_t + oh_p - both).cpu().numpy()
        inter = both.cpu().numpy()
        first_h, count_h = first.cpu().numpy(), count.cpu().numpy()
        for i in range(B):
            ls = range(int(first_h[i]), int(first_h[i] + count_h[i]))
            ious = [1.0 if union[i, l] == 0 else inter[i, l] / float(union[i, l]) for l in ls]
            self.shape_ious[self.label_to_cat[int(first_h[i])]].append(np.mean(ious))

    def result(self):
        all_ious = [iou for cat in self.shape_ious for iou in self.shape_ious[cat]]
        with np.errstate(invalid="ignore", divide="ignore"):
            cat_iou = {cat: np.mean(v) for cat, v in self.shape_ious.items()}
            class_acc = np.mean(self.correct_class / self.seen_class.astype(np.float64))
        return {"accuracy": self.total_correct / float(self.total_seen),
                "class_avg_accuracy": float(class_acc),
                "class_avg_iou": float(np.mean(list(cat_iou.values()))),
                "inctance_avg_iou": float(np.mean(all_ious)),
                "per_category_iou": {k: float(v) for k, v in cat_iou.items()}}


def load_reference_checkpoint(model, path, map_location=None):
    """`best_model.pth` of the reference's training scripts ({'model_state_dict': ...}); weights only
    (no unpickling of arbitrary objects), strict key match (the mirror keeps the reference's keys)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    state = ckpt["model_state_dict"] if "model_state_dict" in ckpt else ckpt
    model.load_state_dict(state, strict=True)
    return ckpt
