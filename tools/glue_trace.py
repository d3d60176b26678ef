"""Which host lines issue the ATen device ops of one training pass (forward AND backward)?  A TorchDispatchMode records,
for every ATen call that launches device work (copies, adds, fills, gathers, cats ...), the innermost frames of this
package on the Python stack (autograd-engine nodes of builtin ops have none: they are listed by op and shape).
    python tools/glue_trace.py [cls|seg] [f32|bf16]"""
import collections
import os
import sys
import traceback
from argparse import Namespace

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpa_amd  # noqa: E402
from mpa_amd.runtime import GraphedTrainStep  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cls"
DT = sys.argv[2] if len(sys.argv) > 2 else "f32"
mpa_amd.ops.set_feature_dtype(torch.bfloat16 if DT == "bf16" else torch.float32)
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
if which == "cls":
    from mpa_amd.models.repsurf.repsurf_ssg_umb import Model, SmoothClsLoss
    B, N = 64, 1024
    x = (torch.rand(B, 3, N, generator=g) * 2 - 1).to(dev)
    y = torch.randint(0, 40, (B,), generator=g).to(dev)
    model = Model(Namespace(num_point=N, return_dist=True, cuda_ops=True, num_class=40)).to(dev).train()
    step = GraphedTrainStep(model, SmoothClsLoss(), (x, y), lr=1e-3)
else:
    from mpa_amd.models.repsurf.pointnet2_part_seg_msg import get_loss, get_model
    B, N = 32, 2048
    x = (torch.rand(B, 3, N, generator=g) * 2 - 1).to(dev)
    label = torch.zeros(B, 1, 16)
    label[torch.arange(B), 0, torch.randint(0, 16, (B,), generator=g)] = 1
    label = label.to(dev)
    target = torch.randint(0, 50, (B, N), generator=g).to(dev)
    model = get_model(50).to(dev).train()

    def compute_loss(model, crit, x, label, target):
        pred, _ = model(x, label)
        return crit(pred.reshape(-1, 50), target.reshape(-1))
    step = GraphedTrainStep(model, get_loss(), (x, label, target), lr=1e-3, compute_loss=compute_loss)

VIEWS = ("view", "reshape", "permute", "transpose", "slice", "select", "expand", "as_strided", "alias", "detach", "unsqueeze",
         "squeeze", "t.", "empty", "_unsafe_view", "split", "unbind", "narrow", "is_", "size", "stride", "lift", "_local_scalar",
         "record_stream", "chunk", "resize_")
agg = collections.Counter()


class Trace(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not any(v in name for v in VIEWS):
            t = next((a for a in args if isinstance(a, torch.Tensor)), None)
            if t is None or t.is_cuda:
                frames = [f for f in traceback.extract_stack() if "amd/" in f.filename and "runtime.py" not in f.filename]
                where = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in frames[-3:][::-1]) or "(autograd engine)"
                shape = "%s %s" % (tuple(t.shape), str(t.dtype).replace("torch.", "")) if t is not None else ""
                agg[(name, where, shape)] += 1
        return out


torch.cuda.synchronize()
with Trace():
    step._fwd_bwd()
torch.cuda.synchronize()
for (name, where, shape), n in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print("%3d  %-34s %-34s %s" % (n, name, shape, where))
