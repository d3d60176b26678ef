"""Adam on flat buckets.

`FlatAdam` takes the flat gradient buckets of a built `distributed.GradReducer`, moves the
corresponding parameters into equally laid out flat parameter buffers (`p.data` becomes a view;
values, names and the state dict are unchanged) and performs torch.optim.Adam's update as one
`mpa_adam_step_f32` launch per bucket.  Parameters that never receive a gradient are left alone,
as torch.optim.Adam leaves parameters whose `.grad` is None.

Everything that changes between steps lives on the device, so the step can be captured in a HIP
graph and replayed: the step count, and the learning rate / weight decay (`hyper`).  FlatAdam is a
`torch.optim.Optimizer` with one param group, so the schedulers the reference's training loops use
(StepLR, CosineAnnealingLR: tool/train_cls_scanobjectnn.py:219-238, tool/train_partseg.py:152-221)
attach to it unchanged; `sync_hyper()` (called by `step()` and by `runtime.GraphedTrainStep` before
every replay) copies `param_groups[0]["lr"]` / `["weight_decay"]` into the device scalars when they
changed.  `set_lr(x)` does both at once.
"""
import torch

from .ops import _launch, _p, _stream


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, reducer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if reducer.buckets is None:
            raise RuntimeError("FlatAdam needs a built GradReducer (run one backward + all_reduce first)")
        params = [p for b in reducer.buckets for p in b["params"]]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.reducer = reducer
        self.groups = []
        for b in reducer.buckets:
            flat_g = b["flat"]
            flat_p = torch.zeros_like(flat_g)
            for p, off in zip(b["params"], b["offsets"]):
                view = flat_p[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
            self.groups.append({"p": flat_p, "g": flat_g, "m": torch.zeros_like(flat_g),
                                "v": torch.zeros_like(flat_g)})
        dev = reducer.buckets[0]["flat"].device
        self.step_count = torch.zeros(1, dtype=torch.float32, device=dev)
        self.hyper = torch.tensor([lr, weight_decay], dtype=torch.float32, device=dev)     # [lr, weight_decay]
        self._hyper_host = (float(lr), float(weight_decay))

    # -- hyper-parameters that schedulers change --------------------------------------------------
    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    def set_lr(self, lr):
        self.param_groups[0]["lr"] = float(lr)
        self.sync_hyper()

    def sync_hyper(self):
        """Bring the device-side [lr, weight_decay] up to date with param_groups[0] (a stream-ordered
        copy, only when something changed; never inside a graph capture)."""
        g = self.param_groups[0]
        cur = (float(g["lr"]), float(g["weight_decay"]))
        if cur != self._hyper_host:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FlatAdam: learning rate changed during graph capture")
            self.hyper.copy_(torch.tensor(cur, dtype=torch.float32), non_blocking=False)
            self._hyper_host = cur

    @torch.no_grad()
    def step(self, closure=None):
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()
        g0 = self.param_groups[0]
        _launch("mpa_scalar_add_f32", _p(self.step_count), 1.0, _stream())
        for g in self.groups:
            _launch("mpa_adam_step_f32", _p(g["p"]), _p(g["g"]), _p(g["m"]), _p(g["v"]), g["p"].numel(),
                    float(g0["lr"]), float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"]),
                    float(g0["weight_decay"]), _p(self.step_count), _p(self.hyper), _stream())

    def zero_grad(self, set_to_none=False):
        self.reducer.zero_grad()

    def state_dict(self):
        g0 = self.param_groups[0]
        return {"step": self.step_count.clone(), "exp_avg": [g["m"].clone() for g in self.groups],
                "exp_avg_sq": [g["v"].clone() for g in self.groups], "lr": g0["lr"], "betas": g0["betas"],
                "eps": g0["eps"], "weight_decay": g0["weight_decay"]}

    def load_state_dict(self, sd):
        self.step_count.copy_(sd["step"])
        for g, m, v in zip(self.groups, sd["exp_avg"], sd["exp_avg_sq"]):
            g["m"].copy_(m)
            g["v"].copy_(v)
        g0 = self.param_groups[0]
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in sd:
                g0[k] = sd[k]
        self.sync_hyper()
