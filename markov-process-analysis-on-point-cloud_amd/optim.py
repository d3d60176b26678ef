"""Adam on flat buckets.

`FlatAdam` takes the flat gradient buckets of a built `distributed.GradReducer`, moves the
corresponding parameters into equally laid out flat parameter buffers (`p.data` becomes a view;
values, names and the state dict are unchanged) and performs torch.optim.Adam's update as one
`mpa_adam_step_f32` launch per bucket.  Parameters that never receive a gradient are left alone,
as torch.optim.Adam leaves parameters whose `.grad` is None.  The step count is a device scalar,
so the step can be captured in a HIP graph.
"""
import torch

from .ops import _launch, _p, _stream


class FlatAdam:
    def __init__(self, reducer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if reducer.buckets is None:
            raise RuntimeError("FlatAdam needs a built GradReducer (run one backward + all_reduce first)")
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
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
        self.step_count = torch.zeros(1, dtype=torch.float32, device=reducer.buckets[0]["flat"].device)

    def step(self):
        _launch("mpa_scalar_add_f32", _p(self.step_count), 1.0, _stream())
        for g in self.groups:
            _launch("mpa_adam_step_f32", _p(g["p"]), _p(g["g"]), _p(g["m"]), _p(g["v"]), g["p"].numel(),
                    float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                    float(self.weight_decay), _p(self.step_count), _stream())

    def zero_grad(self, set_to_none=False):
        self.reducer.zero_grad()

    def state_dict(self):
        return {"step": self.step_count.clone(), "exp_avg": [g["m"].clone() for g in self.groups],
                "exp_avg_sq": [g["v"].clone() for g in self.groups], "lr": self.lr, "betas": self.betas,
                "eps": self.eps, "weight_decay": self.weight_decay}

    def load_state_dict(self, sd):
        self.step_count.copy_(sd["step"])
        for g, m, v in zip(self.groups, sd["exp_avg"], sd["exp_avg_sq"]):
            g["m"].copy_(m)
            g["v"].copy_(v)
