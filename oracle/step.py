"""Oracle: one optimisation step of the reference loop main.py:154-229 on CPU (TEST
INFRASTRUCTURE, see oracle/__init__.py).  Also the ``cpu_baseline`` ("port") timed by bench.py.
"""
from collections import OrderedDict

import torch

from . import aggregation as A
from . import autojac, nets


class OracleTrainer:
    def __init__(self, cfg, seed=0, agg="sum", lr=1e-3, agg_kwargs=None):
        self.cfg = dict(cfg)
        self.arch = nets.ARCHS[cfg["arch"]]
        self.sd = nets.init_state(self.cfg, seed)
        self.params = OrderedDict((n, self.sd[n].requires_grad_(True)) for n in nets.parameter_names(self.sd))
        self.agg = agg
        self.weighting = None if agg in (None, "sum") else A.make_weighting(agg, **(agg_kwargs or {}))
        self.opt = torch.optim.Adam(list(self.params.values()), lr=lr)
        self.last = {}

    def load_state(self, sd):
        with torch.no_grad():
            for k, v in sd.items():
                if k in self.sd:
                    self.sd[k].copy_(torch.as_tensor(v))

    def forward(self, x, eps=None, train=True):
        out = self.arch["forward"](self.sd, x, self.cfg, eps, train=train)
        if self.cfg["arch"] == "betatc_vae":
            ld = self.arch["losses"](x, out, self.cfg, train=train)
        else:
            ld = self.arch["losses"](x, out, self.cfg)
        return out, ld

    def grads(self, x, eps=None):
        out, ld = self.forward(x, eps)
        if self.weighting is None:
            gs = torch.autograd.grad(ld["total_loss"], list(self.params.values()), allow_unused=True)
            grads = OrderedDict((n, torch.zeros_like(p) if g is None else g)
                                for (n, p), g in zip(self.params.items(), gs))
            info = {}
        else:
            comp = [v for k, v in ld.items() if k != "total_loss"]  # main.py:184
            feats = [out[f] for f in self.arch["features"]]
            lv = torch.stack([c.detach() for c in comp]).numpy()  # main.py:185-186 (weighted losses)
            grads, info = autojac.mtl_backward(self.params, comp, feats, self.weighting, lv)
        return out, ld, grads, info

    def step(self, x, eps=None):
        out, ld, grads, info = self.grads(x, eps)
        for n, p in self.params.items():
            p.grad = grads[n].detach().clone() if n in grads else None
        self.opt.step()
        self.last = dict(out=out, losses=ld, grads=grads, info=info)
        return {k: float(v.detach()) if hasattr(v, "detach") else float(v) for k, v in ld.items()}
