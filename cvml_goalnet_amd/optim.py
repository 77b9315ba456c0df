"""`Adam` — `torch.optim.Adam(model.parameters(), lr=...)` (/root/reference/main.py:70, 187, 193) as ONE fused pass over the AVM's flat
parameter arena instead of torch's per-tensor passes over 30 strided views.

    optimizer = cvml_goalnet_amd.optim.Adam(model.parameters(), lr=0.001)      # was: optim.Adam(params = model.parameters(), lr = lr)
    optimizer.zero_grad(); loss.backward(); optimizer.step()                   # main.py:187-193, unchanged

Same defaults and the same arithmetic as torch's single-tensor Adam (betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad,
bias-corrected; `goalnet_adam_step_dev`, csrc/small.hip). The parameters must be those of ONE `AVM` (they are views of its arena, and
autograd leaves their gradients in its gradient arena: `_AVMFunction.backward`); like the stock optimizer it may be constructed before
the first forward (Lazy parameters). `zero_grad()` only drops the `.grad` references: the gradient arena is overwritten by every
backward. State (`exp_avg`, `exp_avg_sq`, step) lives in the model (`AVM._adam_m`, `_adam_v`, the device step counter).
"""
from __future__ import annotations

import torch

from .avm import AVM


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, model: AVM = None):
        params = list(params)
        if not 0.0 <= lr or not 0.0 <= eps or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._model = model

    def _find_model(self):
        if self._model is None:
            raise RuntimeError("cvml_goalnet_amd.optim.Adam: pass model=<the AVM> (or use AVM.make_optimizer()) — the parameters of one AVM "
                               "live in one flat arena and are stepped together")
        return self._model

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        m = self._find_model()
        if m._garena is None:
            raise RuntimeError("step() before any backward: there are no gradients")
        if len(self.param_groups) != 1:
            raise RuntimeError("one parameter group: the arena is updated in one pass")
        g = self.param_groups[0]
        want = {id(getattr(*m._module_of(s.name))) for s in m._specs}
        if {id(p) for p in g["params"]} != want:
            raise RuntimeError("cvml_goalnet_amd.optim.Adam steps ALL parameters of its AVM (model.parameters())")
        m.adam_step(g["lr"], tuple(g["betas"]), g["eps"])
        return loss

    def zero_grad(self, set_to_none: bool = True):
        for g in self.param_groups:
            for p in g["params"]:
                p.grad = None
