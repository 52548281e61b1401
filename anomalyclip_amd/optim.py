"""AcxAdamW: torch.optim.AdamW semantics (decoupled weight decay, bias correction, no amsgrad) with the
update executed by the fused libacx kernel acx_adamw -- one HBM pass over (p, g, m, v) per parameter
(16 B read + 12 B written per value).  Param groups / lr scheduling behave like any torch optimizer, so the
reference's configure_optimizers grouping (anomaly_clip_module.py:693-746) carries over unchanged."""
from __future__ import annotations

import torch

from . import ops


class AcxAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                ops.adamw_(p, g, st["exp_avg"], st["exp_avg_sq"], float(group["lr"]), b1, b2, group["eps"],
                           group["weight_decay"], st["step"])
        ops.WEIGHT_EPOCH[0] += 1
        return loss
