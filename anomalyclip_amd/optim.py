"""AcxAdamW: torch.optim.AdamW semantics (decoupled weight decay, bias correction, no amsgrad) with the
update executed by the fused libacx kernel acx_adamw -- one HBM pass over (p, g, m, v) per parameter
(16 B read + 12 B written per value).  Param groups / lr scheduling behave like any torch optimizer, so the
reference's configure_optimizers grouping (anomaly_clip_module.py:693-746) carries over unchanged."""
from __future__ import annotations

import torch

from . import ops
from .parallel import _is_dense as parallel_is_dense, same_layout


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
        # every tensor that shares (betas, eps, step) -- in practice all of them -- goes into ONE acx_adamw_multi launch
        # (per-tensor lr / weight decay carry the reference's four param groups); 34 launches -> 1 at the UCF config
        batches: dict = {}
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    # moments in the PARAMETER's memory layout: the kernel walks the raw memory of (p, g, m, v) elementwise
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                if not (p.is_contiguous() or parallel_is_dense(p)) or not same_layout(st["exp_avg"], p):
                    raise ops.L.AcxError("AcxAdamW updates parameters in place through raw pointers: dense parameters only "
                                         "(row-major, or a permuted-contiguous layout such as channels-last)")
                g = p.grad
                if not same_layout(g, p):             # same physical element order as the parameter
                    g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
                key = (p.device, float(b1), float(b2), float(group["eps"]), int(st["step"]))
                b = batches.setdefault(key, ([], [], [], [], [], []))
                for lst, val in zip(b, (p, g, st["exp_avg"], st["exp_avg_sq"], float(group["lr"]), float(group["weight_decay"]))):
                    lst.append(val)
        for (_, b1, b2, eps, step), (ps, gs, ms, vs, lrs, wds) in batches.items():
            ops.adamw_multi_(ps, gs, ms, vs, lrs, wds, b1, b2, eps, step)
        ops.WEIGHT_EPOCH[0] += 1
        return loss
