"""WarmupCosineAnnealingLR with the reference's call signature and lr values
(src/models/components/scheduler.py:46-68): linear warm-up from `warmup_lrs` to the base lr over
`warmup_epochs`, then half-cosine to `final_factor`*base over the remaining epochs.  Host-side float
math per epoch (SURVEY.md marks it out of the GPU scope); pinned by tests/golden/tables.npz."""
from __future__ import annotations

import math

from torch.optim.lr_scheduler import _LRScheduler


class WarmupCosineAnnealingLR(_LRScheduler):
    def __init__(self, optimizer, total_epoch, successor=None, final_factor=0, warmup_epochs=0, warmup_powers=1,
                 warmup_lrs=0, last_epoch=-1):
        n = len(optimizer.param_groups)

        def tup(x):
            return [x] * n if isinstance(x, (int, float)) else list(x)

        self.total_epoch, self.final_factor = total_epoch, final_factor
        self.warmup_epochs, self.warmup_powers, self.warmup_lrs = tup(warmup_epochs), tup(warmup_powers), tup(warmup_lrs)
        self.successor = successor            # accepted and ignored, exactly like the reference
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        out = []
        for i, base in enumerate(self.base_lrs):
            we = self.warmup_epochs[i]
            if self.last_epoch < we:
                f = (self.last_epoch / we) ** self.warmup_powers[i]
                out.append(f * (base - self.warmup_lrs[i]) + self.warmup_lrs[i])
            else:
                prog = min((self.last_epoch - we) / (self.total_epoch - we), 1.0)
                cosine = (math.cos(math.pi * prog) + 1) / 2
                out.append(base * (cosine * (1 - self.final_factor) + self.final_factor))
        return out
