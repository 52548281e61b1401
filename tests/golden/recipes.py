"""Seeded input recipes shared by tests/golden/make_golden.py (which feeds them to the REFERENCE)
and by the parity tests (which feed them to the oracle and to the HIP path).  Pure torch; no
reference import.  Large inputs are regenerated from these recipes instead of being stored."""
import torch


def e2e_inputs(seed: int, D: int):
    g = torch.Generator().manual_seed(seed + 1)
    nc = torch.randn(D, generator=g) * 0.1
    test_feats = torch.randn(1, 1, 512 * 2, D, generator=g) * 0.3
    g2 = torch.Generator().manual_seed(seed + 2)
    frames = torch.randn(1, 512, 3, 32, 32, generator=g2)
    g3 = torch.Generator().manual_seed(seed + 3)
    labels = torch.tensor([2, 11, 7, 7])
    train_feats = torch.randn(4, 1, 512, D, generator=g3) * 0.3
    g4 = torch.Generator().manual_seed(seed + 4)
    mask = torch.bernoulli(torch.ones(4, 32) * 0.3, generator=g4)
    # guarantee >= 3 surviving segments per video so that top-k is tie-free
    for r in range(4):
        if mask[r].sum() < 3:
            mask[r, :3] = 1
    return dict(nc=nc, test_feats=test_feats, frames=frames, labels=labels,
                train_feats=train_feats, mask=mask)


def vit_frames(seed: int, n: int, res: int):
    g = torch.Generator().manual_seed(seed + 100)
    return torch.randn(n, 3, res, res, generator=g)


def selector_inputs(seed: int = 11):
    g = torch.Generator().manual_seed(seed)
    B, N, L, D, C = 4, 32, 16, 64, 14
    x = torch.randn(B, N * L, D, generator=g) * 0.3 + 0.1
    tf = torch.randn(C, D, generator=g) * 0.3 + 0.1
    nc = x.reshape(-1, D)[B // 2 * N * L:].mean(0)
    labels = torch.tensor([3, 9, 7, 7])
    rm0 = torch.randn(C - 1, generator=g) * 0.05
    rv0 = torch.rand(C - 1, generator=g) * 0.05 + 0.05
    m1 = torch.bernoulli(torch.ones(B, N) * 0.3, generator=g)
    m2 = torch.bernoulli(torch.ones(B, N) * 0.3, generator=g)
    m1[:, :3] = 1      # >= 3 surviving segments per video: top-k is tie-free
    m2[:, 4:7] = 1
    return dict(x=x, tf=tf, nc=nc, labels=labels, rm0=rm0, rv0=rv0, topk_mask=m1, bottomk_mask=m2)
