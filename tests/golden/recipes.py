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


CONFIG0_LENGTHS = (300, 512, 513, 1000, 1025, 2000, 3000, 5000)


def config0_feature_files(directory, lengths=CONFIG0_LENGTHS, D: int = 512, seed: int = 0):
    """BASELINE.json configs[0] / SURVEY.md 8(d) Config 1: ShanghaiTech-shaped evaluation from pre-extracted features --
    eight `.npy` files (T, 512) float32 ~ N(0, 0.05^2) + 0.1, seed 0.  Returns (paths, arrays, per-frame class labels,
    ncentroid = mean over every frame)."""
    import os
    import numpy as np
    rng = np.random.default_rng(seed)
    paths, arrays, labels = [], [], []
    for i, T in enumerate(lengths):
        a = (rng.standard_normal((T, D)) * 0.05 + 0.1).astype(np.float32)
        p = os.path.join(str(directory), f"sht_{i:02d}_{T}.npy")
        np.save(p, a)
        paths.append(p)
        arrays.append(a)
        lab = np.full(T, 8, dtype=np.int64)         # per-frame labels as the reference's dataset builds them: normal_id 8 ...
        if i % 2:                                   # ... and, in odd videos, one anomalous interval of class i (1, 3, 5, 7)
            lab[T // 3: T // 2] = i
        labels.append(lab)
    nc = torch.from_numpy(np.concatenate(arrays).astype(np.float64).mean(0).astype(np.float32))
    return paths, arrays, labels, nc


def config0_annotation_files(directory, paths, labels):
    """The reference's list files for the videos above (feature_dataset.py:226-241): annotation rows
    `path start end label`, temporal-annotation rows `name label s1 e1`.  -> (annotation file, temporal file)."""
    import os
    ann, tmp = os.path.join(str(directory), "anno.txt"), os.path.join(str(directory), "temporal.txt")
    with open(ann, "w") as fa, open(tmp, "w") as ft:
        for p, lab in zip(paths, labels):
            name = os.path.splitext(os.path.basename(p))[0]
            cls = int(lab.min())                    # 8 (normal video) or the anomalous class (< 8)
            fa.write(f"{name} 0 {len(lab) - 1} {cls}\n")
            nz = (lab != 8).nonzero()[0]
            ft.write(f"{name}.mp4 {cls} {int(nz[0])} {int(nz[-1])}\n" if len(nz) else f"{name}.mp4 {cls} -1 -1\n")
    return ann, tmp


def elem_excess(a, b, rtol: float = 1e-3, afrac: float = 1e-5) -> float:
    """ELEMENT-wise form of north_star's "fp logits within 1e-3 rel": max over elements of
    |a - b| / (rtol * |b| + afrac * max|b|); the comparison passes when the result is <= 1.  (The norm-wise `relerr` of the
    GPU tests lets an element 100x below the tensor's maximum be off by 100x more; this bound does not.)"""
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).abs() / (rtol * b.abs() + afrac * b.abs().max() + 1e-300)).max().item()
