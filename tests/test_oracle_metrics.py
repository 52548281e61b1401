"""Pins oracle/metrics_oracle.py (restated torchmetrics 0.11.0 curves) against scikit-learn and brute force."""
import numpy as np
import pytest

from oracle import metrics_oracle as MO

sk = pytest.importorskip("sklearn.metrics")


def _case(n, seed, ties):
    rng = np.random.default_rng(seed)
    s = rng.random(n).astype(np.float32)
    if ties:
        s = np.round(s * ties) / np.float32(ties)
    t = (rng.random(n) < 0.3 + 0.4 * s).astype(np.int64)
    return s.astype(np.float32), t


@pytest.mark.parametrize("n,ties", [(50, 0), (1000, 0), (1000, 16), (20000, 255), (7, 2)])
def test_auroc_ap_match_sklearn(n, ties):
    s, t = _case(n, n + ties, ties)
    assert abs(MO.binary_auroc(s, t) - sk.roc_auc_score(t, s)) < 1e-12
    assert abs(MO.binary_average_precision(s, t) - sk.average_precision_score(t, s)) < 1e-12
    fpr, tpr, thr = MO.binary_roc(s, t)
    f2, t2, th2 = sk.roc_curve(t, s, drop_intermediate=False)
    assert np.allclose(fpr, f2) and np.allclose(tpr, t2) and np.allclose(thr[1:], th2[1:])


def test_auroc_is_pair_counting():
    s, t = _case(300, 5, 8)
    pos, neg = s[t == 1], s[t == 0]
    u = (pos[:, None] > neg[None, :]).sum() + 0.5 * (pos[:, None] == neg[None, :]).sum()
    assert abs(MO.binary_auroc(s, t) - u / (len(pos) * len(neg))) < 1e-12


def test_degenerate_classes():
    s = np.linspace(0, 1, 10, dtype=np.float32)
    assert MO.binary_auroc(s, np.zeros(10, np.int64)) == 0.0          # torchmetrics: zero curve + warning
    assert np.isnan(MO.binary_average_precision(s, np.zeros(10, np.int64)))
    assert MO.binary_auroc(s, np.ones(10, np.int64)) == 0.0
    thr, k = MO.optimal_threshold(s, (s < 0.5).astype(np.int64))     # anti-correlated: the (0,0) point wins
    assert k == 0 and thr == 1.0


def test_epilogue_shapes_and_consistency():
    rng = np.random.default_rng(1)
    n, C, nid = 4000, 14, 7
    labels = rng.integers(0, C, n)
    labels[rng.random(n) < 0.5] = nid
    s = np.clip(0.6 * (labels != nid) + 0.5 * rng.random(n), 0, 1).astype(np.float32)
    p = rng.random((n, C - 1)).astype(np.float32)
    p = (p / p.sum(1, keepdims=True) * s[:, None]).astype(np.float32)
    r = MO.epilogue(s, labels, p, nid, C)
    assert 0.5 < r["auc_roc"] <= 1 and 0 < r["auc_pr"] <= 1
    assert r["confusion_counts"].sum() == n and r["top5_accuracy"][nid] >= r["top1_accuracy"][nid]
    full = np.concatenate([p[:, :nid], (1 - s)[:, None], p[:, nid:]], 1)
    for c in (0, nid, C - 1):
        assert abs(r["mc_auroc"][c] - sk.roc_auc_score(labels == c, full[:, c])) < 1e-12
