"""GPU parity of the metrics epilogue (section 8f rank 3): libacx sort / curve / counting kernels vs
oracle/metrics_oracle.py (itself pinned to scikit-learn in tests/test_oracle_metrics.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from anomalyclip_amd import metrics as M
from anomalyclip_amd import ops
from oracle import metrics_oracle as MO

DEV = "cuda:0"


@pytest.mark.parametrize("n", [1, 63, 2048, 2049, 100_003, 1_500_000])
@pytest.mark.parametrize("desc", [True, False])
def test_sort_pairs_is_a_stable_sort(n, desc):
    g = torch.Generator().manual_seed(n)
    k = torch.randn(n, generator=g)
    if n > 100:
        k[: n // 2] = torch.round(k[: n // 2] * 50) / 50                   # plant ties
        k[5] = 0.0
        k[6] = -0.0
        k[7] = float("inf")
        k[8] = -float("inf")
    v = torch.arange(n, dtype=torch.int32)
    ks, vs = ops.sort_pairs(k.to(DEV), v.to(DEV), descending=desc)
    ks, vs = ks.cpu(), vs.cpu()
    # stable reference: sort by (key, original index)
    kn = k.numpy().astype(np.float64)
    kn = np.where((kn == 0) & np.signbit(k.numpy()), -1e-300, kn)          # -0.0 orders below +0.0
    order = np.argsort(-kn if desc else kn, kind="stable")
    assert np.array_equal(vs.numpy(), order.astype(np.int32))
    assert np.array_equal(ks.numpy().view(np.uint32), k.numpy()[order].view(np.uint32))


@pytest.mark.parametrize("B,n,shared", [(15, 5000, True), (3, 2049, False), (1, 1, True), (4, 70000, True)])
def test_sort_pairs_batched_equals_separate_sorts(B, n, shared):
    """acx_sort_pairs_batched: B stable sorts in one launch sequence == B calls of acx_sort_pairs, bit for bit, with strided key
    rows and a payload array that is either shared by all problems (the metrics epilogue) or one per problem."""
    g = torch.Generator().manual_seed(B * 131 + n)
    big = torch.randn(B, n + 7, generator=g)
    big[:, : n // 2] = torch.round(big[:, : n // 2] * 20) / 20                # ties
    keys = big.to(DEV)[:, :n]                                                  # row stride n + 7
    vals = torch.randint(0, 1 << 20, (n,) if shared else (B, n), generator=g, dtype=torch.int32).to(DEV)
    ks, vs = ops.sort_pairs_batched(keys, vals, descending=True)
    for b in range(B):
        k1, v1 = ops.sort_pairs(keys[b].contiguous(), vals if shared else vals[b].contiguous(), descending=True)
        assert torch.equal(ks[b].view(torch.int32), k1.view(torch.int32)) and torch.equal(vs[b], v1)


def test_clf_curve_batched_equals_separate_curves():
    """acx_clf_curve_batched: the records of B curves computed in one launch sequence are byte-identical to B acx_clf_curve calls
    (exact int64 AUROC / threshold arithmetic, fixed-order f64 AP), and the curve arrays are those of problem 0."""
    n, B = 30000, 6
    g = torch.Generator().manual_seed(5)
    keys = (torch.round(torch.rand(B, n, generator=g) * 500) / 500).to(DEV)
    lab = torch.randint(0, 5, (n,), generator=g, dtype=torch.int32).to(DEV)
    ks, vs = ops.sort_pairs_batched(keys, lab, descending=True)
    cls, neg = [2, 0, 1, 2, 3, 4], [True, False, False, False, False, False]
    res = torch.zeros(B * ops.CURVE_RESULT_BYTES, dtype=torch.uint8, device=DEV)
    tps, fps, thr = ops.clf_curve_batched(ks, vs, cls, neg, res, curves=True)
    for b in range(B):
        one = torch.zeros(ops.CURVE_RESULT_BYTES, dtype=torch.uint8, device=DEV)
        t1, f1, h1 = ops.clf_curve(ks[b], vs[b], cls[b], neg[b], one, curves=b == 0)
        assert torch.equal(one, res[b * ops.CURVE_RESULT_BYTES:(b + 1) * ops.CURVE_RESULT_BYTES])
        if b == 0:
            nd = int(one[32:40].view(torch.int64).item())                      # n_distinct
            assert torch.equal(tps[:nd], t1[:nd]) and torch.equal(fps[:nd], f1[:nd]) and torch.equal(thr[:nd], h1[:nd])


def _case(n, seed, ties, C=14, nid=7):
    rng = np.random.default_rng(seed)
    labels = rng.integers(0, C, n)
    labels[rng.random(n) < 0.55] = nid
    s = np.clip(0.45 * (labels != nid) + 0.6 * rng.random(n), 0, 1).astype(np.float32)
    if ties:
        s = (np.round(s * ties) / ties).astype(np.float32)
    p = rng.random((n, C - 1)).astype(np.float32) ** 3
    hit = labels != nid
    col = np.where(labels > nid, labels - 1, labels)
    p[hit, col[hit]] += 0.8 * rng.random(hit.sum()).astype(np.float32)
    p = (p / p.sum(1, keepdims=True) * s[:, None]).astype(np.float32)
    return s, labels.astype(np.int64), p


@pytest.mark.parametrize("n,ties", [(5000, 0), (5000, 20), (300_000, 0), (300_000, 1000), (17, 0)])
def test_epilogue_matches_oracle(n, ties):
    C, nid = 14, 7
    s, labels, p = _case(n, n + ties, ties, C, nid)
    want = MO.epilogue(s, labels, p, nid, C)
    got = M.evaluate(torch.from_numpy(s).to(DEV), torch.from_numpy(labels).to(DEV), torch.from_numpy(p).to(DEV), nid, C,
                     curves=True)
    for k in ("auc_roc", "auc_pr", "mean_mc_auroc", "mean_mc_aupr"):
        assert got[k] == pytest.approx(want[k], rel=1e-12, abs=1e-14), k
    assert got["optimal_threshold"] == want["optimal_threshold"]                  # exact: same f32 score picked
    assert np.allclose(got["mc_auroc"], want["mc_auroc"], rtol=1e-12, atol=1e-14)
    assert np.allclose(got["mc_aupr"], want["mc_aupr"], rtol=1e-12, atol=1e-14, equal_nan=True)
    # integer work: bit-exact
    assert np.array_equal(got["y_pred"].cpu().numpy(), want["y_pred"])
    assert np.array_equal(got["confusion_counts"], want["confusion_counts"])
    assert np.allclose(got["top1_accuracy"], want["top1_accuracy"], rtol=0, atol=0, equal_nan=True)
    assert np.allclose(got["top5_accuracy"], want["top5_accuracy"], rtol=0, atol=0, equal_nan=True)
    for t, v in want["f1_scores"].items():
        assert got["f1_scores"][t] == pytest.approx(v, rel=1e-15)
    fpr, tpr, thr = (t.cpu().numpy() for t in got["roc"])
    f2, t2, th2 = MO.binary_roc(s, (labels != nid).astype(np.int64))
    assert np.allclose(fpr, f2, atol=1e-6) and np.allclose(tpr, t2, atol=1e-6) and np.array_equal(thr, th2.astype(np.float32))


def test_absent_classes_and_determinism():
    C, nid = 14, 7
    s, labels, p = _case(40_000, 3, 50, C, nid)
    labels[labels == 2] = nid                                   # class 2 never occurs
    labels[labels == 11] = nid
    want = MO.epilogue(s, labels, p, nid, C)
    args = (torch.from_numpy(s).to(DEV), torch.from_numpy(labels).to(DEV), torch.from_numpy(p).to(DEV), nid, C)
    a, b = M.evaluate(*args), M.evaluate(*args)
    assert a["mc_auroc"][2] == 0.0 and np.isnan(a["mc_aupr"][2]) and np.isnan(a["top1_accuracy"][2])
    assert a["mean_mc_auroc"] == pytest.approx(want["mean_mc_auroc"], rel=1e-12)
    assert a["mean_mc_aupr"] == pytest.approx(want["mean_mc_aupr"], rel=1e-12)
    for k in ("auc_roc", "auc_pr", "mean_mc_auroc", "mean_mc_aupr", "optimal_threshold"):
        assert a[k] == b[k]                                     # run-to-run bit-identical


def test_anticorrelated_scores_pick_the_origin_point():
    n = 5000
    rng = np.random.default_rng(0)
    labels = np.where(rng.random(n) < 0.5, 7, 3).astype(np.int64)
    s = np.where(labels == 7, 0.9, 0.1).astype(np.float32)      # normal frames score HIGH
    p = np.full((n, 13), 1 / 13, np.float32) * s[:, None]
    got = M.evaluate(torch.from_numpy(s).to(DEV), torch.from_numpy(labels).to(DEV), torch.from_numpy(p).to(DEV), 7, 14)
    assert got["auc_roc"] == 0.0 and got["optimal_threshold"] == 1.0
    assert (got["y_pred"].cpu().numpy() == 7).all()


def test_sklearn_agrees_at_scale():
    sk = pytest.importorskip("sklearn.metrics")
    s, labels, p = _case(1_000_000, 9, 0)
    got = M.evaluate(torch.from_numpy(s).to(DEV), torch.from_numpy(labels).to(DEV), torch.from_numpy(p).to(DEV), 7, 14,
                     per_frame=False)
    t = labels != 7
    assert got["auc_roc"] == pytest.approx(sk.roc_auc_score(t, s), rel=1e-10)
    assert got["auc_pr"] == pytest.approx(sk.average_precision_score(t, s), rel=1e-10)
