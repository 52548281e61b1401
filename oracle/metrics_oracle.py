"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the metrics epilogue.

Restates, in numpy, (1) the reference's own epilogue arithmetic (src/models/anomaly_clip_module.py:501-626:
normal-probability insertion, optimal threshold, y_pred / top-1 / top-5 / confusion / F1 bookkeeping) and (2) the
curve functions of its THIRD-PARTY dependency torchmetrics==0.11.0 (requirements.txt:6), which is not vendored
under /root/reference and not installed in this image.  The torchmetrics part follows the published 0.11.0
algorithm (functional/classification/precision_recall_curve.py::_binary_clf_curve, roc.py::_binary_roc_compute,
auroc.py::_binary_auroc_compute -> utilities/compute.py::_auc_compute_without_check (trapezoid),
average_precision.py::_binary_average_precision_compute, and their one-vs-rest multiclass counterparts):

    sort by score descending; keep the LAST index of every run of equal scores ("distinct thresholds");
    tps = cumsum(target)[idx], fps = 1 + idx - tps;
    ROC   : prepend (0, 0) with threshold 1.0; fpr = fps/fps[-1], tpr = tps/tps[-1]; zero curve when a class is absent;
    AUROC : trapz(tpr, fpr);
    AP    : sum_k (recall_k - recall_{k-1}) * precision_k  (nan when there is no positive).

PARITY PINNING: torchmetrics cannot be executed here, so this file is pinned instead against scikit-learn
(roc_auc_score / average_precision_score / roc_curve implement the same definitions) in
tests/test_oracle_metrics.py, and against brute-force O(n^2) pair counting on small cases.  Values are computed in
float64 / exact integers; torchmetrics accumulates in float32, so agreement with a real run is to ~1e-6 relative.
Only tests/ (and smoke/bench checkers) may import this file.
"""
from __future__ import annotations

import numpy as np


def binary_clf_curve(scores: np.ndarray, target: np.ndarray):
    """-> fps, tps (int64), thresholds, and the positions (in the descending-sorted order) of the kept points."""
    order = np.argsort(-scores.astype(np.float64), kind="stable")
    s = scores[order]
    t = target[order].astype(np.int64)
    distinct = np.nonzero(s[1:] != s[:-1])[0]
    idx = np.concatenate([distinct, [len(s) - 1]]).astype(np.int64)
    tps = np.cumsum(t)[idx]
    fps = 1 + idx - tps
    return fps, tps, s[idx], idx


def binary_roc(scores, target):
    fps, tps, thr, _ = binary_clf_curve(scores, target)
    tps = np.concatenate([[0], tps]).astype(np.float64)
    fps = np.concatenate([[0], fps]).astype(np.float64)
    thr = np.concatenate([[1.0], thr.astype(np.float64)])
    fpr = fps / fps[-1] if fps[-1] > 0 else np.zeros_like(fps)
    tpr = tps / tps[-1] if tps[-1] > 0 else np.zeros_like(tps)
    return fpr, tpr, thr


def binary_auroc(scores, target) -> float:
    fpr, tpr, _ = binary_roc(scores, target)
    trap = getattr(np, "trapezoid", None) or np.trapz
    return float(trap(tpr, fpr))


def binary_average_precision(scores, target) -> float:
    fps, tps, _, _ = binary_clf_curve(scores, target)
    if tps[-1] == 0:
        return float("nan")
    precision = tps / (tps + fps)
    recall = tps / tps[-1]
    prev = np.concatenate([[0.0], recall[:-1]])
    return float(np.sum((recall - prev) * precision))


def optimal_threshold(scores, target):
    """anomaly_clip_module.py:526-527: thresholds[argmax(tpr - fpr)] (first maximum; index 0 is the added (0,0)
    point with threshold 1.0).  The argmax is evaluated exactly (integers tps*N - fps*P) rather than on rounded
    float32 rates."""
    fps, tps, thr, _ = binary_clf_curve(scores, target)
    P, N = int(tps[-1]), int(fps[-1])
    j = np.concatenate([[0], tps * N - fps * P])
    k = int(np.argmax(j))
    return (1.0 if k == 0 else float(thr[k - 1])), k


def epilogue(abnormal_scores: np.ndarray, labels: np.ndarray, class_probs: np.ndarray, normal_idx: int,
             num_classes: int) -> dict:
    """anomaly_clip_module.py:501-626 (test_epoch_end) without the plots; `class_probs` is [n, C-1]."""
    n = len(abnormal_scores)
    s = abnormal_scores.astype(np.float32)
    full = np.concatenate([class_probs[:, :normal_idx], (1 - s)[:, None], class_probs[:, normal_idx:]], 1)   # :510-518
    lb = (labels != normal_idx).astype(np.int64)                                                               # :520
    out = {"auc_roc": binary_auroc(s, lb), "auc_pr": binary_average_precision(s, lb)}
    thr, _ = optimal_threshold(s, lb)
    out["optimal_threshold"] = thr
    wo = class_probs                                                                                           # :532-535
    am = np.argmax(wo, 1)
    am = np.where(am >= normal_idx, am + 1, am)
    y_pred = np.where(s < np.float32(thr), normal_idx, am)                                                     # :538-547
    # rank of the true class among the non-normal probabilities (ties -> lower index first, like a stable topk)
    top1 = np.full(num_classes, np.nan)
    top5 = np.full(num_classes, np.nan)
    order = np.argsort(-wo.astype(np.float64), axis=1, kind="stable")[:, :5]
    order = np.where(order >= normal_idx, order + 1, order)                                                    # :556-557
    t5 = np.where((y_pred == normal_idx)[:, None], np.concatenate([np.full((n, 1), normal_idx), order[:, :4]], 1),
                  order)                                                                                       # :559-572
    for c in range(num_classes):                                                                               # :574-581
        m = labels == c
        if m.any():
            top1[c] = np.mean(y_pred[m] == c)
            top5[c] = np.mean((t5[m] == c).any(1))
    out["top1_accuracy"], out["top5_accuracy"], out["y_pred"] = top1, top5, y_pred
    mc_auroc = np.zeros(num_classes)
    mc_aupr = np.zeros(num_classes)
    for c in range(num_classes):                                                                               # :583-584
        t = (labels == c).astype(np.int64)
        mc_auroc[c] = binary_auroc(full[:, c], t)
        mc_aupr[c] = binary_average_precision(full[:, c], t)
    out["mc_auroc"], out["mc_aupr"] = mc_auroc, mc_aupr

    def _mean_wo_normal(v):                                                                                    # :586-592
        w = np.concatenate([v[:normal_idx], v[normal_idx + 1:]]).copy()
        w[w == 0] = np.nan
        return float(np.nanmean(w)) if np.isfinite(w).any() else float("nan")
    out["mean_mc_auroc"], out["mean_mc_aupr"] = _mean_wo_normal(mc_auroc), _mean_wo_normal(mc_aupr)
    f1 = {}
    for i in range(10):                                                                                        # :621-626
        th = (i + 1) / 10
        pb = np.where(s < th, 0, 1)
        tp = int(np.sum((pb == 1) & (lb == 1)))
        fp = int(np.sum((pb == 1) & (lb == 0)))
        fn = int(np.sum((pb == 0) & (lb == 1)))
        f1[th] = (2 * tp / (2 * tp + fp + fn)) if (2 * tp + fp + fn) else 0.0
    out["f1_scores"] = f1
    cm = np.zeros((num_classes, num_classes), np.int64)                                                        # :673
    np.add.at(cm, (labels, y_pred), 1)
    out["confusion_counts"] = cm
    rs = cm.sum(1, keepdims=True).astype(np.float64)
    out["confusion_matrix"] = np.divide(cm, rs, out=np.zeros_like(rs.repeat(num_classes, 1)), where=rs > 0)
    return out
