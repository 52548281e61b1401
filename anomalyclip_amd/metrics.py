"""Metrics epilogue on the GPU: the numbers `test_epoch_end` /
`on_validation_epoch_end` write to metrics.json (reference anomaly_clip_module.py:339-404, 501-626), computed
over all frames by libacx's sort / scan / count kernels instead of torchmetrics (==0.11.0 in the reference,
not installed here).  Plots are out of scope.

Per curve (1 binary + C one-vs-rest): one stable radix sort of (score, label) pairs and one scan pass; the
AUROC and the Youden-optimal threshold are exact integer arithmetic, AP a fixed-order f64 sum, so results are
run-to-run deterministic and agree with torchmetrics' float32 accumulation to ~1e-6."""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict

import numpy as np
import torch

from . import _lib as L
from . import ops


def _records(buf: torch.Tensor, k: int):
    raw = buf.cpu().numpy().tobytes()
    return [L.CurveResult.from_buffer_copy(raw[i * ops.CURVE_RESULT_BYTES:(i + 1) * ops.CURVE_RESULT_BYTES])
            for i in range(k)]


def _mean_without_normal(v: np.ndarray, normal_idx: int) -> float:
    w = np.concatenate([v[:normal_idx], v[normal_idx + 1:]]).astype(np.float64)      # :586-592
    w[w == 0] = np.nan
    return float(np.nanmean(w)) if np.isfinite(w).any() else float("nan")


@torch.no_grad()
def evaluate(abnormal_scores: torch.Tensor, labels: torch.Tensor, class_probs: torch.Tensor, normal_idx: int,
             num_classes: int, per_frame: bool = True, curves: bool = False) -> Dict[str, object]:
    """abnormal_scores f32 [n], labels int [n], class_probs f32 [n, C-1] (softmax * score, anomaly_clip_module.py:474-477),
    all on the GPU.  Returns the reference's metrics.json keys (+ y_pred, f1_scores, confusion_matrix when
    `per_frame`; + the ROC / PR curve points when `curves`)."""
    dev = abnormal_scores.device
    n, Cn = abnormal_scores.numel(), num_classes
    s = abnormal_scores.contiguous().float()
    lab64 = labels.to(torch.int64).contiguous()
    lab32 = lab64.to(torch.int32)
    probs = class_probs.contiguous().float()
    assert probs.shape == (n, Cn - 1)
    res = torch.zeros((Cn + 1) * ops.CURVE_RESULT_BYTES, dtype=torch.uint8, device=dev)
    rec = lambda i: res[i * ops.CURVE_RESULT_BYTES:(i + 1) * ops.CURVE_RESULT_BYTES]

    # All Cn + 1 score columns in one [Cn + 1, n] array: row 0 = the anomaly score (binary curve, target = labels != normal,
    # :520-530), rows 1.. = class_probs with the normal column (1 - score) inserted (:507-518, :583-584); they are sorted by
    # ONE batched launch sequence (the label vector is the shared payload) instead of Cn + 1 sorts of 12 launches each.
    allc = torch.empty(Cn + 1, n, dtype=torch.float32, device=dev)
    allc[0].copy_(s)
    cols = ops.transpose(probs)                                   # [C-1, n], one contiguous score column per class
    allc[1:1 + normal_idx].copy_(cols[:normal_idx])
    allc[2 + normal_idx:].copy_(cols[normal_idx:])
    normal_col = allc[1 + normal_idx]
    normal_col.fill_(1.0)
    ops.axpby_(normal_col, s, -1.0, 1.0)                          # 1 - s
    ks, vs = ops.sort_pairs_batched(allc, lab32, descending=True)
    # ... and turned into their curves by ONE batched launch sequence: problem 0 = "label != normal", problem c + 1 = "label == c"
    cv = ops.clf_curve_batched(ks, vs, [normal_idx] + list(range(Cn)), [True] + [False] * Cn, res, curves=curves)

    out: Dict[str, object] = {}
    y_pred = counts = None
    if per_frame:
        thr_dev = rec(0)[48:52].view(torch.float32)               # &result[0].opt_threshold, stays on the device
        y_pred, counts = ops.eval_counts(s, probs, lab64, Cn, normal_idx, thr_dev)
    recs = _records(res, Cn + 1)                                  # the only device->host sync of the epilogue
    b = recs[0]
    out["auc_roc"], out["auc_pr"], out["optimal_threshold"] = b.auroc, b.ap, float(b.opt_threshold)
    mc_auroc = np.array([r.auroc for r in recs[1:]])
    mc_aupr = np.array([r.ap for r in recs[1:]])
    out["mc_auroc"], out["mc_aupr"] = mc_auroc.tolist(), mc_aupr.tolist()
    out["mean_mc_auroc"] = _mean_without_normal(mc_auroc, normal_idx)
    out["mean_mc_aupr"] = _mean_without_normal(mc_aupr, normal_idx)
    if per_frame:
        cn = counts.cpu().numpy()
        t1, t5, cls_n = cn[:Cn], cn[Cn:2 * Cn], cn[2 * Cn:3 * Cn]
        with np.errstate(invalid="ignore", divide="ignore"):
            out["top1_accuracy"] = (t1 / cls_n).tolist()          # nan for classes without frames, like .mean() of empty
            out["top5_accuracy"] = (t5 / cls_n).tolist()
            cm = cn[3 * Cn:3 * Cn + Cn * Cn].reshape(Cn, Cn)
            rs = cm.sum(1, keepdims=True)
            out["confusion_matrix"] = np.where(rs > 0, cm / np.maximum(rs, 1), 0.0)      # normalize="true", nan -> 0
        out["confusion_counts"] = cm
        f = cn[3 * Cn + Cn * Cn:]
        out["f1_scores"] = {(i + 1) / 10: (2 * f[i] / (2 * f[i] + f[10 + i] + f[20 + i]) if (2 * f[i] + f[10 + i] + f[20 + i]) else 0.0)
                            for i in range(10)}
        out["y_pred"] = y_pred
    if curves:
        k = int(b.n_distinct)
        tps, fps, thr = (t[:k] for t in cv)
        P, N = max(int(b.n_pos), 0), max(int(b.n_neg), 0)
        z = torch.zeros(1, device=dev)
        tpr = torch.cat([z, tps.float() / P]) if P else torch.zeros(k + 1, device=dev)
        fpr = torch.cat([z, fps.float() / N]) if N else torch.zeros(k + 1, device=dev)
        out["roc"] = (fpr, tpr, torch.cat([torch.ones(1, device=dev), thr]))
        prec = tps.float() / (tps + fps).float()
        out["pr_curve"] = (prec, tps.float() / P if P else torch.full((k,), math.nan, device=dev), thr)
    return out
