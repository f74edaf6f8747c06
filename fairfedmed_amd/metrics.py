"""Host-side metrics the trainer reports: AUC as the reference's ``compute_auc``
(evaluation/metrics.py:340-356 -> sklearn roc_auc_score on one-hot labels,
macro average over the one-vs-rest columns) and macro-F1
(evaluation/evaluator_oph.py).  Everything else in evaluation/ (ES-AUC, DPD,
EOD ...) is out of scope for this round."""
from __future__ import annotations

import numpy as np


def _midranks(x: np.ndarray) -> np.ndarray:
    order = np.argsort(x, kind="mergesort")
    xs = x[order]
    # boundaries of runs of equal values
    new = np.concatenate(([True], xs[1:] != xs[:-1]))
    start = np.flatnonzero(new)
    end = np.concatenate((start[1:], [len(xs)]))
    ranks_sorted = np.empty(len(xs), dtype=np.float64)
    for s, e in zip(start, end):
        ranks_sorted[s:e] = 0.5 * (s + e - 1) + 1.0
    ranks = np.empty(len(xs), dtype=np.float64)
    ranks[order] = ranks_sorted
    return ranks


def auc_binary(score: np.ndarray, positive: np.ndarray) -> float:
    """Mann-Whitney U / (n0 n1) with mid-ranks: the area under sklearn's ROC curve."""
    score = np.asarray(score, dtype=np.float64)
    pos = np.asarray(positive).astype(bool)
    n1 = int(pos.sum())
    n0 = len(pos) - n1
    if n0 == 0 or n1 == 0:
        return 1.0
    r = _midranks(score)
    return float((r[pos].sum() - n1 * (n1 + 1) / 2.0) / (n0 * n1))


def auc_macro_ovr(prob: np.ndarray, label: np.ndarray) -> float:
    """Mean over classes of the one-vs-rest AUC of that class's probability column.
    A single-class batch reports 1 (trainers/GLP_OT_SVLoRA.py:965-967)."""
    prob = np.asarray(prob)
    y = np.asarray(label).astype(np.int64)
    if y.min() == y.max():
        return 1.0
    return float(np.mean([auc_binary(prob[:, c], y == c) for c in range(prob.shape[1])]))


def macro_f1(pred: np.ndarray, label: np.ndarray, num_classes: int) -> float:
    f = []
    for c in range(num_classes):
        tp = float(np.sum((pred == c) & (label == c)))
        fp = float(np.sum((pred == c) & (label != c)))
        fn = float(np.sum((pred != c) & (label == c)))
        d = 2 * tp + fp + fn
        f.append(0.0 if d == 0 else 2 * tp / d)
    return float(np.mean(f))
