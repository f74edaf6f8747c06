"""Host-side metrics the trainer reports: AUC as the reference's ``compute_auc``
(evaluation/metrics.py:340-356 -> sklearn roc_auc_score on one-hot labels,
macro average over the one-vs-rest columns), macro-F1
(evaluation/evaluator_oph.py) and the fairness scores of ``evalute_comprehensive_perf_scores``
(evaluation/metrics.py:197-311): per-group AUC, equity-scaled AUC, between-group disparity, and the
demographic-parity / equalized-odds differences.

The last two come from a third-party package that is absent from the build image, ``fairlearn`` (the reference
pins no version; the definitions below are those of fairlearn 0.7-0.10, ``fairlearn.metrics``:
``demographic_parity_difference`` = max - min over groups of the selection rate P(y_hat = 1 | g);
``equalized_odds_difference`` = the larger of the max - min gaps of the true-positive and false-positive rates).
They are restated from the published definition; with the package absent they cannot be pinned against the reference's
own call, so they are pinned against values derived BY HAND from that definition
(tests/test_host_cpu.py::test_dpd_eod_against_hand_derived_fairlearn_values: several groups, the -1 "unknown" value as
a group, empty-denominator rates, the single-group case).  AUC, per-group AUC, ES-AUC and the disparity ratios are
pinned against the imported reference functions (tests/golden/make_golden.py, ``fair.*``)."""
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
    """sklearn f1_score(average="macro", labels=np.unique(y_true)) as the reference's evaluator calls it
    (evaluation/evaluator_oph.py:70-75): the mean runs over the classes PRESENT in the labels only."""
    f = []
    for c in range(num_classes):
        if not np.any(label == c):
            continue
        tp = float(np.sum((pred == c) & (label == c)))
        fp = float(np.sum((pred == c) & (label != c)))
        fn = float(np.sum((pred != c) & (label == c)))
        d = 2 * tp + fp + fn
        f.append(0.0 if d == 0 else 2 * tp / d)
    return float(np.mean(f))


# ------------------------------------------------------------------------------------------------------------
# Fairness scores (evaluation/metrics.py:197-311, 513-552).  prob [N, n_cls], label [N], attr [N] (-1 = unknown).
# ------------------------------------------------------------------------------------------------------------
def group_aucs(prob: np.ndarray, label: np.ndarray, attr: np.ndarray) -> np.ndarray:
    """AUC of every group present in attr (ascending group id, -1 skipped), evaluation/metrics.py:232-244."""
    prob, label, attr = np.asarray(prob), np.asarray(label), np.asarray(attr)
    return np.array([auc_macro_ovr(prob[attr == g], label[attr == g]) for g in np.unique(attr).astype(int) if g != -1])


def equity_scaled_auc(prob: np.ndarray, label: np.ndarray, attr: np.ndarray, alpha: float = 1.0) -> float:
    """ES-AUC = AUC / (1 + alpha * sum_g |AUC_g - AUC|), evaluation/metrics.py:513-547."""
    overall = auc_macro_ovr(prob, label)
    gaps = np.abs(group_aucs(prob, label, attr) - overall).sum()
    return float(overall / (alpha * gaps + 1.0))


def between_group_disparity(aucs: np.ndarray, overall: float):
    """(std, max - min) of the group AUCs relative to the overall AUC, evaluation/metrics.py:549-550."""
    aucs = np.asarray(aucs, dtype=np.float64)
    return float(np.std(aucs) / overall), float((aucs.max() - aucs.min()) / overall)


def _rates(pred: np.ndarray, label: np.ndarray, mask: np.ndarray):
    p, y = pred[mask], label[mask]
    sel = float(p.mean()) if len(p) else 0.0
    tpr = float(p[y == 1].mean()) if np.any(y == 1) else 0.0
    fpr = float(p[y == 0].mean()) if np.any(y == 0) else 0.0
    return sel, tpr, fpr


def demographic_parity_difference(label: np.ndarray, pred: np.ndarray, attr: np.ndarray) -> float:
    """fairlearn.metrics.demographic_parity_difference: max - min over groups of P(pred = 1 | group)."""
    pred, label, attr = np.asarray(pred).astype(np.float64), np.asarray(label), np.asarray(attr)
    sel = [_rates(pred, label, attr == g)[0] for g in np.unique(attr)]
    return float(max(sel) - min(sel))


def equalized_odds_difference(label: np.ndarray, pred: np.ndarray, attr: np.ndarray) -> float:
    """fairlearn.metrics.equalized_odds_difference: max(TPR gap, FPR gap), gap = max - min over groups."""
    pred, label, attr = np.asarray(pred).astype(np.float64), np.asarray(label), np.asarray(attr)
    r = [_rates(pred, label, attr == g) for g in np.unique(attr)]
    tpr, fpr = [x[1] for x in r], [x[2] for x in r]
    return float(max(max(tpr) - min(tpr), max(fpr) - min(fpr)))


def comprehensive_scores(prob: np.ndarray, label: np.ndarray, attrs: np.ndarray) -> dict:
    """The dictionary Classification_oph.evaluate adds for binary tasks (evaluation/evaluator_oph.py:69-113):
    attrs is [n_attr, N]."""
    prob, label, attrs = np.asarray(prob), np.asarray(label), np.atleast_2d(np.asarray(attrs))
    overall = auc_macro_ovr(prob, label)
    pred = prob.argmax(-1)
    out = {"overall_auc": overall, "esaucs_by_attrs": [], "aucs_by_attrs": [], "dpds": [], "eods": [],
           "between_group_disparity": []}
    for a in attrs:
        ga = group_aucs(prob, label, a)
        out["aucs_by_attrs"].append(ga)
        out["esaucs_by_attrs"].append(equity_scaled_auc(prob, label, a))
        out["between_group_disparity"].append(between_group_disparity(ga, overall))
        out["dpds"].append(demographic_parity_difference(label, pred, a))
        out["eods"].append(equalized_odds_difference(label, pred, a))
    return out


# ------------------------------------------------------------------------------------------------------------
# The same scores from the integer counts of ffm_eval_counts (include/ffm_hip.h): rows = groups 0..G-1, unknown
# (attribute -1), all samples; columns n_pos n_neg win1 tie1 win0 tie0 TP FP TN FN.  Host arithmetic on ~100
# integers; the pass over the samples ran on the GPU.
# ------------------------------------------------------------------------------------------------------------
N_POS, N_NEG, WIN1, TIE1, WIN0, TIE0, TP, FP, TN, FN = range(10)


def _auc_from_row(r) -> float:
    """auc_macro_ovr of the samples behind one count row (1.0 for a single-class set, as above)."""
    n1, n0 = int(r[N_POS]), int(r[N_NEG])
    if n1 == 0 or n0 == 0:
        return 1.0
    a1 = (int(r[WIN1]) + 0.5 * int(r[TIE1])) / (n1 * n0)
    a0 = (int(r[WIN0]) + 0.5 * int(r[TIE0])) / (n1 * n0)
    return float(np.mean([a0, a1]))


def _rates_from_row(r):
    tp, fp, tn, fn = (int(r[k]) for k in (TP, FP, TN, FN))
    n = tp + fp + tn + fn
    sel = (tp + fp) / n if n else 0.0
    tpr = tp / (tp + fn) if (tp + fn) else 0.0
    fpr = fp / (fp + tn) if (fp + tn) else 0.0
    return sel, tpr, fpr


def basic_from_counts(counts: np.ndarray):
    """[accuracy %, error %, macro-F1 %, AUC %] of SimpleTrainer.test from the 'all' row: the first four entries of
    Classification_oph.evaluate's results (evaluation/evaluator_oph.py:66-96: macro-F1 over the classes present in the
    labels, 100 * compute_auc)."""
    r = np.asarray(counts)[-1]
    tp, fp, tn, fn = (int(r[k]) for k in (TP, FP, TN, FN))
    n = tp + fp + tn + fn
    acc = 100.0 * (tp + tn) / n
    f = []
    for a, b, c, present in ((tn, fn, fp, tn + fp > 0), (tp, fp, fn, tp + fn > 0)):   # class 0: tp' = TN, fp' = FN, fn' = FP
        if not present:
            continue
        d = 2 * a + b + c
        f.append(0.0 if d == 0 else 2 * a / d)
    return [acc, 100.0 - acc, 100.0 * float(np.mean(f)), 100.0 * _auc_from_row(r)]


def comprehensive_scores_from_counts(counts_by_attr) -> dict:
    """comprehensive_scores() from one count table per attribute column."""
    tables = [np.asarray(c) for c in counts_by_attr]
    overall = _auc_from_row(tables[0][-1])
    out = {"overall_auc": overall, "esaucs_by_attrs": [], "aucs_by_attrs": [], "dpds": [], "eods": [],
           "between_group_disparity": []}
    for t in tables:
        groups, unknown = t[:-2], t[-2]
        present = [g for g in groups if int(g[N_POS]) + int(g[N_NEG]) > 0]
        ga = np.array([_auc_from_row(g) for g in present])
        out["aucs_by_attrs"].append(ga)
        out["esaucs_by_attrs"].append(float(overall / (np.abs(ga - overall).sum() + 1.0)))
        out["between_group_disparity"].append(between_group_disparity(ga, overall))
        rows = ([unknown] if int(unknown[N_POS]) + int(unknown[N_NEG]) > 0 else []) + present   # -1 is a group here
        r = [_rates_from_row(g) for g in rows]
        sel, tpr, fpr = [x[0] for x in r], [x[1] for x in r], [x[2] for x in r]
        out["dpds"].append(float(max(sel) - min(sel)))
        out["eods"].append(float(max(max(tpr) - min(tpr), max(fpr) - min(fpr))))
    return out
