"""Frame-level ROC-AUC with the semantics of the reference's ``utils.eval_utils.eval`` (:139-143 -> cal_auc :21-24 =
sklearn ``roc_curve`` + ``auc``, pos_label=1): tie-aware trapezoid.  Host-side numpy (scores are a few thousand floats
per evaluation); pinned against the reference's own function in tests/golden/misc.npz."""
import numpy as np


def roc_auc(scores, labels) -> float:
    s = np.asarray(scores, dtype=np.float64).ravel()
    y = np.asarray(labels, dtype=np.float64).ravel()
    order = np.argsort(-s, kind="mergesort")
    s, y = s[order], y[order]
    last_of_tie = np.r_[np.nonzero(np.diff(s))[0], y.size - 1]
    tp = np.r_[0.0, np.cumsum(y)[last_of_tie]]
    fp = np.r_[0.0, 1.0 + last_of_tie - np.cumsum(y)[last_of_tie]]
    if tp[-1] == 0 or fp[-1] == 0:
        return float("nan")
    tpr, fpr = tp / tp[-1], fp / fp[-1]
    return float(np.sum((fpr[1:] - fpr[:-1]) * (tpr[1:] + tpr[:-1]) * 0.5))


def eval(total_scores, total_labels, logger=None) -> float:   # noqa: A001  (reference name)
    return roc_auc(np.array(total_scores), np.array(total_labels))
