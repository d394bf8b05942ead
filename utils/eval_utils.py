"""Reference import path ``utils.eval_utils`` -> lstc_vad_amd.metrics (``eval`` = frame-level ROC-AUC, :139-143)."""
from lstc_vad_amd.metrics import eval, roc_auc  # noqa: F401,A004

cal_auc = roc_auc
