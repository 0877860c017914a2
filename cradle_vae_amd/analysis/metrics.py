"""ATE / DEG agreement metrics between the data-side and the model-side average treatment effects
-- mirror of ``get_ate_metrics`` / ``jaccard_sim`` (reference eval.py:132-175) and
``align_adatas`` (cradle_vae/data/utils/anndata.py:6-14).

Host-side numpy on (n_perturbations, n_genes) arrays, as in the reference; nothing here is on the
training or sampling hot path.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import pandas as pd


def _top_k_by_abs(X: np.ndarray, k: int) -> np.ndarray:
    """Indices of the k largest |X| per row (unordered within the k), eval.py:134-146."""
    return np.argpartition(np.abs(X), X.shape[1] - k, axis=1)[:, -k:]


def _pearsonr(x: np.ndarray, y: np.ndarray) -> float:
    x = np.asarray(x, dtype=np.float64) - np.mean(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64) - np.mean(y, dtype=np.float64)
    return float(np.dot(x, y) / np.sqrt(np.dot(x, x) * np.dot(y, y)))


def _r2_score(y_true: np.ndarray, y_pred: np.ndarray) -> float:
    """sklearn.metrics.r2_score(y_true, y_pred) for 1-D input: 1 - SS_res / SS_tot."""
    y_true = np.asarray(y_true, dtype=np.float64)
    y_pred = np.asarray(y_pred, dtype=np.float64)
    ss_res = float(np.sum((y_true - y_pred) ** 2))
    ss_tot = float(np.sum((y_true - y_true.mean()) ** 2))
    return 1.0 - ss_res / ss_tot


def jaccard_sim(X: np.ndarray, Y: np.ndarray) -> float:
    """Mean over rows of |set(x) & set(y)| / |set(x) | set(y)| (eval.py:173-175)."""
    vals = []
    for x, y in zip(X, Y):
        inter = np.intersect1d(x, y).size
        vals.append(inter / (np.unique(x).size + np.unique(y).size - inter))
    return float(np.mean(vals))


def get_ate_metrics(data_ate, model_ate) -> Dict[str, float]:
    """eval.py:132-170.  ``data_ate`` / ``model_ate``: objects with ``.X`` (n_perturbations, n_genes),
    row-aligned (see ``align_adatas``)."""
    dX, mX = np.asarray(data_ate.X), np.asarray(model_ate.X)
    top20_d, top20_m = _top_k_by_abs(dX, 20), _top_k_by_abs(mX, 20)
    top50_d, top50_m = _top_k_by_abs(dX, 50), _top_k_by_abs(mX, 50)
    metrics = {}
    metrics["ATE_pearsonr"] = _pearsonr(dX.ravel(), mX.ravel())
    metrics["ATE_r2"] = _r2_score(dX.ravel(), mX.ravel())
    x = np.take_along_axis(dX, top20_d, axis=-1).ravel()  # the data's top-20 DE genes per perturbation
    y = np.take_along_axis(mX, top20_d, axis=-1).ravel()
    metrics["ATE_pearsonr_top20"] = _pearsonr(x, y)
    metrics["ATE_r2_top20"] = _r2_score(x, y)
    metrics["jaccard_sim_top20"] = jaccard_sim(top20_d, top20_m)
    metrics["jaccard_sim_top50"] = jaccard_sim(top50_d, top50_m)
    return metrics


def align_adatas(*adatas):
    """Rows of every input restricted to, and ordered by, the obs-index keys they all share
    (inner join in the order of the first input)."""
    frames = [pd.DataFrame({i: np.arange(len(a)), "key": a.obs.index.values}) for i, a in enumerate(adatas)]
    joined = frames[0]
    for f in frames[1:]:
        joined = joined.merge(f, on="key", how="inner")
    return [a[joined[i].values] for i, a in enumerate(adatas)]
