"""Mirror of ``cradle_vae/analysis/average_treatment_effects.py``: the model-side estimator lives in
``cradle_vae_amd.ate`` (HIP path); the data-side estimator (:11-75) is host numpy."""
from __future__ import annotations

from typing import Any, Literal

import numpy as np
import pandas as pd

from ..ate import estimate_model_average_treatment_effect  # noqa: F401


def estimate_data_average_treatment_effects(adata, label_col: str, control_label: Any,
                                            method: Literal["mean", "perturbseq"], compute_fdr: bool = False):
    """Mean (optionally library-normalised, log2-transformed) expression of every perturbation label
    minus that of the control label.  -> AnnData-like with the effects in X, the alternative labels
    as obs index (order of first appearance), ``adata.var`` and ``uns['control']``."""
    from ..predictor import make_anndata

    if compute_fdr:
        raise NotImplementedError
    assert method in ("mean", "perturbseq"), "Method must be one of ['mean', 'perturbseq']"
    labels = adata.obs[label_col]
    perturbations = list(pd.unique(labels))
    assert control_label in perturbations
    alt_labels = [x for x in perturbations if x != control_label]
    X = adata.X
    if hasattr(X, "toarray"):  # scipy sparse
        X = X.toarray()
    X = np.asarray(X)
    if method == "perturbseq":
        X = np.log2(1e4 * X / np.sum(X, axis=1, keepdims=True) + 1)
    lab = labels.to_numpy()
    control_mean = X[lab == control_label].mean(0)
    effects = np.stack([X[lab == a].mean(0) - control_mean for a in alt_labels])
    return make_anndata(effects, obs=pd.DataFrame(index=alt_labels), var=adata.var.copy(), uns=dict(control=control_label))
