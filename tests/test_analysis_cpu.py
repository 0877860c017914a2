"""Analysis helpers (data-side ATE, eval.py's metric stage, align_adatas, train-set log statistics)
against outputs of the reference's own functions (tests/golden/metrics_tiny.npz, written by
`python -m oracle.make_golden metrics`)."""
import os

import numpy as np
import pandas as pd
import torch

from tests.golden_io import GOLDEN


def _load():
    return np.load(os.path.join(GOLDEN, "metrics_tiny.npz"), allow_pickle=False)


def test_data_side_ate_matches_reference():
    from cradle_vae_amd.analysis.average_treatment_effects import estimate_data_average_treatment_effects
    from cradle_vae_amd.predictor import make_anndata

    z = _load()
    X, labels = z["X"], z["labels"]
    adata = make_anndata(X, obs=pd.DataFrame({"pert": labels}, index=[f"c{i}" for i in range(len(X))]),
                         var=pd.DataFrame(index=[f"gene{i}" for i in range(X.shape[1])]))
    for method in ("mean", "perturbseq"):
        res = estimate_data_average_treatment_effects(adata, "pert", "ctrl", method)
        assert list(res.obs.index) == list(z[f"data_ate/{method}_index"]) and res.uns["control"] == "ctrl"
        np.testing.assert_allclose(res.X, z[f"data_ate/{method}"], rtol=1e-5, atol=1e-6)
        assert list(res.var.index) == list(adata.var.index)


def test_align_and_metric_stage_match_reference():
    from cradle_vae_amd.analysis.metrics import align_adatas, get_ate_metrics, jaccard_sim
    from cradle_vae_amd.predictor import make_anndata

    z = _load()
    data_ate = make_anndata(z["data_ate/perturbseq"], obs=pd.DataFrame(index=z["data_ate/perturbseq_index"]))
    model_ate = make_anndata(z["model_ate/X"], obs=pd.DataFrame(index=z["model_ate/index"]))
    a, b = align_adatas(data_ate, model_ate)
    assert list(a.obs.index) == list(b.obs.index) == list(z["aligned/index"])
    np.testing.assert_array_equal(a.X, z["aligned/data_X"])
    np.testing.assert_array_equal(b.X, z["aligned/model_X"])
    got = get_ate_metrics(a, b)
    ref = {k[len("metrics/"):]: float(z[k]) for k in z.files if k.startswith("metrics/")}
    assert set(got) == set(ref)
    for k, v in ref.items():
        assert abs(got[k] - v) <= 1e-6 * max(1.0, abs(v)), (k, got[k], v)
    assert jaccard_sim(np.array([[1, 2, 3]]), np.array([[2, 3, 4]])) == 0.5


def test_log_statistics_match_reference():
    from cradle_vae_amd.data import batch_log_mean, batch_log_std

    z = _load()
    X = torch.from_numpy(z["X"])
    lm = batch_log_mean(X, step_size=100)
    ls = batch_log_std(X, lm, step_size=100)
    np.testing.assert_allclose(lm.numpy(), z["stats/log_mean"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ls.numpy(), z["stats/log_std"], rtol=1e-6, atol=1e-7)
