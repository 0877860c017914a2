"""Test-set evaluation -- mirror of the reference's ``evaluate_checkpoint`` (eval.py:41-129): predictive
IWELBO on the test split (P = 100), model-vs-data average-treatment-effect agreement for all
perturbations and per split.  The qc_ratio stage (eval.py:119-127: rapids_singlecell / scrublet on sampled
cells) is outside this build and is left out of the returned metrics.
"""
from __future__ import annotations

from typing import Any, Dict, Literal, Optional

import torch

from .analysis.metrics import align_adatas, get_ate_metrics
from .data import DeviceBatchLoader, TensorPerturbationDataset


def evaluate_module(lightning_module, data_module, average_treatment_effect_method: Literal["mean", "perturbseq"],
                    batch_size: int = 500, ate_n_particles: int = 2500, qc_pass: bool = False,
                    iwelbo_n_particles: int = 100) -> Dict[str, Any]:
    predictor = lightning_module.predictor
    device = next(lightning_module.parameters()).device
    metrics: Dict[str, Any] = {}
    ds = data_module.test_dataloader().dataset
    if isinstance(ds, TensorPerturbationDataset):
        test_loader = DeviceBatchLoader(ds, batch_size, device=device)
    else:
        test_loader = torch.utils.data.DataLoader(ds, batch_size=batch_size)
    iw = predictor.compute_predictive_iwelbo(loaders=test_loader, n_particles=iwelbo_n_particles)
    metrics["test/IWELBO"] = float(iw["IWELBO"].mean())

    data_ate = data_module.get_estimated_average_treatment_effects(method=average_treatment_effect_method, qc_pass=qc_pass)
    if data_ate is not None:
        model_ate = predictor.estimate_average_effects_data_module(
            data_module=data_module, control_label=data_ate.uns["control"], method=average_treatment_effect_method,
            n_particles=ate_n_particles, condition_values=dict(library_size=10000 * torch.ones((1,))),
            batch_size=batch_size)
        data_ate, model_ate = align_adatas(data_ate, model_ate)
        info = data_module.get_unique_observed_intervention_info()
        metrics["ATE_n_particles"] = ate_n_particles
        for k, v in get_ate_metrics(data_ate, model_ate).items():
            metrics[f"{k}-all"] = v
        for split in ("train", "val", "test"):
            idx = data_ate.obs.index.isin(info[info[split]].index)
            if idx.sum() == 0:
                continue
            for k, v in get_ate_metrics(data_ate[idx], model_ate[idx]).items():
                metrics[f"{k}-{split}"] = v
    return metrics


def evaluate_checkpoint(checkpoint_path: str, average_treatment_effect_method: Literal["mean", "perturbseq"],
                        batch_size: int = 500, ate_n_particles: int = 2500, qc_pass: bool = False, thr: int = 5,
                        devices=None, data_module: Optional[Any] = None) -> Dict[str, Any]:
    """eval.py:41-129 with the same leading arguments; ``data_module`` may be passed instead of being
    rebuilt from the checkpoint's config."""
    from .lightning_module import load_checkpoint

    device = f"cuda:{devices[0] if isinstance(devices, (list, tuple)) else (devices or 0)}"
    module = load_checkpoint(checkpoint_path, map_location=device)
    if data_module is None:
        data_module = module.get_data_module()
    return evaluate_module(module, data_module, average_treatment_effect_method, batch_size, ate_n_particles, qc_pass)
