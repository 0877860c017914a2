"""Experiment configuration -- the reference's yaml semantics (train_norman.py:163-227; identical in
train_{adamson,dixit,replogle}.py).

A config file is a flat mapping.  ``preprocess_config``: a key ``a--b`` gives its value to both ``a`` and ``b``
(:167-174); a dotted key ``model_kwargs.n_latent`` nests one level (:180-185); a top-level ``n_latent`` is copied
into ``model_kwargs`` and ``guide_kwargs`` (:188-190); ``gradient_clip_norm: -1`` means no clipping (:193-194).
``add_data_info_to_config`` injects what only the data knows (:201-227): treatment / gene / qc counts, the
train-set normalisation statistics, the pass / fail cell counts.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Any, DefaultDict, Dict


def load_config(path: str) -> Dict[str, Any]:
    import yaml

    with open(path) as f:
        cfg = yaml.safe_load(f)
    return {} if cfg is None else cfg


def preprocess_config(config: Dict[str, Any]) -> DefaultDict[str, Any]:
    """train_norman.py:163-198 (without its print)."""
    shared: Dict[str, Any] = {}
    for key, value in config.items():
        for k in (key.split("--") if "--" in key else [key]):
            shared[k] = value
    out: DefaultDict[str, Any] = defaultdict(dict)
    for key, value in shared.items():
        if "." in key:
            group, rest = key.split(".", maxsplit=1)  # one level of nesting: "a.b.c" -> out["a"]["b.c"]
            out[group][rest] = value
        else:
            out[key] = value
    if "n_latent" in out:
        out["model_kwargs"]["n_latent"] = out["n_latent"]
        out["guide_kwargs"]["n_latent"] = out["n_latent"]
    if out.get("gradient_clip_norm") == -1:
        out["gradient_clip_norm"] = None
    return out


def add_data_info_to_config(config: Dict[str, Any], data_module) -> Dict[str, Any]:
    """train_norman.py:201-227 (the wandb side effect at :222-225 left out)."""
    config.setdefault("model_kwargs", dict())
    config.setdefault("guide_kwargs", dict())
    n_treatments = data_module.get_d_var_info().shape[0]
    n_phenos = data_module.get_x_var_info().shape[0]
    n_qc = data_module.get_qc_var_info().shape[0]
    for group in ("model_kwargs", "guide_kwargs"):
        config[group]["n_treatments"] = n_treatments
        config[group]["n_phenos"] = n_phenos
    config["guide_kwargs"]["x_normalization_stats"] = data_module.get_x_train_statistics()
    config["model_kwargs"]["n_qc"] = n_qc
    config["guide_kwargs"]["n_qc"] = n_qc
    qc_n = data_module.get_qc_n()
    config["data_module_kwargs"]["n_qc_pass"] = qc_n["n_qc_pass"]  # KeyError without data_module_kwargs, as in the reference
    config["data_module_kwargs"]["n_qc_fail"] = qc_n["n_qc_fail"]
    return config
