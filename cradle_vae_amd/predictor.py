"""Predictor -- mirror of the reference's ``PerturbationPlatedPredictor``
(cradle_vae/models/utils/predictor.py:24-410) and ``cradle_vae_Predictor``
(cradle_vae/models/cradle_vae/predictor.py:6-13) on the HIP kernels.

Results come back in ``anndata.AnnData`` when anndata is importable, otherwise in
``AnnDataLite`` (same ``X`` / ``obs`` / ``var`` / ``uns`` attributes).
``estimate_qc_ratio`` (predictor.py:412-438) needs rapids_singlecell/cupy/scanpy, which
are CUDA-only third-party tools outside this path; it raises ``NotImplementedError``.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Literal, Optional, Sequence

import numpy as np
import pandas as pd
import torch
from torch import nn

from .ate import estimate_model_average_treatment_effect

try:  # pragma: no cover - anndata is absent from the build image
    from anndata import AnnData as _AnnData
except Exception:  # noqa: BLE001
    _AnnData = None


class AnnDataLite:
    """The slice of the AnnData surface the reference's callers of the predictor touch
    (eval.py:88-133): ``X``, ``obs``, ``var``, ``uns``, ``shape``, ``obs_names``/``var_names``."""

    def __init__(self, X=None, obs: Optional[pd.DataFrame] = None, var: Optional[pd.DataFrame] = None, uns=None):
        self.X = np.asarray(X)
        self.obs = obs if obs is not None else pd.DataFrame(index=pd.RangeIndex(self.X.shape[0]).astype(str))
        self.var = var if var is not None else pd.DataFrame(index=pd.RangeIndex(self.X.shape[1]).astype(str))
        self.uns = dict(uns) if uns is not None else {}

    @property
    def shape(self):
        return self.X.shape

    def __len__(self):
        return self.X.shape[0]

    def __getitem__(self, idx):
        """Row selection by boolean mask or integer positions (``adata[mask]``, eval.py:115)."""
        idx = np.asarray(idx)
        obs = self.obs[idx] if idx.dtype == bool else self.obs.iloc[idx]
        return AnnDataLite(X=self.X[idx], obs=obs, var=self.var, uns=self.uns)

    @property
    def obs_names(self):
        return self.obs.index

    @property
    def var_names(self):
        return self.var.index


def make_anndata(X, obs=None, var=None, uns=None):
    if _AnnData is not None:  # pragma: no cover
        kw = {k: v for k, v in dict(obs=obs, var=var, uns=uns).items() if v is not None}
        return _AnnData(X=X, **kw)
    return AnnDataLite(X=X, obs=obs, var=var, uns=uns)


class PerturbationPlatedPredictor(nn.Module):
    def __init__(
        self,
        model: nn.Module,
        guide: nn.Module,
        local_variables: Optional[Iterable[str]] = None,
        perturbation_plated_variables: Optional[Iterable[str]] = None,
        dosage_independent_variables: Optional[Iterable[str]] = None,
    ):
        super().__init__()
        local_variables = list(local_variables) if local_variables is not None else []
        perturbation_plated_variables = (
            list(perturbation_plated_variables) if perturbation_plated_variables is not None else [])
        # predictor.py:44-57
        assert sorted(model.get_var_keys()) == sorted(guide.get_var_keys()), "Mismatch in model and guide variables"
        variables = local_variables + perturbation_plated_variables
        assert sorted(list(model.get_var_keys())) == sorted(variables), (
            "Mismatch between model variables and variables specified to loss module")
        if dosage_independent_variables is not None:
            assert set(dosage_independent_variables).issubset(set(variables))
        self.model = model.eval()
        self.guide = guide.eval()
        self.local_variables = local_variables
        self.perturbation_plated_variables = perturbation_plated_variables
        # IWELBO with qc in {0,1}: score the live decoder variant only (falls back to both by itself)
        self.live_variant_only = not model.sum_mode
        self._nonbinary_qc = None
        self.dosage_independent_variables = dosage_independent_variables

    def _get_device(self):
        return next(self.model.parameters()).device

    # ------------------------------------------------------------------ IWELBO
    @torch.no_grad()
    def iwelbo_batch(self, batch: Dict[str, torch.Tensor], plated: Dict[str, torch.Tensor], n_particles: int,
                     noise: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
        """One batch of predictor.py:125-185 -> (n,) log-mean-exp over particles of
        log p(x|z,...) + log p(z) - log q(z|x,...), the plated variables held at ``plated``."""
        X, D, qc = batch["X"], batch["D"], batch["qc"]
        guide_dists, guide_samples = self.guide(
            X=X, D=D, qc=qc, condition_values=plated, n_particles=n_particles, noise=noise, need_cf_basal=False)
        guide_samples = dict(guide_samples)
        guide_samples["library_size"] = batch["library_size"]
        b_qc = qc.max(dim=1)[0] if qc.shape[1] > 1 else qc[:, 0]  # :160-164
        if self.live_variant_only:
            # (1 - b) lp_good + b lp_bad with b in {0,1} is the live variant's term: one decoder pass and
            # one K1 launch instead of two; non-binary qc is recorded and the caller recomputes with both
            model_dists, _ = self.model(D=D, qc=qc, condition_values=guide_samples, n_particles=n_particles,
                                        sample_x=False, live_variant=b_qc)
            terms = model_dists["p_x_live"].log_prob_sum(X)  # (P, n): K1 forward
            bad = ((b_qc != 0) & (b_qc != 1)).any()
            self._nonbinary_qc = bad if self._nonbinary_qc is None else (self._nonbinary_qc | bad)
        else:
            model_dists, _ = self.model(D=D, qc=qc, condition_values=guide_samples, n_particles=n_particles,
                                        sample_x=False)
            lp_good = model_dists["p_x_good"].log_prob_sum(X)  # (P, n): K1 forward
            lp_bad = model_dists["p_x_bad"].log_prob_sum(X)
            terms = (1 - b_qc) * lp_good + b_qc * lp_bad
        for var_name in self.local_variables:  # :166-177
            v = guide_samples[var_name]
            terms = terms + model_dists[f"p_{var_name}"].log_prob(v).sum(-1) - guide_dists[f"q_{var_name}"].log_prob(v).sum(-1)
        return torch.logsumexp(terms, dim=0) - math.log(n_particles)  # :183-185

    @torch.no_grad()
    def compute_predictive_iwelbo(self, loaders, n_particles: int,
                                  noise: Optional[Dict[str, torch.Tensor]] = None) -> pd.DataFrame:
        """predictor.py:77-200.  ``noise`` (u_mask, eps_E, eps_qc_E (P,...) and eps_z (P, n_total, L)
        in loader order) injects the draws for parity tests."""
        if not isinstance(loaders, (list, tuple)):
            loaders = [loaders]
        device = self._get_device()
        self._nonbinary_qc = None
        self.guide.dosage_overflow = self.model.dosage_overflow = None
        _, guide_samples = self.guide(n_particles=n_particles, noise=noise)  # :104
        plated = {k: guide_samples[k] for k in self.perturbation_plated_variables}
        id_list, iwelbo_list = [], []
        seen = 0
        for loader in loaders:
            idx_list_curr = []
            for batch in loader:
                # (the cell indices are only read on the host, after the loop: a loader that keeps them there is not made
                # to copy them to the device batch by batch -- a pageable copy stalls the host ~0.3 ms each time)
                batch = {k: (v.to(device) if torch.is_tensor(v) and k != "idx" else v) for k, v in batch.items()}
                idx_list_curr.append(batch["idx"].detach())  # moved to the host once, after the loop: no per-batch sync
                n = batch["X"].shape[0]
                nz = None
                if noise is not None and "eps_z" in noise:
                    nz = {"eps_z": noise["eps_z"][:, seen:seen + n].to(device).contiguous()}
                seen += n
                iwelbo_list.append(self.iwelbo_batch(batch, plated, n_particles, nz))
            idx_curr = torch.cat(idx_list_curr).cpu().numpy() if idx_list_curr else np.zeros((0,), dtype=np.int64)
            id_list.append(loader.dataset.convert_idx_to_ids(idx_curr))
        over = [m.dosage_overflow for m in (self.guide, self.model) if m.dosage_overflow is not None]
        if over and any(int(f.item()) & 1 for f in over) and not self.guide.dense_dosage_matmul:
            # a cell with more non-zero dosages than the gathers planned for: redo with dense products
            self.guide.dense_dosage_matmul = self.model.dense_dosage_matmul = True
            try:
                return self.compute_predictive_iwelbo(loaders, n_particles, noise)
            finally:
                self.guide.dense_dosage_matmul = self.model.dense_dosage_matmul = False
        if self.live_variant_only and self._nonbinary_qc is not None and bool(self._nonbinary_qc):
            self.live_variant_only = False  # fractional qc weights both variants: score both
            try:
                return self.compute_predictive_iwelbo(loaders, n_particles, noise)
            finally:
                self.live_variant_only = True
        values = torch.cat(iwelbo_list).cpu().numpy() if iwelbo_list else np.zeros((0,), dtype=np.float32)
        return pd.DataFrame(index=np.concatenate(id_list), columns=["IWELBO"], data=values.reshape(-1, 1))

    # ------------------------------------------------------------------ sampling
    @torch.no_grad()
    def sample_observations(
        self,
        dosages: torch.Tensor,
        perturbation_names: Optional[Sequence[str]],
        n_particles: int = 1,
        condition_values: Optional[Dict[str, torch.Tensor]] = None,
        x_var_info: Optional[pd.DataFrame] = None,
        qc=None,
        *,
        sampler: Literal["nb", "mean"] = "nb",
        noise: Optional[Dict[str, torch.Tensor]] = None,
        particle_idx: Literal["reference", "particle"] = "reference",
        seed: Optional[int] = None,
    ):
        """predictor.py:202-273 -> (adata_good, adata_bad); rows ordered perturbation-major
        (all particles of dosage 0, then dosage 1, ...) like the reference's concatenation.

        All dosages are decoded in one pass and drawn by one sampler launch per variant instead
        of the reference's per-dosage loop (:239-245); each dosage still gets its own prior draw of
        ``z_basal`` (model.py:112), as in the reference's per-dosage model calls.

        ``particle_idx="reference"`` (default) writes what the reference writes into that column,
        ``np.tile(np.arange(n_dosages), n_particles)`` (:251 -- a dosage counter); ``"particle"`` writes the
        particle index the column name promises.  ``sampler="mean"`` replaces the count draw by the distribution
        mean and ``noise`` (u_mask, eps_E, eps_qc_E (P, ...), z (n_dosages, P, 1, L)) injects the latent draws:
        the deterministic form the golden fixture pins against the reference."""
        device = self._get_device()
        dosages = dosages.to(device=device, dtype=torch.float32)
        qc = qc.to(device=device, dtype=torch.float32)
        noise = {} if noise is None else {k: v.to(device=device, dtype=torch.float32) for k, v in noise.items()}
        _, guide_samples = self.guide(n_particles=n_particles, noise=noise)  # :230
        cv = {} if condition_values is None else {k: v.to(device) for k, v in condition_values.items()}
        for var_name in self.perturbation_plated_variables:
            cv[var_name] = guide_samples[var_name]
        n = dosages.shape[0]
        if "z" in noise:  # per-dosage prior draws (n, P, 1, L) -> (P, n, L)
            cv["z_basal"] = noise["z"][:, :, 0, :].permute(1, 0, 2).contiguous()
        qc_rows = qc.expand(n, -1) if qc.shape[0] == 1 else qc
        k = int((dosages != 0).sum(dim=1).max().item()) if dosages.numel() else 1
        dense0, kmax0 = self.model.dense_dosage_matmul, self.model.max_dosages_per_cell
        self.model.dense_dosage_matmul, self.model.max_dosages_per_cell = dense0 or k > 64, max(kmax0, min(k, 64))
        try:
            dists, _ = self.model(D=dosages, qc=qc_rows, condition_values=cv, n_particles=n_particles, sample_x=False)
        finally:
            self.model.dense_dosage_matmul, self.model.max_dosages_per_cell = dense0, kmax0
        out = []
        for j, key in enumerate(("p_x_good", "p_x_bad")):
            if sampler == "mean":
                x = dists[key].mean
            else:
                x = dists[key].sample(seed=None if seed is None else 2 * seed + j)  # (P, n, G)
            out.append(x.permute(1, 0, 2).reshape(n * n_particles, -1).cpu().numpy())
        adatas = []
        for x in out:
            obs = pd.DataFrame(index=np.arange(x.shape[0]))
            obs["perturbation_idx"] = np.repeat(np.arange(n), n_particles)
            if particle_idx == "reference":
                obs["particle_idx"] = np.tile(np.arange(n), n_particles)  # as written at predictor.py:251
            else:
                obs["particle_idx"] = np.tile(np.arange(n_particles), n)
            if perturbation_names is not None:
                obs["perturbation_name"] = np.array(perturbation_names)[obs["perturbation_idx"].to_numpy()]
            adatas.append(make_anndata(x, obs=obs, var=x_var_info.copy() if x_var_info is not None else None))
        return adatas[0], adatas[1]

    def sample_observations_data_module(self, data_module, n_particles: int,
                                        condition_values: Optional[Dict[str, torch.Tensor]] = None, qc=None):
        """predictor.py:275-310."""
        perturbation_names = data_module.get_unique_observed_intervention_info().index
        D = data_module.get_unique_observed_intervention_dosages(perturbation_names)
        return self.sample_observations(
            dosages=D, perturbation_names=perturbation_names, x_var_info=data_module.get_x_var_info(),
            n_particles=n_particles, condition_values=condition_values, qc=qc)

    # ------------------------------------------------------------------ ATE
    @torch.no_grad()
    def estimate_average_treatment_effects(
        self,
        dosages_alt: torch.Tensor,
        dosages_control: torch.Tensor,
        quality: torch.Tensor,
        method: Literal["mean", "perturbseq"],
        n_particles: int = 1000,
        condition_values: Optional[Dict[str, torch.Tensor]] = None,
        perturbation_names_alt: Optional[Sequence[str]] = None,
        perturbation_name_control: Optional[str] = None,
        x_var_info: Optional[pd.DataFrame] = None,
        batch_size: int = 500,
        **estimator_kwargs,
    ):
        """predictor.py:312-371."""
        device = self._get_device()
        cv = None if condition_values is None else {k: v.to(device) for k, v in condition_values.items()}
        ate = estimate_model_average_treatment_effect(
            model=self.model, guide=self.guide, dosages_alt=dosages_alt.to(device),
            dosages_control=dosages_control.to(device), quality=quality.to(device), n_particles=n_particles,
            method=method, condition_values=cv, batch_size=batch_size,
            dosage_independent_variables=self.dosage_independent_variables, **estimator_kwargs)
        obs = pd.DataFrame(index=np.array(perturbation_names_alt)) if perturbation_names_alt is not None else None
        uns = {"control": perturbation_name_control} if perturbation_name_control is not None else None
        return make_anndata(ate, obs=obs, var=x_var_info.copy() if x_var_info is not None else None, uns=uns)

    def estimate_average_effects_data_module(
        self, data_module, control_label: str, method: Literal["mean", "perturbseq"], n_particles: int = 100,
        condition_values: Optional[Dict[str, torch.Tensor]] = None, batch_size: int = 500, **estimator_kwargs,
    ):
        """predictor.py:373-410."""
        perturbation_names = data_module.get_unique_observed_intervention_info().index
        perturbation_names_alt = [name for name in perturbation_names if name != control_label]
        dosages_alt = data_module.get_unique_observed_intervention_dosages(perturbation_names_alt)
        dosages_ref = data_module.get_unique_observed_intervention_dosages([control_label])
        good_quality = torch.ones((1, data_module.get_qc_var_info().shape[0]))  # :394
        return self.estimate_average_treatment_effects(
            dosages_alt=dosages_alt, dosages_control=dosages_ref, quality=good_quality, method=method,
            n_particles=n_particles, condition_values=condition_values, perturbation_names_alt=perturbation_names_alt,
            perturbation_name_control=control_label, x_var_info=data_module.get_x_var_info(), batch_size=batch_size,
            **estimator_kwargs)

    def estimate_qc_ratio(self, *args, **kwargs):
        raise NotImplementedError(
            "estimate_qc_ratio (predictor.py:412-438) post-processes samples with rapids_singlecell / cupy "
            "(CUDA-only third-party tools); it is outside the MI355X hot path. Use sample_observations_data_module "
            "and annotate the counts with a CPU QC tool.")


class cradle_vae_Predictor(PerturbationPlatedPredictor):
    """cradle_vae/models/cradle_vae/predictor.py:6-13."""

    def __init__(self, model: nn.Module, guide: nn.Module):
        super().__init__(model=model, guide=guide, local_variables=["z_basal"],
                         perturbation_plated_variables=["E", "mask", "qc_E"])
