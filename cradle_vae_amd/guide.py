"""Variational guide q(M) q(E|M) q(qc_E) q(z_basal | x, M, E, qc) -- mirror of the reference's
``cradle_vae_CorrelatedNormalGuide`` (cradle_vae/models/cradle_vae/guides/
correlated_normal_guide.py:18-215) on the HIP kernels."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Literal, Optional, Tuple

import torch
from torch import nn

from .distributions import GumbelSoftmaxBernoulliStraightThrough
from .mlp import get_likelihood_mlp


@dataclass
class ObservationNormalizationStatistics:
    """cradle_vae/data/utils/perturbation_datamodule.py:10-15."""

    x_mean: Optional[torch.Tensor] = None
    x_std: Optional[torch.Tensor] = None
    log_x_mean: Optional[torch.Tensor] = None
    log_x_std: Optional[torch.Tensor] = None


class LogStandardizationModule(nn.Module):
    """cradle_vae/models/utils/normalization.py:35-45."""

    def __init__(self, normalization_stats):
        super().__init__()
        self.register_buffer("log_mean", normalization_stats.log_x_mean)
        self.register_buffer("log_scale", normalization_stats.log_x_std)

    def forward(self, x):
        normx = (torch.log(x + 1) - self.log_mean) / self.log_scale
        return torch.nan_to_num(normx, nan=1e-6, posinf=1e6, neginf=1e-6)


class cradle_vae_CorrelatedNormalGuide(nn.Module):
    def __init__(
        self,
        n_latent: int,
        n_treatments: int,
        n_phenos: int,
        n_qc: int,
        basal_encoder_n_layers: int,
        basal_encoder_n_hidden: int,
        basal_encoder_input_normalization: Optional[Literal["standardize", "log_standardize"]],
        embedding_encoder_n_layers: int,
        embedding_encoder_n_hidden: int,
        x_normalization_stats: Optional[ObservationNormalizationStatistics],
        mask_init_logits: float = 0,
        gs_temperature: float = 1,
        mean_field_encoder: bool = False,
    ):
        super().__init__()
        if mean_field_encoder:
            raise NotImplementedError("mean_field_encoder=True: no shipped config uses it (SURVEY.md section 2)")
        if basal_encoder_input_normalization != "log_standardize":
            raise NotImplementedError("only basal_encoder_input_normalization='log_standardize' is on the HIP path")
        assert x_normalization_stats is not None, "Missing x_normalization_stats"
        self.n_latent, self.n_treatments, self.n_phenos, self.n_qc = n_latent, n_treatments, n_phenos, n_qc
        self.basal_encoder_input_normalization = basal_encoder_input_normalization
        self.x_normalization_stats = x_normalization_stats
        self.mean_field_encoder = mean_field_encoder

        self.param_dict = nn.ParameterDict()
        self.param_dict["q_mask_logits"] = nn.Parameter(mask_init_logits * torch.ones((n_treatments, n_latent)))
        self.embedding_encoder = get_likelihood_mlp(
            "normal", n_latent + n_treatments, n_latent, embedding_encoder_n_layers, embedding_encoder_n_hidden)
        # kept for state-dict compatibility (correlated_normal_guide.py:64); the HIP path never
        # multiplies by it: the one-hot block of the first layer is a column gather
        self.register_buffer("treatment_one_hot", torch.eye(n_treatments))
        self.normalization_module = LogStandardizationModule(x_normalization_stats)
        self.register_buffer("qc_encoder_input", torch.ones((n_qc, n_latent)))
        self.qc_encoder = get_likelihood_mlp(
            "normal", n_latent, n_latent, embedding_encoder_n_layers, embedding_encoder_n_hidden)
        self.z_basal_encoder = get_likelihood_mlp(
            "normal", n_phenos + 2 * n_latent, n_latent, basal_encoder_n_layers, basal_encoder_n_hidden)
        self.register_buffer("gs_temperature", gs_temperature * torch.ones((1,)))
        self.var_eps = 1e-4

    def get_var_keys(self):
        return ["z_basal", "E", "mask", "qc_E"]

    @staticmethod
    def sanitize_tensor(tensor):
        tensor = torch.where(torch.isnan(tensor), torch.zeros_like(tensor), tensor)
        return torch.where(torch.isinf(tensor), torch.zeros_like(tensor), tensor)

    def forward(
        self,
        X: Optional[torch.Tensor] = None,
        D: Optional[torch.Tensor] = None,
        qc: Optional[torch.Tensor] = None,
        condition_values: Optional[Dict[str, torch.Tensor]] = None,
        n_particles: int = 1,
        noise: Optional[Dict[str, torch.Tensor]] = None,
        need_cf_basal: bool = True,
    ) -> Tuple[Dict[str, object], Dict[str, torch.Tensor]]:
        """correlated_normal_guide.py:102-209.  ``noise`` (keys u_mask, eps_E, eps_qc_E, eps_z)
        injects the random draws; missing keys are drawn with torch's generator.
        ``need_cf_basal=False`` (inference only) skips ``q_z_cf_basal``, which the predictor never reads."""
        cv = {} if condition_values is None else condition_values
        noise = {} if noise is None else noise
        P = n_particles
        dists: Dict[str, object] = {}
        samples: Dict[str, torch.Tensor] = {}

        def rsample(dist, key):
            eps = noise.get(key)
            if eps is None:
                return dist.rsample()
            return dist.mean + dist.stddev * eps

        logits = self.sanitize_tensor(self.param_dict["q_mask_logits"])
        dists["q_mask"] = GumbelSoftmaxBernoulliStraightThrough(temperature=self.gs_temperature, logits=logits)
        if "mask" not in cv:
            samples["mask"] = dists["q_mask"].rsample((P,), uniforms=noise.get("u_mask"))
        else:
            samples["mask"] = cv["mask"]

        one_hot = self.treatment_one_hot.unsqueeze(0).expand(P, -1, -1)
        dists["q_E"] = self.embedding_encoder(torch.cat([samples["mask"], one_hot], dim=-1))
        samples["E"] = cv["E"] if "E" in cv else rsample(dists["q_E"], "eps_E")

        dists["q_qc_E"] = self.qc_encoder(self.qc_encoder_input.unsqueeze(0).expand(P, -1, -1))
        samples["qc_E"] = cv["qc_E"] if "qc_E" in cv else rsample(dists["q_qc_E"], "eps_qc_E")

        if X is not None and D is not None and qc is not None:
            xn = self.normalization_module(X)
            latent_offset = torch.matmul(D, samples["mask"] * samples["E"])
            latent_qc_offset = torch.matmul(qc, samples["qc_E"])
            latent_qc_cf_offset = torch.matmul(1 - qc, samples["qc_E"])
            if not torch.is_grad_enabled() and X.is_cuda and latent_offset.dim() == 3:
                # inference (predictor, validation): x W_x^T once per cell, not once per (particle, cell) row
                # and per encoder pass -- SURVEY 8a restructuring (ii)
                enc = self.z_basal_encoder
                xw = enc.shared_input_part(xn)
                if need_cf_basal:
                    dists["q_z_cf_basal"] = enc.forward_shared(
                        xw, torch.cat([latent_offset, latent_qc_cf_offset], dim=-1))
                dists["q_z_basal"] = enc.forward_shared(xw, torch.cat([latent_offset, latent_qc_offset], dim=-1))
            else:
                enc_in = xn.unsqueeze(0).expand(P, -1, -1)
                dists["q_z_cf_basal"] = self.z_basal_encoder(
                    torch.cat([enc_in, latent_offset, latent_qc_cf_offset], dim=-1))
                dists["q_z_basal"] = self.z_basal_encoder(torch.cat([enc_in, latent_offset, latent_qc_offset], dim=-1))
            samples["z_basal"] = rsample(dists["q_z_basal"], "eps_z")
        if "z_basal" in cv:
            samples["z_basal"] = cv["z_basal"]
        return dists, samples
