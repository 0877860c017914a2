"""Generative model p(z_basal) p(E) p(M) p(qc_E) p(x | ...) -- mirror of the reference's
``cradle_vae_Model`` (cradle_vae/models/cradle_vae/model.py:9-144) on the HIP kernels."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
from torch.distributions import Bernoulli, Normal

from .mlp import LIKELIHOOD_KEY_DTYPE, get_likelihood_mlp


class cradle_vae_Model(torch.nn.Module):
    def __init__(
        self,
        n_latent: int,
        n_treatments: int,
        n_phenos: int,
        n_qc: int,
        mask_prior_prob: float,
        embedding_prior_scale: float,
        likelihood_key: LIKELIHOOD_KEY_DTYPE,
        decoder_n_layers: int,
        decoder_n_hidden: int,
        sum_mode: bool = True,
    ):
        super().__init__()
        self.n_latent = n_latent
        self.n_treatments = n_treatments
        self.n_phenos = n_phenos
        self.n_qc = n_qc
        self.likelihood_key = likelihood_key
        self.decoder_n_layers = decoder_n_layers
        self.decoder_n_hidden = decoder_n_hidden
        self.sum_mode = sum_mode
        self.max_dosages_per_cell = 8  # see guide.dosage_product
        self.dense_dosage_matmul = False
        self.dosage_overflow = None
        # model.py:33-44
        self.register_buffer("p_E_loc", torch.zeros((n_treatments, n_latent)))
        self.register_buffer("p_E_scale", embedding_prior_scale * torch.ones((n_treatments, n_latent)))
        self.register_buffer("p_mask_probs", mask_prior_prob * torch.ones((n_treatments, n_latent)))
        self.register_buffer("p_qc_E_loc", torch.zeros((n_qc, n_latent)))
        self.register_buffer("p_qc_E_scale", embedding_prior_scale * torch.ones((n_qc, n_latent)))
        # model.py:46-54
        self.decoder = get_likelihood_mlp(
            likelihood_key=likelihood_key,
            n_input=n_latent if sum_mode else 3 * n_latent,
            n_output=n_phenos,
            n_layers=decoder_n_layers + 1,
            n_hidden=decoder_n_hidden,
        )

    def get_var_keys(self) -> List[str]:
        return ["z_basal", "E", "mask", "qc_E"]

    def dosage_product(self, D: torch.Tensor, table: torch.Tensor, P: int) -> torch.Tensor:
        """torch.matmul(D, E * mask) (model.py:118-122) through the non-zero dosages of each row."""
        from . import functional as F_hip

        if self.dense_dosage_matmul:  # the caller saw dosage_overflow
            return F_hip.dense_dosage_matmul(D, table, P)
        plan = F_hip.DosagePlan(D, self.max_dosages_per_cell)
        self.dosage_overflow = plan.overflow if self.dosage_overflow is None else self.dosage_overflow | plan.overflow
        return F_hip.dosage_matmul(plan, table, P)

    def forward(
        self,
        D: torch.Tensor,
        qc: Optional[torch.Tensor],
        condition_values: Optional[Dict[str, torch.Tensor]] = None,
        n_particles: int = 1,
        sample_x: bool = True,
        live_variant: Optional[torch.Tensor] = None,
    ) -> Tuple[Dict[str, object], Dict[str, torch.Tensor]]:
        """model.py:59-144.  ``sample_x=False`` skips the x_good/x_bad draws (the training loss
        never reads them: loss_modules.py:123-143)."""
        n, device = D.shape[0], D.device
        cv = {} if condition_values is None else condition_values
        dists: Dict[str, object] = {}
        dists["p_z_basal"] = Normal(
            torch.zeros((n, self.n_latent), device=device), torch.ones((n, self.n_latent), device=device),
            validate_args=False)
        dists["p_E"] = Normal(self.p_E_loc, self.p_E_scale, validate_args=False)
        dists["p_mask"] = Bernoulli(logits=torch.logit(self.p_mask_probs), validate_args=False)
        dists["p_qc_E"] = Normal(self.p_qc_E_loc, self.p_qc_E_scale, validate_args=False)

        samples: Dict[str, torch.Tensor] = {}
        for k in self.get_var_keys():  # model.py:101-112
            value = cv.get(k)
            if value is not None:
                if value.dim() == 2:
                    value = value.unsqueeze(0).expand((n_particles, -1, -1))
                samples[k] = value
            else:
                samples[k] = dists[f"p_{k}"].sample((n_particles,))

        off = self.dosage_product(D, samples["E"] * samples["mask"], n_particles)
        if qc is None:
            raise NotImplementedError("qc=None (model.py:134-142) is not used by any caller of the reference")
        # matmul(ones(n, Q), qc_E) = sum of the Q rows for every cell; matmul(zeros(n, Q), qc_E) = 0
        shift = samples["qc_E"].sum(dim=1, keepdim=True).expand(-1, n, -1)
        if self.sum_mode:  # model.py:117-119
            z_good = samples["z_basal"] + off
            z_bad = z_good + shift
        else:  # model.py:120-122
            z_good = torch.cat([samples["z_basal"], off, torch.zeros_like(shift)], dim=-1)
            z_bad = torch.cat([samples["z_basal"], off, shift], dim=-1)
        lib = cv["library_size"]
        if live_variant is not None:
            # qc in {0,1}: a cell's likelihood weight sits on one variant only (loss_modules.py:142,
            # predictor.py:160-164); decode that one.  live_variant (n,): 0 = good, 1 = bad.
            sel = live_variant.reshape(1, -1, 1)
            dists["p_x_live"] = self.decoder(torch.where(sel != 0, z_bad, z_good), lib)
            return dists, samples
        dists["p_x_good"] = self.decoder(z_good, lib)
        dists["p_x_bad"] = self.decoder(z_bad, lib)
        if sample_x:  # model.py:131-132
            samples["x_good"] = dists["p_x_good"].sample()
            samples["x_bad"] = dists["p_x_bad"].sample()
        return dists, samples
