"""Variational guide q(M) q(E|M) q(qc_E) q(z_basal | x, M, E, qc) -- mirror of the reference's
``cradle_vae_CorrelatedNormalGuide`` (cradle_vae/models/cradle_vae/guides/
correlated_normal_guide.py:18-215) on the HIP kernels."""
from __future__ import annotations

import contextlib
import os

from dataclasses import dataclass
from typing import Dict, Literal, Optional, Tuple

import torch
from torch import nn

from . import functional as F_hip
from . import ops
from .distributions import GumbelSoftmaxBernoulliStraightThrough
from .mlp import get_likelihood_mlp

# (particle, treatment) rows of the embedding encoder evaluated at once when no gradient is needed: with
# thousands of treatments and a hundred particles the rows run into the millions (2 GB per activation at
# H = 400), so the plated variables are then evaluated in slabs of treatments (rows are independent)
MAX_PLATED_ROWS = 1 << 17
MAX_ONE_HOT_BUFFER = 4096


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
        # kept for state-dict compatibility (correlated_normal_guide.py:64) while it is small; the HIP path never
        # multiplies by it (the one-hot block of the first layer is a column lookup), and at 10 k treatments it
        # would be 400 MB of identity matrix: beyond MAX_ONE_HOT_BUFFER treatments the key is dropped on load
        if n_treatments <= MAX_ONE_HOT_BUFFER:
            self.register_buffer("treatment_one_hot", torch.eye(n_treatments))
        self.normalization_module = LogStandardizationModule(x_normalization_stats)
        self.register_buffer("qc_encoder_input", torch.ones((n_qc, n_latent)))
        self.qc_encoder = get_likelihood_mlp(
            "normal", n_latent, n_latent, embedding_encoder_n_layers, embedding_encoder_n_hidden)
        self.z_basal_encoder = get_likelihood_mlp(
            "normal", n_phenos + 2 * n_latent, n_latent, basal_encoder_n_layers, basal_encoder_n_hidden)
        self.register_buffer("gs_temperature", gs_temperature * torch.ones((1,)))
        self.var_eps = 1e-4
        # matmul(D, .) runs through the non-zero dosages of each cell (<= this many; a row with more raises the
        # `dosage_overflow` flag that the callers check once per call and answer with dense_dosage_matmul = True)
        self.max_dosages_per_cell = 8
        self.dense_dosage_matmul = False
        self.dosage_overflow = None
        self._qc_dist_cache = None
        self._temp_cache = None

    def get_var_keys(self):
        return ["z_basal", "E", "mask", "qc_E"]

    def _temperature_value(self) -> float:
        """gs_temperature as a host float, read from the device once per value (a buffer: load_state_dict may change it)."""
        key = (self.gs_temperature.data_ptr(), self.gs_temperature._version)
        if self._temp_cache is None or self._temp_cache[0] != key:
            self._temp_cache = (key, float(self.gs_temperature))
        return self._temp_cache[1]

    @contextlib.contextmanager
    def frozen_qc_dist(self):
        """Inference loops that call the guide many times with unchanged parameters (the ATE estimator: once per
        particle chunk): q(qc_E) does not depend on the call -- its encoder's input is the constant ones(Q, L)
        (correlated_normal_guide.py:162-166) -- so its mean and scale are evaluated once, on the Q rows, and reused
        inside the block; the reference pushes P identical rows through the encoder in every call.  Same values:
        a row's result does not depend on how many copies of it the product holds."""
        with torch.no_grad():
            q = self.qc_encoder(self.qc_encoder_input.unsqueeze(0))
            self._qc_dist_cache = (q.mean[0].contiguous(), q.stddev[0].contiguous())
        try:
            yield
        finally:
            self._qc_dist_cache = None

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        if self.n_treatments > MAX_ONE_HOT_BUFFER:  # a reference checkpoint carries eye(T): not needed here
            state_dict.pop(prefix + "treatment_one_hot", None)
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

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
        generator: Optional[torch.Generator] = None,
    ) -> Tuple[Dict[str, object], Dict[str, torch.Tensor]]:
        """correlated_normal_guide.py:102-209.  ``noise`` (keys u_mask, eps_E, eps_qc_E, eps_z)
        injects the random draws; missing keys are drawn with torch's generator.
        ``need_cf_basal=False`` (inference only) skips ``q_z_cf_basal``, which the predictor never reads.
        ``generator`` draws whatever ``noise`` does not supply (default: torch's global generator)."""
        cv = {} if condition_values is None else condition_values
        noise = {} if noise is None else noise
        P = n_particles
        dists: Dict[str, object] = {}
        samples: Dict[str, torch.Tensor] = {}

        def rsample(dist, key):
            eps = noise.get(key)
            if eps is None:
                eps = torch.randn(dist.mean.shape, device=dist.mean.device, dtype=dist.mean.dtype, generator=generator)
            return dist.mean + dist.stddev * eps

        def uniforms(t0, t1):
            if "u_mask" in noise:
                return noise["u_mask"][:, t0:t1]
            return torch.rand((P, t1 - t0, self.n_latent), device=logits.device, generator=generator)

        logits = self.sanitize_tensor(self.param_dict["q_mask_logits"])
        dists["q_mask"] = GumbelSoftmaxBernoulliStraightThrough(temperature=self.gs_temperature, logits=logits)
        T, L = self.n_treatments, self.n_latent
        slab = T
        if not torch.is_grad_enabled() and logits.is_cuda and P * T > MAX_PLATED_ROWS:
            slab = max(1, MAX_PLATED_ROWS // P)
        # inference calls on the GPU draw q(M) and q(E|M) through one launch each (ops.mask_sample / gauss_sample: the
        # fused step's own expressions) instead of ~25 element-wise ones; CRADLE_FUSED_DRAWS=0 keeps the torch form
        fused_draws = (not torch.is_grad_enabled() and logits.is_cuda and logits.dtype == torch.float32
                       and os.environ.get("CRADLE_FUSED_DRAWS", "1") != "0")
        if fused_draws:
            slab = min(slab, T)
            whole = slab >= T
            mask = cv["mask"] if "mask" in cv else (None if whole else torch.empty((P, T, L), device=logits.device))
            E = cv["E"] if "E" in cv else (None if whole else torch.empty((P, T, L), device=logits.device))
            temp = self._temperature_value()
            for t0 in range(0, T, slab):
                t1 = min(T, t0 + slab)
                if "mask" not in cv:
                    m_slab = ops.mask_sample(logits[t0:t1], uniforms(t0, t1), temp)
                    if whole:
                        mask = m_slab
                    else:
                        mask[:, t0:t1] = m_slab
                if "E" not in cv:
                    eps = noise["eps_E"][:, t0:t1] if "eps_E" in noise else torch.randn(
                        (P, t1 - t0, L), device=logits.device, generator=generator)
                    lins = self.embedding_encoder.mlp.linears()
                    h = F_hip.onehot_linear(mask[:, t0:t1], lins[0].weight, lins[0].bias, t0)
                    e_slab = self.embedding_encoder.sample_from_hidden(h, eps)
                    if whole:
                        E = e_slab
                    else:
                        E[:, t0:t1] = e_slab
            samples["mask"], samples["E"] = mask, E  # (no dists["q_E"]: no caller of the inference path reads it)
        elif slab >= T:
            if "mask" not in cv:
                samples["mask"] = dists["q_mask"].rsample((P,), uniforms=uniforms(0, T))
            else:
                samples["mask"] = cv["mask"]
            dists["q_E"] = self._embedding_dist(samples["mask"], 0)
            samples["E"] = cv["E"] if "E" in cv else rsample(dists["q_E"], "eps_E")
        else:
            # inference with many treatments: q(M), q(E|M) in slabs of treatments (every (particle, treatment) row
            # is independent: correlated_normal_guide.py:132-160 row by row); only the samples are kept, the
            # Normal's parameters of all T would be two more (P, T, L) tensors
            mask = torch.empty((P, T, L), device=logits.device) if "mask" not in cv else cv["mask"]
            E = torch.empty((P, T, L), device=logits.device) if "E" not in cv else cv["E"]
            for t0 in range(0, T, slab):
                t1 = min(T, t0 + slab)
                if "mask" not in cv:
                    q = GumbelSoftmaxBernoulliStraightThrough(temperature=self.gs_temperature, logits=logits[t0:t1])
                    mask[:, t0:t1] = q.rsample((P,), uniforms=uniforms(t0, t1))
                if "E" not in cv:
                    qE = self._embedding_dist(mask[:, t0:t1], t0)
                    eps = noise["eps_E"][:, t0:t1] if "eps_E" in noise else torch.randn(
                        qE.mean.shape, device=logits.device, generator=generator)
                    E[:, t0:t1] = qE.mean + qE.stddev * eps
            samples["mask"], samples["E"] = mask, E  # (no dists["q_E"] in this mode: no caller of the inference path reads it)

        if self._qc_dist_cache is not None and not torch.is_grad_enabled():
            qm, qs = self._qc_dist_cache  # (Q, L) each: the encoder's input is the same ones(Q, L) for every particle
            dists["q_qc_E"] = torch.distributions.Normal(  # (validate_args: its check reads the device -- a host sync per call)
                qm.unsqueeze(0).expand(P, -1, -1), qs.unsqueeze(0).expand(P, -1, -1), validate_args=False)
        else:
            dists["q_qc_E"] = self.qc_encoder(self.qc_encoder_input.unsqueeze(0).expand(P, -1, -1))
        samples["qc_E"] = cv["qc_E"] if "qc_E" in cv else rsample(dists["q_qc_E"], "eps_qc_E")

        if X is not None and D is not None and qc is not None:
            xn = self.normalization_module(X)
            latent_offset = self.dosage_product(D, samples["mask"] * samples["E"], P)
            latent_qc_offset = self.qc_product(qc, samples["qc_E"], P)
            latent_qc_cf_offset = self.qc_product(1 - qc, samples["qc_E"], P)
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

    # ------------------------------------------------------------------ pieces
    def _embedding_dist(self, mask: torch.Tensor, t0: int) -> torch.distributions.Normal:
        """q(E | M) for the treatments t0 .. t0 + mask.shape[1] - 1 (correlated_normal_guide.py:147-157).  The
        reference feeds [mask | one_hot(t)] to the encoder: a (P T) x (L + T) input and a GEMM against eye(T); here
        the one-hot block of the first layer is a lookup of its weight columns (exact)."""
        enc = self.embedding_encoder
        lins = enc.mlp.linears()
        h = F_hip.onehot_linear(mask, lins[0].weight, lins[0].bias, t0)
        return enc.forward_from_hidden(h, first_layer_done=True)

    def dosage_product(self, D: torch.Tensor, table: torch.Tensor, P: int) -> torch.Tensor:
        """torch.matmul(D, table) for dosages D (N, T) and table (P, T, L) (correlated_normal_guide.py:186-188)
        through the <= max_dosages_per_cell non-zeros of each row."""
        if self.dense_dosage_matmul:  # the caller saw dosage_overflow: a cell with more dosages than planned for
            return F_hip.dense_dosage_matmul(D, table, P)
        plan = F_hip.DosagePlan(D, self.max_dosages_per_cell)
        self.dosage_overflow = plan.overflow if self.dosage_overflow is None else self.dosage_overflow | plan.overflow
        return F_hip.dosage_matmul(plan, table, P)

    @staticmethod
    def qc_product(qc: torch.Tensor, table: torch.Tensor, P: int) -> torch.Tensor:
        """torch.matmul(qc, qc_E) for qc (N, Q), Q small (correlated_normal_guide.py:189-194): every column counts."""
        plan = F_hip.DosagePlan(qc, qc.shape[1])
        return F_hip.dosage_matmul(plan, table, P)
