"""MLP blocks of the reference (cradle_vae/models/utils/mlp.py) on the HIP GEMM.

Module / parameter names reproduce the reference's state-dict keys
(``mlp.layers.{i}.0.{weight,bias}``, ``mean_encoder.*``, ``log_var_encoder.*``,
``normalized_mean_decoder.0.*``, ``log_concentration``) so that checkpoints interchange.
Only the variants the shipped configs select are provided.
"""
from __future__ import annotations

from typing import Literal

import torch
from torch import nn

from . import functional as F_hip
from . import ops
from .distributions import LibraryGammaPoisson

LIKELIHOOD_KEY_DTYPE = Literal["normal", "poisson", "library_nb"]
VAR_EPS = 1e-4


def nan_to_num_ref(t: torch.Tensor) -> torch.Tensor:
    return torch.nan_to_num(t, nan=1e-6, posinf=1e6, neginf=1e-6)


class ResidualMLP(nn.Module):
    """``ResidualMLP`` with ``last_layer_residual=True`` (mlp.py:83-119): LeakyReLU after every
    Linear, residual connection on every layer but the first."""

    def __init__(self, n_input: int, n_output: int, n_layers: int, n_hidden: int):
        super().__init__()
        dims = [n_input] + n_layers * [n_hidden] + [n_output]  # mlp.py:58
        assert n_output == n_hidden  # mlp.py:109
        self.layers = nn.ModuleList(
            [nn.Sequential(nn.Linear(dims[i - 1], dims[i]), nn.LeakyReLU()) for i in range(1, len(dims))]
        )

    def linears(self):
        return [layer[0] for layer in self.layers]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        for i, lin in enumerate(self.linears()):
            x = F_hip.linear(x, lin.weight, lin.bias, act=True, residual=None if i == 0 else x)
        return x


class GaussianLikelihoodResidualMLP(nn.Module):
    """mlp.py:217-243 + BaseGaussianLikelihoodMLP.forward :122-142 -> torch Normal."""

    def __init__(self, n_input: int, n_output: int, n_layers: int, n_hidden: int, var_eps: float = VAR_EPS):
        super().__init__()
        self.mlp = ResidualMLP(n_input, n_hidden, n_layers - 1, n_hidden)
        self.mean_encoder = nn.Linear(n_hidden, n_output)
        self.log_var_encoder = nn.Linear(n_hidden, n_output)
        self.var_eps = var_eps

    def mean_var(self, x: torch.Tensor):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        z = self.mlp(x2)
        mean = F_hip.linear(z, self.mean_encoder.weight, self.mean_encoder.bias)
        var = torch.exp(F_hip.linear(z, self.log_var_encoder.weight, self.log_var_encoder.bias)) + self.var_eps
        mean = nan_to_num_ref(mean.reshape(*shape[:-1], -1))
        var = nan_to_num_ref(torch.abs(var.reshape(*shape[:-1], -1)))
        return mean, var

    def forward(self, x: torch.Tensor) -> torch.distributions.Normal:
        mean, var = self.mean_var(x)
        return torch.distributions.Normal(mean, var.sqrt(), validate_args=False)

    def forward_from_hidden(self, h: torch.Tensor, first_layer_done: bool = True) -> torch.distributions.Normal:
        """The rest of forward() given the first trunk layer's output h (..., H) (the caller evaluated that layer
        in a cheaper form: guide._embedding_dist)."""
        shape = h.shape
        z = h.reshape(-1, shape[-1])
        for lin in self.mlp.linears()[1 if first_layer_done else 0:]:
            z = F_hip.linear(z, lin.weight, lin.bias, act=True, residual=z)
        mean = F_hip.linear(z, self.mean_encoder.weight, self.mean_encoder.bias)
        var = torch.exp(F_hip.linear(z, self.log_var_encoder.weight, self.log_var_encoder.bias)) + self.var_eps
        mean = nan_to_num_ref(mean.reshape(*shape[:-1], -1))
        var = nan_to_num_ref(torch.abs(var.reshape(*shape[:-1], -1)))
        return torch.distributions.Normal(mean, var.sqrt(), validate_args=False)

    @torch.no_grad()
    def sample_from_hidden(self, h: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
        """forward_from_hidden(h).rsample() with the noise given, without the Normal in between: the two heads' raw
        outputs go through ONE launch (ops.gauss_sample) instead of nine element-wise ones (inference calls)."""
        shape = h.shape
        z = h.reshape(-1, shape[-1])
        for lin in self.mlp.linears()[1:]:
            z = F_hip.linear(z, lin.weight, lin.bias, act=True, residual=z)
        mean = F_hip.linear(z, self.mean_encoder.weight, self.mean_encoder.bias)
        lv = F_hip.linear(z, self.log_var_encoder.weight, self.log_var_encoder.bias)
        return ops.gauss_sample(mean, lv, eps).reshape(*shape[:-1], -1)

    # ---- inference form for inputs [x_shared | tail] whose first block is the same for every particle
    # (the basal encoder reads [log-standardised X | latent offset | qc offset], correlated_normal_guide.py:
    # 190-204): x W_x^T is computed once per cell and enters the (P*n)-row product of the tail as a
    # row-bias of the GEMM epilogue.  Equal to forward(cat([x.expand(P), tail])) up to fp reassociation.
    @torch.no_grad()
    def shared_input_part(self, x_shared: torch.Tensor) -> torch.Tensor:
        lin0 = self.mlp.linears()[0]
        return ops.gemm(ops.NT, x_shared.contiguous(), lin0.weight[:, : x_shared.shape[-1]])  # (n, H)

    @torch.no_grad()
    def forward_shared(self, xw: torch.Tensor, tail: torch.Tensor) -> torch.distributions.Normal:
        P, n, r = tail.shape
        lins = self.mlp.linears()
        k0 = lins[0].weight.shape[1] - r
        h = ops.gemm(ops.NT, tail.reshape(P * n, r).contiguous(), lins[0].weight[:, k0:], bias=lins[0].bias, act=True,
                     rowbias=xw, rowbias_mode=2, rowbias_n=n)
        for lin in lins[1:]:
            h = F_hip.linear(h, lin.weight, lin.bias, act=True, residual=h)
        mean = F_hip.linear(h, self.mean_encoder.weight, self.mean_encoder.bias)
        var = torch.exp(F_hip.linear(h, self.log_var_encoder.weight, self.log_var_encoder.bias)) + self.var_eps
        mean = nan_to_num_ref(mean.reshape(P, n, -1))
        var = nan_to_num_ref(torch.abs(var.reshape(P, n, -1)))
        return torch.distributions.Normal(mean, var.sqrt(), validate_args=False)


class LibraryGammaPoissonSharedConcentrationResidualMLP(nn.Module):
    """mlp.py:286-317 + forward :145-168.  Returns a ``LibraryGammaPoisson`` that keeps the
    pre-softmax logits so that ``log_prob`` runs the fused HIP kernel."""

    def __init__(self, n_input: int, n_output: int, n_layers: int, n_hidden: int):
        super().__init__()
        self.mlp = ResidualMLP(n_input, n_hidden, n_layers - 1, n_hidden)
        # the Softmax module has no parameters; it keeps the reference's key "normalized_mean_decoder.0"
        self.normalized_mean_decoder = nn.Sequential(nn.Linear(n_hidden, n_output), nn.Softmax(dim=-1))
        self.log_concentration = nn.Parameter(torch.zeros((n_output,)))

    def forward(self, x: torch.Tensor, library_size: torch.Tensor) -> LibraryGammaPoisson:
        if x.dim() == 3:
            P, n, _ = x.shape
        else:
            P, n = 1, x.shape[0]
        h = self.mlp(x.reshape(P * n, -1))
        lin = self.normalized_mean_decoder[0]
        logits = F_hip.linear(h, lin.weight, lin.bias)
        batch_shape = (P, n) if x.dim() == 3 else (n,)
        return LibraryGammaPoisson(logits, library_size.reshape(-1), self.log_concentration, P, n, batch_shape)


def get_likelihood_mlp(likelihood_key: LIKELIHOOD_KEY_DTYPE, n_input: int, n_output: int, n_layers: int,
                       n_hidden: int, use_batch_norm: bool = False, activation_fn=nn.LeakyReLU) -> nn.Module:
    """mlp.py:12-40 for the keys the shipped configs use."""
    if use_batch_norm or activation_fn is not nn.LeakyReLU:
        raise NotImplementedError("cradle_vae_amd: only LeakyReLU MLPs without batch norm are on the HIP path")
    if likelihood_key == "normal":
        return GaussianLikelihoodResidualMLP(n_input, n_output, n_layers, n_hidden)
    if likelihood_key == "library_nb":
        return LibraryGammaPoissonSharedConcentrationResidualMLP(n_input, n_output, n_layers, n_hidden)
    raise NotImplementedError(
        f"likelihood_key={likelihood_key!r}: no shipped CRADLE-VAE config selects it; out of the HIP path's scope"
    )
