"""Distribution objects the guide / model hand back to callers.

The reference returns ``torch.distributions`` / pyro objects; callers use ``.mean``,
``.stddev``, ``.log_prob``, ``.rsample``, ``.sample`` and ``q_mask.probs``
(lightning_callbacks.py:73-80, predictor.py:158-177, loss_modules.py:134-161).
"""
from __future__ import annotations

import torch
from torch.distributions import Bernoulli
from torch.distributions.utils import clamp_probs

from . import functional as F_hip

MU_EPS = 1e-4
RATE_MIN = 1e-6


class GumbelSoftmaxBernoulliStraightThrough:
    """gumbel_softmax_bernoulli.py:5-17 over pyro RelaxedBernoulliStraightThrough (pyro-ppl
    1.9.1): relaxed-Bernoulli sample, hard-rounded with a straight-through gradient;
    ``log_prob`` is the *discrete* Bernoulli log-probability."""

    has_rsample = True

    def __init__(self, temperature: torch.Tensor, logits: torch.Tensor):
        self.temperature = temperature
        self.logits = logits

    @property
    def probs(self) -> torch.Tensor:
        return torch.sigmoid(self.logits)

    def rsample(self, sample_shape=torch.Size(), uniforms: torch.Tensor = None) -> torch.Tensor:
        shape = torch.Size(sample_shape) + self.logits.shape
        probs = clamp_probs(self.probs.expand(shape))
        if uniforms is None:
            uniforms = torch.rand(shape, dtype=probs.dtype, device=probs.device)
        u = clamp_probs(uniforms)
        y = (u.log() - (-u).log1p() + probs.log() - (-probs).log1p()) / self.temperature
        soft = clamp_probs(torch.sigmoid(y))
        return soft + (soft.round() - soft).detach()

    sample = rsample

    def log_prob(self, value: torch.Tensor) -> torch.Tensor:
        return Bernoulli(probs=self.probs, validate_args=False).log_prob(value)

    @property
    def mode(self):
        return (self.probs > 0.5).to(self.probs)

    @property
    def mean(self):
        return self.probs


class LibraryGammaPoisson:
    """GammaPoisson(concentration, rate) of mlp.py:160-166 kept in its pre-softmax form.

    ``log_prob(X).sum(-1)`` is what every caller computes; ``log_prob_sum`` does it in one HIP
    kernel (K1).  ``log_prob`` (per gene) and ``sample`` are provided for API compatibility.
    """

    def __init__(self, logits, library_size, log_concentration, P, n, batch_shape):
        self.logits = logits  # (P*n, G)
        self.library_size = library_size  # (n,) or (1,)
        self.log_concentration = log_concentration
        self.P, self.n = P, n
        self._batch_shape = tuple(batch_shape) + (logits.shape[-1],)

    # ---- fused path
    def log_prob_sum(self, X: torch.Tensor) -> torch.Tensor:
        """sum over genes of log p(X) -> batch_shape without the gene axis."""
        lib = self.library_size.expand(self.n) if self.library_size.numel() == 1 else self.library_size
        out = F_hip.nb_logprob(self.logits, X.reshape(self.n, -1), lib.contiguous(), self.log_concentration, self.P)
        return out.reshape(self._batch_shape[:-1])

    # ---- torch.distributions-like surface
    @property
    def concentration(self) -> torch.Tensor:
        return torch.exp(self.log_concentration)

    @property
    def mean(self) -> torch.Tensor:
        return self.concentration / self.rate

    @property
    def rate(self) -> torch.Tensor:
        lib = self.library_size.reshape(-1, 1)
        if lib.shape[0] == self.n and self.P > 1:
            lib = lib.repeat(self.P, 1)
        mu = (lib * torch.softmax(self.logits, dim=-1)).reshape(self._batch_shape)
        return torch.clamp(self.concentration / (mu + MU_EPS), min=RATE_MIN)

    def log_prob(self, value: torch.Tensor) -> torch.Tensor:
        c, rate = self.concentration, self.rate
        post = c + value
        log_beta = torch.lgamma(c) + torch.lgamma(value + 1) - torch.lgamma(post + 1)
        return -log_beta - post.log() + c * rate.log() - post * (1 + rate).log()

    def sample(self, sample_shape=torch.Size(), seed: int = None) -> torch.Tensor:
        """x ~ GammaPoisson, drawn by the HIP sampler kernel (K9).  The Philox key comes from
        ``seed`` or, like pyro's sampler, from torch's global generator state."""
        from . import ops

        if len(tuple(sample_shape)) != 0:
            raise NotImplementedError("LibraryGammaPoisson.sample: only sample_shape=() (all reference call sites)")
        if seed is None:
            seed = int(torch.randint(0, 2**62, (1,)).item())
        with torch.no_grad():
            consts = ops.nb_gene_consts(self.log_concentration.detach())
            _, x = ops.nb_sample_accumulate(
                self.logits.detach(), self.library_size.contiguous(), consts, self.P, self.n, seed=seed, return_samples=True)
        return x.reshape(self._batch_shape)
