"""Seeded synthetic Perturb-seq batches with the reference's batch layout.

Layout = what ``SCRNASeqTensorPerturbationDataset.__getitem__`` + ``default_collate``
hand to the training step (reference ``cradle_vae/data/utils/perturbation_dataset.py:85-89``,
``cradle_vae/data/norman/data_module.py:100-102,186-189``):

* ``X`` (N,G) raw integer counts stored as fp32, ``library_size`` (N,) = ``X.sum(1)``
* ``D`` (N,T) 0/1 dosages, at most two ones per row
* ``qc`` (N,Q) 1 = QC-fail
* ``cfX`` (N,G) integer-valued counterfactual, an all-zero row = "no counterfactual"
* ``D_obs_counts`` (T,), ``qc_obs_counts`` (Q,) train-set observation counts
* ``log_x_mean`` / ``log_x_std`` (G,) train-set statistics of ``log(x+1)``

Recipe: SURVEY.md section 8(d).  There is no network in this pipeline, so every
benchmark and test uses this generator instead of the Norman/Adamson/Replogle files.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch


@dataclass(frozen=True)
class Shape:
    """Problem shape of one ELBO step (names follow the reference's yaml keys)."""

    name: str
    n_cells: int  # data_module_kwargs.batch_size
    n_particles: int  # lightning_module_kwargs.n_particles
    n_phenos: int  # genes, from the data
    n_treatments: int  # guide columns, from the data
    n_latent: int  # model_kwargs.n_latent
    n_qc: int = 1
    decoder_n_hidden: int = 400
    basal_encoder_n_hidden: int = 400
    embedding_encoder_n_hidden: int = 400
    embedding_encoder_n_layers: int = 4
    decoder_n_layers: int = 1
    basal_encoder_n_layers: int = 1
    beta: float = 0.5
    mask_prior_prob: float = 0.01


# demo/cradle_vae_norman.yaml:16,28-34,39-45,50,54 ; G,T per BASELINE.md
NORMAN = Shape("norman", 512, 5, 5000, 105, 200, beta=0.5, mask_prior_prob=0.01)
# demo/cradle_vae_adamson.yaml (L=100, basal hidden 200, beta 0.1, p0 0.001); G,T stand-ins
ADAMSON = Shape("adamson", 512, 5, 5000, 90, 100, basal_encoder_n_hidden=200, beta=0.1, mask_prior_prob=0.001)
REPLOGLE = Shape("replogle", 512, 5, 8000, 2000, 100, basal_encoder_n_hidden=200, beta=0.1, mask_prior_prob=0.001)
SCALEUP = Shape("scaleup", 512, 5, 8000, 10000, 100, basal_encoder_n_hidden=200, beta=0.1, mask_prior_prob=0.001)
TINY = Shape("tiny", 8, 3, 32, 6, 4, decoder_n_hidden=8, basal_encoder_n_hidden=8, embedding_encoder_n_hidden=8)
MID = Shape("mid", 64, 5, 300, 12, 16, decoder_n_hidden=32, basal_encoder_n_hidden=32, embedding_encoder_n_hidden=32)

SHAPES = {s.name: s for s in (NORMAN, ADAMSON, REPLOGLE, SCALEUP, TINY, MID)}


def make_batch(
    shape: Shape,
    seed: int = 0,
    device: str = "cpu",
    n_train: Optional[int] = None,
    dtype: torch.dtype = torch.float32,
) -> Dict[str, torch.Tensor]:
    """One batch + the train-set statistics the step needs.  Deterministic in ``seed``
    (always generated on CPU with a seeded generator, then moved)."""
    g = torch.Generator().manual_seed(seed)
    N, G, T, Q = shape.n_cells, shape.n_phenos, shape.n_treatments, shape.n_qc
    n_train = n_train or 2048

    lam = torch.exp(-1.0 + 1.5 * torch.randn(G, generator=g, dtype=torch.float64))
    lam = lam * (1e4 * (G / 5000.0) / lam.sum())  # mean library ~ 1e4 * G/5000

    def counts(n):
        s = torch.exp(0.3 * torch.randn(n, 1, generator=g, dtype=torch.float64))
        return torch.poisson(lam.unsqueeze(0) * s, generator=g)

    X = counts(N)
    cfX = counts(N)
    cfX[torch.rand(N, generator=g) < 1.0 / 3.0] = 0.0
    qc = (torch.rand(N, Q, generator=g) < 0.2).to(torch.float64)

    def dosages(n):
        D = torch.zeros(n, T, dtype=torch.float64)
        r = torch.rand(n, generator=g)
        t1 = torch.randint(0, T, (n,), generator=g)
        t2 = torch.randint(0, T, (n,), generator=g)
        rows = torch.arange(n)
        one = r >= 0.1
        two = r >= 0.7
        D[rows[one], t1[one]] = 1.0
        D[rows[two], t2[two]] = 1.0
        return D

    D = dosages(N)
    # train-set statistics from a larger synthetic "train set" of the same law
    D_train = dosages(n_train)
    D_obs_counts = (D_train != 0).sum(0) + (D != 0).sum(0)  # >= the batch's own counts
    D_obs_counts = torch.clamp(D_obs_counts, min=1)
    qc_obs_counts = torch.clamp(((torch.rand(n_train, Q, generator=g) < 0.2).sum(0) + (qc != 0).sum(0)), min=1)
    # statistics of log(x+1) over a synthetic train set of the same law (plus the batch and
    # its counterfactuals, which the reference also draws from the train split); a gene that
    # is zero everywhere keeps std 0: the reference lets nan_to_num act on it
    logx = torch.log(torch.cat([counts(n_train), X, cfX]) + 1)
    log_x_mean = logx.mean(0)
    log_x_std = logx.std(0)

    out = dict(
        X=X,
        cfX=cfX,
        D=D,
        qc=qc,
        library_size=X.sum(1),
        D_obs_counts=D_obs_counts.to(torch.float64),
        qc_obs_counts=qc_obs_counts.to(torch.float64),
        log_x_mean=log_x_mean,
        log_x_std=log_x_std,
        idx=torch.arange(N),
    )
    res = {}
    for k, v in out.items():
        res[k] = v.to(device) if k == "idx" else v.to(dtype).to(device)
    return res


def make_noise(shape: Shape, seed: int, device: str = "cpu", dtype: torch.dtype = torch.float32):
    """Injected noise for one guide pass, in the reference's RNG consumption order
    (mask uniforms, E, qc_E, z_basal: ``correlated_normal_guide.py:139,157,166,202``)."""
    g = torch.Generator().manual_seed(seed)
    P, N, T, L, Q = shape.n_particles, shape.n_cells, shape.n_treatments, shape.n_latent, shape.n_qc
    noise = dict(
        u_mask=torch.rand(P, T, L, generator=g, dtype=torch.float32).to(torch.float64),
        eps_E=torch.randn(P, T, L, generator=g, dtype=torch.float64),
        eps_qc_E=torch.randn(P, Q, L, generator=g, dtype=torch.float64),
        eps_z=torch.randn(P, N, L, generator=g, dtype=torch.float64),
    )
    return {k: v.to(dtype).to(device) for k, v in noise.items()}
