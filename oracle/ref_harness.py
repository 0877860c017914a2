"""Runs the reference's own hot-path modules (read from /root/reference) on CPU.

TEST INFRASTRUCTURE ONLY, and only usable in the build container: the GPU box has no
/root/reference.  Used by ``oracle/make_golden.py`` to produce ``tests/golden/*.npz`` (the
fixtures ``tests/test_oracle_golden.py`` then pins the oracle to, wherever it runs).

The reference imports packages that are not installed anywhere in this pipeline
(pyro, pytorch_lightning, anndata, scanpy, rapids_singlecell, cupy, wandb).  They are
replaced in ``sys.modules`` before import:

* ``pyro.distributions``: a torch restatement of the two classes the hot path uses,
  ``GammaPoisson`` and ``RelaxedBernoulliStraightThrough`` (pyro-ppl 1.9.1 published
  algorithms) and a dummy ``Delta``.  Everything that flows through these two classes is
  therefore "parity unpinned by the reference" -- see DESIGN.md.
* the rest: empty modules / trivial base classes; nothing numeric.
* ``cradle_vae.data`` is installed as a bare namespace whose ``__path__`` points at the real
  directory, so its ``__init__`` (scanpy/rapids/...) never runs while
  ``cradle_vae.data.utils.perturbation_datamodule`` still loads.

All numeric reference code (model.py, correlated_normal_guide.py, loss_modules.py, mlp.py,
normalization.py, predictor.py, average_treatment_effects.py) executes verbatim.

Random draws are replaced by injected tensors: ``torch.rand``, ``torch.normal`` and
``torch.distributions.normal._standard_normal`` are patched inside ``injected_noise`` to pop
from a queue (asserting the requested shape).
"""
from __future__ import annotations

import contextlib
import os
import sys
import types
from typing import Dict, List, Tuple

import torch

REFERENCE_ROOT = os.environ.get("CRADLE_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "cradle_vae", "models"))


# --------------------------------------------------------------------------------------
# stubs
# --------------------------------------------------------------------------------------
class _GammaPoisson:
    """pyro.distributions.GammaPoisson (pyro-ppl 1.9.1, conjugate.py) restated."""

    SAMPLE_MODE = "mean"  # "mean": deterministic stand-in; "random": Gamma then Poisson

    def __init__(self, concentration, rate, validate_args=None):
        self.concentration, self.rate = torch.distributions.utils.broadcast_all(concentration, rate)
        self._validate_args = True if validate_args is None else validate_args
        if self._validate_args:
            if not (self.concentration > 0).all() or not (self.rate > 0).all():
                raise ValueError("GammaPoisson: parameters must be positive")

    @property
    def mean(self):
        return self.concentration / self.rate

    def sample(self, sample_shape=torch.Size()):
        with torch.no_grad():
            if self.SAMPLE_MODE == "mean":
                return (self.concentration / self.rate).expand(
                    torch.Size(sample_shape) + self.concentration.shape
                ).clone()
            rate = torch.distributions.Gamma(self.concentration, self.rate).sample(sample_shape)
            return torch.poisson(rate)

    def log_prob(self, value):
        if self._validate_args:
            if not ((value >= 0) & (value % 1 == 0)).all():
                raise ValueError("GammaPoisson.log_prob: value outside the support")
        post_value = self.concentration + value
        log_beta = self.concentration.lgamma() + (value + 1).lgamma() - (post_value + 1).lgamma()
        return (
            -log_beta
            - post_value.log()
            + self.concentration * self.rate.log()
            - post_value * (1 + self.rate).log()
        )


class _RelaxedBernoulliStraightThrough(torch.distributions.RelaxedBernoulli):
    """pyro.distributions.RelaxedBernoulliStraightThrough (pyro-ppl 1.9.1,
    relaxed_straight_through.py) restated: soft sample -> clamp_probs -> round, identity
    backward."""

    def rsample(self, sample_shape=torch.Size()):
        soft = super().rsample(sample_shape)
        soft = torch.distributions.utils.clamp_probs(soft)
        hard = soft.round()
        return soft + (hard - soft).detach()


def _install_stubs():
    if "pyro" in sys.modules and getattr(sys.modules["pyro"], "_cradle_stub", False):
        return

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    dist = mod(
        "pyro.distributions",
        GammaPoisson=_GammaPoisson,
        RelaxedBernoulliStraightThrough=_RelaxedBernoulliStraightThrough,
        Delta=type("Delta", (), {}),
    )
    mod("pyro", distributions=dist, _cradle_stub=True)

    class _AnnData:
        """Just enough of anndata.AnnData for the reference's analysis helpers: X / obs / var / uns,
        len(), .shape and row selection by boolean mask or integer positions."""

        def __init__(self, X=None, obs=None, var=None, uns=None):
            self.X, self.obs, self.var, self.uns = X, obs, var, ({} if uns is None else uns)

        def __len__(self):
            return self.X.shape[0]

        @property
        def shape(self):
            return self.X.shape

        def __getitem__(self, idx):
            import numpy as np

            idx = np.asarray(idx)
            obs = self.obs[idx] if idx.dtype == bool else self.obs.iloc[idx]
            return _AnnData(X=self.X[idx], obs=obs, var=self.var, uns=dict(self.uns))

    mod("anndata", AnnData=_AnnData)
    types_mod = mod("pytorch_lightning.utilities.types", STEP_OUTPUT=object)
    util = mod("pytorch_lightning.utilities", types=types_mod)

    class _LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

    mod(
        "pytorch_lightning",
        LightningModule=_LightningModule,
        LightningDataModule=object,
        Callback=object,
        utilities=util,
    )
    for name in ("scanpy", "rapids_singlecell", "cupy", "wandb"):
        mod(name)

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import cradle_vae  # the reference's (empty) top-level package

    assert os.path.realpath(cradle_vae.__path__[0]).startswith(os.path.realpath(REFERENCE_ROOT)), (
        "a different cradle_vae package is already imported"
    )
    data_pkg = types.ModuleType("cradle_vae.data")
    data_pkg.__path__ = [os.path.join(REFERENCE_ROOT, "cradle_vae", "data")]
    sys.modules["cradle_vae.data"] = data_pkg
    cradle_vae.data = data_pkg


def load_reference():
    """Returns the reference's ``cradle_vae.models`` module and a few helpers."""
    if not reference_available():
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT}")
    _install_stubs()
    import cradle_vae.models as models
    from cradle_vae.analysis.average_treatment_effects import estimate_model_average_treatment_effect
    from cradle_vae.data.utils.perturbation_datamodule import ObservationNormalizationStatistics

    return types.SimpleNamespace(
        models=models,
        estimate_model_average_treatment_effect=estimate_model_average_treatment_effect,
        ObservationNormalizationStatistics=ObservationNormalizationStatistics,
        GammaPoisson=_GammaPoisson,
    )


# --------------------------------------------------------------------------------------
# noise injection
# --------------------------------------------------------------------------------------
@contextlib.contextmanager
def injected_noise(queue: List[Tuple[str, torch.Tensor]]):
    """``queue``: [("uniform"|"normal", tensor), ...] consumed in call order.  A ``None``
    tensor means "draw for real" (used for calls whose values do not matter)."""
    import torch.distributions.normal as tdn

    real_rand, real_normal, real_std = torch.rand, torch.normal, tdn._standard_normal
    q = list(queue)

    def pop(kind, shape):
        assert q, f"noise queue exhausted at a {kind} draw of shape {tuple(shape)}"
        k, t = q.pop(0)
        assert k == kind, f"expected a {k} draw, reference asked for {kind}"
        if t is not None:
            assert tuple(t.shape) == tuple(shape), f"{kind}: queue {tuple(t.shape)} vs request {tuple(shape)}"
        return t

    def rand(*size, **kw):
        shape = size[0] if len(size) == 1 and not isinstance(size[0], int) else size
        t = pop("uniform", shape)
        return real_rand(*size, **kw) if t is None else t.to(kw.get("dtype", t.dtype))

    def normal(mean, std, **kw):
        t = pop("normal", mean.shape)
        return real_normal(mean, std, **kw) if t is None else mean + std * t

    def standard_normal(shape, dtype, device):
        t = pop("normal", shape)
        return real_std(shape, dtype, device) if t is None else t.to(dtype)

    torch.rand, torch.normal, tdn._standard_normal = rand, normal, standard_normal
    try:
        yield q
    finally:
        torch.rand, torch.normal, tdn._standard_normal = real_rand, real_normal, real_std


@contextlib.contextmanager
def default_dtype(dtype):
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        yield
    finally:
        torch.set_default_dtype(old)


def guide_noise_queue(noise: Dict[str, torch.Tensor], with_z: bool = True):
    q = [("uniform", noise["u_mask"]), ("normal", noise["eps_E"]), ("normal", noise["eps_qc_E"])]
    if with_z:
        q.append(("normal", noise["eps_z"]))
    return q


# --------------------------------------------------------------------------------------
# building the reference modules from a flat state dict
# --------------------------------------------------------------------------------------
def build_reference_loss_module(ref, params: Dict[str, torch.Tensor], cfg: Dict, dtype=torch.float32):
    """cfg keys: n_latent, n_treatments, n_phenos, n_qc, decoder_n_layers, decoder_n_hidden,
    basal_encoder_n_layers, basal_encoder_n_hidden, embedding_encoder_n_layers,
    embedding_encoder_n_hidden, mask_prior_prob, embedding_prior_scale, beta, alpha."""
    stats = ref.ObservationNormalizationStatistics(
        log_x_mean=params["guide.normalization_module.log_mean"].clone(),
        log_x_std=params["guide.normalization_module.log_scale"].clone(),
    )
    model = ref.models.cradle_vae_Model(
        n_latent=cfg["n_latent"],
        n_treatments=cfg["n_treatments"],
        n_phenos=cfg["n_phenos"],
        n_qc=cfg["n_qc"],
        mask_prior_prob=cfg.get("mask_prior_prob", 0.01),
        embedding_prior_scale=cfg.get("embedding_prior_scale", 1.0),
        likelihood_key="library_nb",
        decoder_n_layers=cfg.get("decoder_n_layers", 1),
        decoder_n_hidden=cfg["decoder_n_hidden"],
        sum_mode=False,
    )
    guide = ref.models.cradle_vae_CorrelatedNormalGuide(
        n_latent=cfg["n_latent"],
        n_treatments=cfg["n_treatments"],
        n_phenos=cfg["n_phenos"],
        n_qc=cfg["n_qc"],
        basal_encoder_n_layers=cfg.get("basal_encoder_n_layers", 1),
        basal_encoder_n_hidden=cfg["basal_encoder_n_hidden"],
        basal_encoder_input_normalization="log_standardize",
        embedding_encoder_n_layers=cfg.get("embedding_encoder_n_layers", 4),
        embedding_encoder_n_hidden=cfg["embedding_encoder_n_hidden"],
        x_normalization_stats=stats,
    )
    loss_module = ref.models.cradle_vae_ELBOLossModule(
        model=model, guide=guide, beta=cfg.get("beta", 1), alpha=cfg.get("alpha", None)
    )
    loss_module = loss_module.to(dtype)
    missing, unexpected = loss_module.load_state_dict({k: v.to(dtype) for k, v in params.items()}, strict=True)
    assert not missing and not unexpected
    return loss_module


def reference_guide_samples(loss_module, batch, noise, n_particles):
    """The reference guide's own draws for the main pass (correlated_normal_guide.py:102-209) with
    the noise injected: mask (hard 0/1), E, qc_E, z_basal -- what "exact mask gathers" is pinned to."""
    q = guide_noise_queue(noise)
    with torch.no_grad(), default_dtype(batch["X"].dtype), injected_noise(q) as rest:
        _, samples = loss_module.guide(X=batch["X"], D=batch["D"], qc=batch["qc"], n_particles=n_particles)
        assert not rest, "reference consumed less noise than queued"
    return {k: samples[k].detach().clone() for k in ("mask", "E", "qc_E", "z_basal")}


def reference_loss_and_grads(loss_module, batch, noise, cf_noise, n_particles):
    """loss_module.loss(...) + backward, with both guide passes' noise injected."""
    for p in loss_module.parameters():
        p.grad = None
    q = guide_noise_queue(noise) + guide_noise_queue(cf_noise)
    # the reference allocates with the default dtype (model.py:121-122), so an fp64 run of it
    # needs the default dtype switched for the duration of the call
    with default_dtype(batch["X"].dtype), injected_noise(q) as rest:
        loss, metrics = loss_module.loss(
            X=batch["X"],
            D=batch["D"],
            D_obs_counts=batch["D_obs_counts"],
            qc=batch["qc"],
            qc_obs_counts=batch["qc_obs_counts"],
            cfX=batch["cfX"],
            condition_values={
                "idx": batch["idx"],
                "qc": batch["qc"],
                "cfX": batch["cfX"],
                "library_size": batch["library_size"],
            },
            n_particles=n_particles,
        )
        assert not rest, "reference consumed less noise than queued"
    loss.backward()
    grads = {
        k: (torch.zeros_like(p) if p.grad is None else p.grad.detach().clone())
        for k, p in loss_module.named_parameters()
    }
    return loss.detach(), {k: torch.as_tensor(v).detach() for k, v in metrics.items()}, grads
