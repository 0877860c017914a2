"""CPU oracle for the CRADLE-VAE ELBO step and ATE predictor.

TEST INFRASTRUCTURE ONLY.  This file is a plain-PyTorch (CPU, fp32 or fp64)
restatement of the reference's hot path, written function by function from the
reference source, with every random draw replaced by an explicitly injected noise
tensor.  It is imported by ``tests/``, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / the CPU baseline, never by
the product package ``cradle_vae_amd`` (which fails loudly without its HIP library).

Parity pinning: this restatement is checked in ``tests/test_oracle_golden.py``
against fixtures under ``tests/golden/`` that were produced by executing the
reference's own modules (``/root/reference/cradle_vae/models/**``) in the build
container -- see ``oracle/make_golden.py``.  Two third-party classes the reference
uses live in ``pyro-ppl==1.9.1`` (reference ``environment.yml:177``), which is not
installed anywhere in this pipeline: ``pyro.distributions.GammaPoisson`` and
``pyro.distributions.RelaxedBernoulliStraightThrough``.  Their published algorithms
are restated here (``gamma_poisson_log_prob``, ``relaxed_bernoulli_st_rsample``);
for those two pieces: PARITY UNPINNED by the reference (it ships neither the classes nor
vectors for them) -- they are held by closed-form cross-checks (``scipy.stats.nbinom``,
``torch.distributions.NegativeBinomial``) only.

All citations ``file:line`` are relative to ``/root/reference``.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

LEAKY_SLOPE = 0.01  # torch.nn.LeakyReLU default, cradle_vae/models/utils/mlp.py:19
VAR_EPS = 1e-4  # mlp.py:227,243
MU_EPS = 1e-4  # mlp.py:161
RATE_MIN = 1e-6  # mlp.py:163


# --------------------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------------------
def init_params(
    *,
    n_latent: int,
    n_treatments: int,
    n_phenos: int,
    n_qc: int,
    decoder_n_layers: int = 1,
    decoder_n_hidden: int = 400,
    basal_encoder_n_layers: int = 1,
    basal_encoder_n_hidden: int = 400,
    embedding_encoder_n_layers: int = 4,
    embedding_encoder_n_hidden: int = 400,
    mask_prior_prob: float = 0.01,
    embedding_prior_scale: float = 1.0,
    mask_init_logits: float = 0.0,
    gs_temperature: float = 1.0,
    log_x_mean: Optional[Tensor] = None,
    log_x_std: Optional[Tensor] = None,
    seed: int = 0,
    dtype: torch.dtype = torch.float32,
    randomize: bool = True,
) -> Params:
    """State dict with the reference's key names (``loss_module.`` prefix dropped).

    Shapes follow model.py:33-54, correlated_normal_guide.py:48-94, mlp.py:57-74,
    217-243, 286-317 (``sum_mode=False``, ``mean_field_encoder=False``: every shipped
    yaml).  ``randomize`` perturbs ``q_mask_logits`` / ``log_concentration`` away from
    their constant initial values so that tests exercise non-trivial values.
    """
    g = torch.Generator().manual_seed(seed)
    L, T, G, Q = n_latent, n_treatments, n_phenos, n_qc
    p: Params = {}

    def linear(key: str, n_in: int, n_out: int):
        bound = 1.0 / math.sqrt(n_in)  # nn.Linear default init range
        p[f"{key}.weight"] = (torch.rand((n_out, n_in), generator=g, dtype=torch.float64) * 2 - 1) * bound
        p[f"{key}.bias"] = (torch.rand((n_out,), generator=g, dtype=torch.float64) * 2 - 1) * bound

    def gaussian_mlp(prefix: str, n_in: int, n_layers: int, n_hidden: int):
        dims = [n_in] + (n_layers - 1) * [n_hidden] + [n_hidden]  # mlp.py:58 via :230-240
        for i in range(1, len(dims)):
            linear(f"{prefix}.mlp.layers.{i - 1}.0", dims[i - 1], dims[i])
        linear(f"{prefix}.mean_encoder", n_hidden, L)
        linear(f"{prefix}.log_var_encoder", n_hidden, L)

    # model (model.py:33-54); decoder n_layers = decoder_n_layers + 1 (model.py:50)
    p["model.p_E_loc"] = torch.zeros((T, L), dtype=torch.float64)
    p["model.p_E_scale"] = embedding_prior_scale * torch.ones((T, L), dtype=torch.float64)
    p["model.p_mask_probs"] = mask_prior_prob * torch.ones((T, L), dtype=torch.float64)
    p["model.p_qc_E_loc"] = torch.zeros((Q, L), dtype=torch.float64)
    p["model.p_qc_E_scale"] = embedding_prior_scale * torch.ones((Q, L), dtype=torch.float64)
    dec_layers = decoder_n_layers + 1
    dims = [3 * L] + (dec_layers - 1) * [decoder_n_hidden] + [decoder_n_hidden]
    for i in range(1, len(dims)):
        linear(f"model.decoder.mlp.layers.{i - 1}.0", dims[i - 1], dims[i])
    linear("model.decoder.normalized_mean_decoder.0", decoder_n_hidden, G)
    p["model.decoder.log_concentration"] = torch.zeros((G,), dtype=torch.float64)

    # guide (correlated_normal_guide.py:48-94)
    p["guide.param_dict.q_mask_logits"] = mask_init_logits * torch.ones((T, L), dtype=torch.float64)
    gaussian_mlp("guide.embedding_encoder", L + T, embedding_encoder_n_layers, embedding_encoder_n_hidden)
    p["guide.treatment_one_hot"] = torch.eye(T, dtype=torch.float64)
    p["guide.qc_encoder_input"] = torch.ones((Q, L), dtype=torch.float64)
    gaussian_mlp("guide.qc_encoder", L, embedding_encoder_n_layers, embedding_encoder_n_hidden)
    gaussian_mlp("guide.z_basal_encoder", G + 2 * L, basal_encoder_n_layers, basal_encoder_n_hidden)
    p["guide.gs_temperature"] = gs_temperature * torch.ones((1,), dtype=torch.float64)
    p["guide.normalization_module.log_mean"] = (
        torch.zeros((G,), dtype=torch.float64) if log_x_mean is None else log_x_mean.to(torch.float64)
    )
    p["guide.normalization_module.log_scale"] = (
        torch.ones((G,), dtype=torch.float64) if log_x_std is None else log_x_std.to(torch.float64)
    )
    if randomize:
        p["guide.param_dict.q_mask_logits"] += 1.5 * torch.randn((T, L), generator=g, dtype=torch.float64)
        p["model.decoder.log_concentration"] += 0.7 * torch.randn((G,), generator=g, dtype=torch.float64)
    return {k: v.to(dtype) for k, v in p.items()}


TRAINABLE_SUFFIXES = (".weight", ".bias", "q_mask_logits", "log_concentration")


def trainable_keys(params: Params):
    return [k for k in params if k.endswith(TRAINABLE_SUFFIXES)]


# --------------------------------------------------------------------------------------
# distributions restated
# --------------------------------------------------------------------------------------
def clamp_probs(p: Tensor) -> Tensor:
    """torch.distributions.utils.clamp_probs (eps = machine eps of the dtype)."""
    eps = torch.finfo(p.dtype).eps
    return p.clamp(min=eps, max=1 - eps)


def normal_log_prob(value: Tensor, loc: Tensor, scale: Tensor) -> Tensor:
    """torch.distributions.Normal.log_prob."""
    var = scale**2
    return -((value - loc) ** 2) / (2 * var) - scale.log() - math.log(math.sqrt(2 * math.pi))


def bernoulli_logits_log_prob(value: Tensor, logits: Tensor) -> Tensor:
    """torch.distributions.Bernoulli.log_prob = -BCE_with_logits(logits, value)."""
    logits, value = torch.broadcast_tensors(logits, value)
    return -F.binary_cross_entropy_with_logits(logits, value, reduction="none")


def bernoulli_probs_log_prob(value: Tensor, probs: Tensor) -> Tensor:
    """Bernoulli(probs=p).log_prob: logits = probs_to_logits(p, is_binary=True)."""
    ps = clamp_probs(probs)
    logits = ps.log() - (-ps).log1p()
    return bernoulli_logits_log_prob(value, logits)


def relaxed_bernoulli_st_rsample(logits: Tensor, temperature: Tensor, uniforms: Tensor) -> Tuple[Tensor, Tensor]:
    """pyro 1.9.1 RelaxedBernoulliStraightThrough.rsample over torch
    LogitRelaxedBernoulli.rsample + SigmoidTransform, with the uniform draw injected.
    (gumbel_softmax_bernoulli.py:5-17; call site correlated_normal_guide.py:139-141.)

    Returns (hard sample with straight-through gradient, soft sample)."""
    probs = clamp_probs(torch.sigmoid(logits).expand(uniforms.shape))
    u = clamp_probs(uniforms)
    y = (u.log() - (-u).log1p() + probs.log() - (-probs).log1p()) / temperature
    soft = clamp_probs(torch.sigmoid(y))
    hard = soft.round()
    # straight-through: forward value = hard, backward = identity onto soft
    return soft + (hard - soft).detach(), soft


def gamma_poisson_log_prob(value: Tensor, concentration: Tensor, rate: Tensor) -> Tensor:
    """pyro 1.9.1 GammaPoisson.log_prob (call sites loss_modules.py:134-135,
    predictor.py:158-159)."""
    post = concentration + value
    log_beta = torch.lgamma(concentration) + torch.lgamma(value + 1) - torch.lgamma(post + 1)
    return -log_beta - post.log() + concentration * rate.log() - post * (1 + rate).log()


# --------------------------------------------------------------------------------------
# networks
# --------------------------------------------------------------------------------------
def residual_mlp(params: Params, prefix: str, x: Tensor) -> Tensor:
    """ResidualMLP.forward with last_layer_residual=True (mlp.py:111-119)."""
    i = 0
    while f"{prefix}.layers.{i}.0.weight" in params:
        y = F.leaky_relu(
            F.linear(x, params[f"{prefix}.layers.{i}.0.weight"], params[f"{prefix}.layers.{i}.0.bias"]),
            LEAKY_SLOPE,
        )
        x = y if i == 0 else y + x
        i += 1
    return x


def gaussian_likelihood_mlp(params: Params, prefix: str, x: Tensor) -> Tuple[Tensor, Tensor]:
    """BaseGaussianLikelihoodMLP.forward (mlp.py:122-142) -> (mean, stddev)."""
    shape = x.shape
    if x.dim() == 3:
        x = x.reshape(shape[0] * shape[1], shape[2])
    z = residual_mlp(params, f"{prefix}.mlp", x)
    mean = F.linear(z, params[f"{prefix}.mean_encoder.weight"], params[f"{prefix}.mean_encoder.bias"])
    var = torch.exp(
        F.linear(z, params[f"{prefix}.log_var_encoder.weight"], params[f"{prefix}.log_var_encoder.bias"])
    ) + VAR_EPS
    if len(shape) == 3:
        mean = mean.reshape(shape[0], shape[1], -1)
        var = var.reshape(shape[0], shape[1], -1)
    var = torch.abs(var)
    var = torch.nan_to_num(var, nan=1e-6, posinf=1e6, neginf=1e-6)
    mean = torch.nan_to_num(mean, nan=1e-6, posinf=1e6, neginf=1e-6)
    return mean, var.sqrt()


def library_nb_decoder(params: Params, z: Tensor, library_size: Tensor) -> Tuple[Tensor, Tensor]:
    """BaseLibraryGammaPoissonSharedConcentrationLikelihood.forward (mlp.py:145-168)
    -> (concentration (G,), rate (P,N,G))."""
    P, N, _ = z.shape
    x = z.reshape(P * N, -1)
    lib = library_size.expand(P, -1).reshape(-1, 1)
    h = residual_mlp(params, "model.decoder.mlp", x)
    logits = F.linear(
        h,
        params["model.decoder.normalized_mean_decoder.0.weight"],
        params["model.decoder.normalized_mean_decoder.0.bias"],
    )
    mu = (lib * torch.softmax(logits, dim=-1)).reshape(P, N, -1)
    concentration = torch.exp(params["model.decoder.log_concentration"])
    rate = torch.clamp(concentration / (mu + MU_EPS), min=RATE_MIN)
    return concentration, rate


def log_standardize(params: Params, x: Tensor) -> Tensor:
    """LogStandardizationModule.forward (normalization.py:41-45)."""
    normx = (torch.log(x + 1) - params["guide.normalization_module.log_mean"]) / params[
        "guide.normalization_module.log_scale"
    ]
    return torch.nan_to_num(normx, nan=1e-6, posinf=1e6, neginf=1e-6)


def sanitize_tensor(t: Tensor) -> Tensor:
    """correlated_normal_guide.py:212-215."""
    t = torch.where(torch.isnan(t), torch.zeros_like(t), t)
    return torch.where(torch.isinf(t), torch.zeros_like(t), t)


# --------------------------------------------------------------------------------------
# guide / model
# --------------------------------------------------------------------------------------
def guide_forward(
    params: Params,
    noise: Dict[str, Tensor],
    n_particles: int,
    X: Optional[Tensor] = None,
    D: Optional[Tensor] = None,
    qc: Optional[Tensor] = None,
    condition_values: Optional[Dict[str, Tensor]] = None,
) -> Tuple[Dict[str, Tensor], Dict[str, Tensor]]:
    """cradle_vae_CorrelatedNormalGuide.forward (correlated_normal_guide.py:102-209).

    ``noise``: ``u_mask`` (P,T,L) uniforms, ``eps_E`` (P,T,L), ``eps_qc_E`` (P,Q,L),
    ``eps_z`` (P,N,L) standard normals, consumed in that (the reference's RNG) order.
    Distributions are returned as plain tensors: ``q_<var>_mean/_std`` and
    ``q_mask_probs`` / ``q_mask_logits``.
    """
    cv = {} if condition_values is None else condition_values
    P = n_particles
    dists: Dict[str, Tensor] = {}
    samples: Dict[str, Tensor] = {}

    logits = sanitize_tensor(params["guide.param_dict.q_mask_logits"])  # :132
    dists["q_mask_logits"] = logits
    dists["q_mask_probs"] = torch.sigmoid(logits)
    if "mask" not in cv:  # :138-143
        samples["mask"], dists["q_mask_soft"] = relaxed_bernoulli_st_rsample(
            logits, params["guide.gs_temperature"], noise["u_mask"]
        )
    else:
        samples["mask"] = cv["mask"]

    one_hot = params["guide.treatment_one_hot"].unsqueeze(0).expand(P, -1, -1)  # :147-149
    emb_in = torch.cat([samples["mask"], one_hot], dim=-1)  # :150-152
    dists["q_E_mean"], dists["q_E_std"] = gaussian_likelihood_mlp(params, "guide.embedding_encoder", emb_in)
    samples["E"] = cv["E"] if "E" in cv else dists["q_E_mean"] + dists["q_E_std"] * noise["eps_E"]  # :155-159

    qc_in = params["guide.qc_encoder_input"].unsqueeze(0).expand(P, -1, -1)  # :162
    dists["q_qc_E_mean"], dists["q_qc_E_std"] = gaussian_likelihood_mlp(params, "guide.qc_encoder", qc_in)
    samples["qc_E"] = (
        cv["qc_E"] if "qc_E" in cv else dists["q_qc_E_mean"] + dists["q_qc_E_std"] * noise["eps_qc_E"]
    )  # :164-168

    if X is not None and D is not None and qc is not None:  # :170
        enc_in = log_standardize(params, X).unsqueeze(0).expand(P, -1, -1)  # :173-182
        latent_offset = torch.matmul(D, samples["mask"] * samples["E"])  # :186-188
        latent_qc_offset = torch.matmul(qc, samples["qc_E"])  # :189-191
        latent_qc_cf_offset = torch.matmul(1 - qc, samples["qc_E"])  # :192-194
        cf_in = torch.cat([enc_in, latent_offset, latent_qc_cf_offset], dim=-1)  # :196
        dists["q_z_cf_basal_mean"], dists["q_z_cf_basal_std"] = gaussian_likelihood_mlp(
            params, "guide.z_basal_encoder", cf_in
        )  # :197
        main_in = torch.cat([enc_in, latent_offset, latent_qc_offset], dim=-1)  # :198
        dists["q_z_basal_mean"], dists["q_z_basal_std"] = gaussian_likelihood_mlp(
            params, "guide.z_basal_encoder", main_in
        )  # :201
        samples["z_basal"] = dists["q_z_basal_mean"] + dists["q_z_basal_std"] * noise["eps_z"]  # :202-204
    if "z_basal" in cv:  # :206-207
        samples["z_basal"] = cv["z_basal"]
    return dists, samples


def model_decode(
    params: Params,
    D: Tensor,
    qc: Tensor,
    samples: Dict[str, Tensor],
    library_size: Tensor,
) -> Tuple[Tensor, Tensor, Tensor]:
    """cradle_vae_Model.forward, sum_mode=False branch (model.py:116-129) without the
    x_good/x_bad draws -> (concentration, rate_good, rate_bad)."""
    ME = samples["E"] * samples["mask"]
    off = torch.matmul(D, ME)
    z_good = torch.cat([samples["z_basal"], off, torch.matmul(torch.zeros_like(qc), samples["qc_E"])], dim=-1)
    z_bad = torch.cat([samples["z_basal"], off, torch.matmul(torch.ones_like(qc), samples["qc_E"])], dim=-1)
    c, rate_good = library_nb_decoder(params, z_good, library_size)
    _, rate_bad = library_nb_decoder(params, z_bad, library_size)
    return c, rate_good, rate_bad


# --------------------------------------------------------------------------------------
# loss
# --------------------------------------------------------------------------------------
def reweight_plated(cond: Tensor, counts: Tensor, term: Tensor) -> Tensor:
    """_compute_reweighted_perturbation_plated_loss_term (loss_modules.py:200-232)."""
    nz = (cond != 0).to(term.dtype)
    scaling = 1 / counts.to(term.dtype)
    scaling = torch.where(torch.isinf(scaling), torch.zeros_like(scaling), scaling).reshape(1, -1)
    return torch.matmul(nz * scaling, term)


def kldiv_normal(mu1: Tensor, sigma1: Tensor, mu2: Tensor, sigma2: Tensor) -> Tensor:
    """loss_modules.py:190-196 (not the textbook KL: reproduced as written)."""
    lv1 = 2 * sigma1.log()
    lv2 = 2 * sigma2.log()
    return torch.mean(-0.5 * torch.sum(1.0 + lv1 - lv2 - (mu1 - mu2) ** 2 - (lv1 - lv2).exp(), dim=1), dim=0)


def elbo_loss(
    params: Params,
    *,
    X: Tensor,
    D: Tensor,
    qc: Tensor,
    cfX: Tensor,
    library_size: Tensor,
    D_obs_counts: Tensor,
    qc_obs_counts: Tensor,
    noise: Dict[str, Tensor],
    cf_noise: Dict[str, Tensor],
    n_particles: int,
    beta: float = 1.0,
    alpha: Optional[float] = None,
    as_written: bool = False,
) -> Tuple[Tensor, Dict[str, Tensor], Dict[str, Tensor]]:
    """PerturbationPlatedELBOLossModule.loss (loss_modules.py:110-187) with
    .forward (:54-106) inlined.  Returns (loss, metrics, intermediates).

    ``as_written`` (timing only, bench.py's cpu_baseline): also does the work the reference does and
    throws away -- the x_good / x_bad Gamma -> Poisson draws of model.py:131-132 (pyro
    GammaPoisson.sample = Poisson(Gamma(c, rate).sample()).sample()) and the argument / support checks
    torch.distributions runs with validate_args on (its default in the reference).  The loss is the same."""
    P = n_particles
    g_d, g_s = guide_forward(params, noise, P, X=X, D=D, qc=qc)  # :69-74
    c, rate_good, rate_bad = model_decode(params, D, qc, g_s, library_size)  # :88-93
    if as_written:
        with torch.no_grad():
            for rate in (rate_good, rate_bad):  # model.py:131-132
                assert bool((c > 0).all()) and bool((rate > 0).all())  # GammaPoisson arg_constraints
                lam = torch._standard_gamma(c.expand_as(rate).contiguous()) / rate
                assert bool((lam >= 0).all())  # Poisson(rate) arg constraint
                torch.poisson(lam)
            assert bool(((X >= 0) & (X == X.round())).all())  # GammaPoisson.support in log_prob, twice (:134-135)
            assert bool(((X >= 0) & (X == X.round())).all())
            for k in ("q_E_std", "q_qc_E_std", "q_z_basal_std", "q_z_cf_basal_std"):  # Normal(scale > 0)
                assert bool((g_d[k] > 0).all())
    cf_d, cf_s = guide_forward(params, cf_noise, P, X=cfX, D=D, qc=1 - qc)  # :97-102

    logp_good = gamma_poisson_log_prob(X, c, rate_good)  # :134
    logp_bad = gamma_poisson_log_prob(X, c, rate_bad)  # :135
    a = alpha if alpha is not None else qc.sum() / qc.shape[0]  # :137-139
    qc1 = qc.max(dim=1)[0].unsqueeze(-1) if qc.shape[1] > 1 else qc  # :140-141
    terms: Dict[str, Tensor] = {}
    terms["reconstruction"] = (a * ((1 - qc1) * logp_good) + (1 - a) * (qc1 * logp_bad)).sum(-1)  # :142-143

    idx = torch.nonzero(cfX.sum(1) != 0).squeeze(1)  # :145
    kl = beta * kldiv_normal(
        cf_d["q_z_basal_mean"][:, idx, :],
        cf_d["q_z_basal_std"][:, idx, :],
        g_d["q_z_cf_basal_mean"][:, idx, :],
        g_d["q_z_cf_basal_std"][:, idx, :],
    ).sum()  # :146-151

    # :153-177, in the guide's insertion order mask, E, qc_E, z_basal
    lt = bernoulli_logits_log_prob(g_s["mask"], torch.logit(params["model.p_mask_probs"])) - bernoulli_probs_log_prob(
        g_s["mask"], g_d["q_mask_probs"]
    )
    terms["mask"] = reweight_plated(D, D_obs_counts, lt).sum(-1)
    lt = normal_log_prob(g_s["E"], params["model.p_E_loc"], params["model.p_E_scale"]) - normal_log_prob(
        g_s["E"], g_d["q_E_mean"], g_d["q_E_std"]
    )
    terms["E"] = reweight_plated(D, D_obs_counts, lt).sum(-1)
    lt = normal_log_prob(g_s["qc_E"], params["model.p_qc_E_loc"], params["model.p_qc_E_scale"]) - normal_log_prob(
        g_s["qc_E"], g_d["q_qc_E_mean"], g_d["q_qc_E_std"]
    )
    terms["qc_E"] = reweight_plated(qc1, qc_obs_counts, lt).sum(-1)
    z = g_s["z_basal"]
    lt = normal_log_prob(z, torch.zeros_like(z), torch.ones_like(z)) - normal_log_prob(
        z, g_d["q_z_basal_mean"], g_d["q_z_basal_std"]
    )
    terms["z_basal"] = lt.sum(-1)

    batch_elbo = sum(terms.values())  # :179
    loss = -batch_elbo.mean() + kl  # :180-181
    metrics = {f"loss_term_{k}": -v.detach().mean() for k, v in terms.items()}  # :182-184
    metrics["cf_KLdivergence"] = kl.detach()  # :185
    inter = {
        "mask": g_s["mask"],
        "cf_mask": cf_s["mask"],
        "E": g_s["E"],
        "qc_E": g_s["qc_E"],
        "z_basal": g_s["z_basal"],
        "q_z_basal_mean": g_d["q_z_basal_mean"],
        "q_z_basal_std": g_d["q_z_basal_std"],
        "q_z_cf_basal_mean": g_d["q_z_cf_basal_mean"],
        "q_z_cf_basal_std": g_d["q_z_cf_basal_std"],
        "cf_q_z_basal_mean": cf_d["q_z_basal_mean"],
        "cf_q_z_basal_std": cf_d["q_z_basal_std"],
        "rate_good": rate_good,
        "rate_bad": rate_bad,
        "batch_elbo": batch_elbo,
    }
    inter.update({f"term_{k}": v for k, v in terms.items()})
    return loss, metrics, inter


def elbo_loss_and_grads(params: Params, **kw):
    """loss + gradients of every trainable tensor (autograd on the restatement)."""
    keys = trainable_keys(params)
    leaves = {k: params[k].detach().clone().requires_grad_(True) for k in keys}
    p2 = dict(params)
    p2.update(leaves)
    loss, metrics, inter = elbo_loss(p2, **kw)
    grads = torch.autograd.grad(loss, [leaves[k] for k in keys], allow_unused=True)
    gdict = {k: (torch.zeros_like(leaves[k]) if g is None else g) for k, g in zip(keys, grads)}
    return loss.detach(), metrics, {k: v.detach() for k, v in inter.items()}, gdict


# --------------------------------------------------------------------------------------
# predictor
# --------------------------------------------------------------------------------------
def predictive_iwelbo(
    params: Params,
    *,
    X: Tensor,
    D: Tensor,
    qc: Tensor,
    library_size: Tensor,
    plated: Dict[str, Tensor],
    eps_z: Tensor,
    n_particles: int,
) -> Tensor:
    """One batch of PerturbationPlatedPredictor.compute_predictive_iwelbo
    (predictor.py:125-185): plated variables (mask, E, qc_E) fixed, z_basal drawn from
    the guide with injected noise; returns (N,) IWELBO."""
    g_d, g_s = guide_forward(
        params, {"eps_z": eps_z}, n_particles, X=X, D=D, qc=qc, condition_values=dict(plated)
    )
    c, rate_good, rate_bad = model_decode(params, D, qc, g_s, library_size)
    lp_good = gamma_poisson_log_prob(X, c, rate_good)
    lp_bad = gamma_poisson_log_prob(X, c, rate_bad)
    b_qc = qc.max(dim=1)[0].unsqueeze(-1) if qc.shape[1] > 1 else qc  # :160-164 (no alpha)
    terms = ((1 - b_qc) * lp_good + b_qc * lp_bad).sum(-1)
    z = g_s["z_basal"]
    terms = terms + normal_log_prob(z, torch.zeros_like(z), torch.ones_like(z)).sum(-1)
    terms = terms - normal_log_prob(z, g_d["q_z_basal_mean"], g_d["q_z_basal_std"]).sum(-1)
    return torch.logsumexp(terms, dim=0) - math.log(n_particles)  # :183-185


def nb_mean_good(params: Params, D: Tensor, samples: Dict[str, Tensor], library_size: Tensor) -> Tensor:
    """Mean of p_x_good = concentration / rate for a dosage matrix D (rows = conditions)."""
    qc = torch.zeros((D.shape[0], samples["qc_E"].shape[1]), dtype=D.dtype)
    c, rate_good, _ = model_decode(params, D, qc, samples, library_size)
    return c / rate_good


def ate_from_draws(X_control: Tensor, X_alt: Tensor, method: str) -> Tensor:
    """Accumulation of estimate_model_average_treatment_effect
    (analysis/average_treatment_effects.py:171-201) given the drawn counts.
    X_control (P,G), X_alt (T_alt,P,G) -> (T_alt,G)."""
    if method == "perturbseq":
        X_control = torch.log2(1e4 * X_control / X_control.sum(dim=-1, keepdim=True) + 1)
        X_alt = torch.log2(1e4 * X_alt / X_alt.sum(dim=-1, keepdim=True) + 1)
    n = X_control.shape[0]
    return X_alt.sum(1) / n - X_control.sum(0, keepdim=True) / n


@torch.no_grad()
def model_average_treatment_effect(
    params: Params,
    *,
    dosages_alt: Tensor,
    dosages_control: Tensor,
    quality: Tensor,
    library_size: Tensor,
    n_particles: int,
    method: str,
    batch_size: int = 500,
    seed: int = 0,
    max_alt: Optional[int] = None,
    max_chunks: Optional[int] = None,
) -> Tuple[Tensor, int]:
    """estimate_model_average_treatment_effect as the reference writes it
    (analysis/average_treatment_effects.py:78-203), every draw from torch's global generator in the
    reference's order: per particle chunk one guide call for mask / E / qc_E (:143-148), one model call under
    the control dosage (:157-162) and ONE MODEL CALL PER ALTERNATIVE PERTURBATION (:178-195), each with a fresh
    prior draw of z_basal (model.py:112), both decoder variants and the Gamma -> Poisson draw of x_good AND
    x_bad (model.py:128-132; only x_good is read, :190).

    ``max_alt`` / ``max_chunks`` bound the work for timing (bench.py's cpu_baseline: a sample of the same
    workload); the returned effects then cover that sample only.  -> ((n_alt_used, G) effects, model calls made)."""
    assert method in ("mean", "perturbseq")  # :94-95
    torch.manual_seed(seed)  # :92
    dt = params["model.decoder.log_concentration"].dtype
    T, L = params["model.p_E_loc"].shape
    Q = params["model.p_qc_E_loc"].shape[0]
    G = params["model.decoder.log_concentration"].shape[0]
    n_alt = dosages_alt.shape[0] if max_alt is None else min(max_alt, dosages_alt.shape[0])
    X_control_sum = torch.zeros((1, G), dtype=dt)
    X_alt_sums = torch.zeros((n_alt, G), dtype=dt)
    calls = 0

    def model_call(D: Tensor, samples: Dict[str, Tensor], P: int) -> Tensor:
        s = dict(samples)
        s["z_basal"] = torch.randn((P, D.shape[0], L), dtype=dt)  # p_z_basal.sample (model.py:89,112)
        if "E" not in s:  # the n_phenos probe call (:117-120) conditions on nothing: prior draws, model.py:101-112
            s["E"] = params["model.p_E_loc"] + params["model.p_E_scale"] * torch.randn((P, T, L), dtype=dt)
            s["mask"] = torch.bernoulli(params["model.p_mask_probs"].expand(P, T, L))
            s["qc_E"] = params["model.p_qc_E_loc"] + params["model.p_qc_E_scale"] * torch.randn((P, Q, L), dtype=dt)
        c, rate_good, rate_bad = model_decode(params, D, quality, s, library_size)
        xs = []
        for rate in (rate_good, rate_bad):  # model.py:131-132: pyro GammaPoisson.sample = Poisson(Gamma(c, rate))
            lam = (torch._standard_gamma(c.expand_as(rate).contiguous()) / rate).clamp_(min=torch.finfo(dt).tiny)
            xs.append(torch.poisson(lam))
        return xs[0].squeeze(1)  # x_good (:171,190)

    def transform(X: Tensor) -> Tensor:
        if method == "perturbseq":  # :172-175
            X = 1e4 * X / torch.sum(X, dim=1, keepdim=True)
            X = torch.log2(X + 1)
        return X

    model_call(dosages_control, {}, 1)  # :117-120: one throw-away call to learn n_phenos; it consumes the generator
    done = 0
    for ci, i in enumerate(range(0, n_particles, batch_size)):
        if max_chunks is not None and ci >= max_chunks:
            break
        P = min(batch_size, n_particles - i)
        noise = dict(u_mask=torch.rand((P, T, L), dtype=dt), eps_E=torch.randn((P, T, L), dtype=dt),
                     eps_qc_E=torch.randn((P, Q, L), dtype=dt))
        _, g_s = guide_forward(params, noise, P)  # :143-148
        X_control_sum[0] += transform(model_call(dosages_control, g_s, P)).sum(0)  # :157-176
        calls += 1
        for t_idx in range(n_alt):  # :178-195
            X_alt_sums[t_idx] += transform(model_call(dosages_alt[t_idx:t_idx + 1], g_s, P)).sum(0)
            calls += 1
        done += P
    return X_alt_sums / max(done, 1) - X_control_sum / max(done, 1), calls  # :199-201
