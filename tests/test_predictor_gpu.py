"""Predictor (IWELBO, ATE) and the Gamma-Poisson sampler kernel on the GPU.

IWELBO and ATE are checked against the reference's own outputs (tests/golden/predictor_tiny.npz,
written by oracle/make_golden.py with injected latent noise; in the ATE fixture the count
sampler is replaced by its mean because the sampler is pyro's).  The sampler itself cannot be
bit-compared with torch's generator; it is checked through its moments, its pmf against
scipy's negative binomial, and exact identities between its outputs.  Parity of the sampled
ATE with the reference's sampled ATE is therefore statistical only ("parity unpinned").
"""
import numpy as np
import pytest
import torch

from tests import golden_io as gio
from tests.test_modular_gpu import build_loss_module

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _predictor(case):
    from cradle_vae_amd.predictor import cradle_vae_Predictor

    lm = build_loss_module(case)
    return cradle_vae_Predictor(lm.model, lm.guide)


class _Loader:
    """Minimal DataLoader stand-in (ordered batches + .dataset.convert_idx_to_ids)."""

    def __init__(self, batch, bs):
        self.batch, self.bs = batch, bs
        self.dataset = self

    def convert_idx_to_ids(self, idx):
        return np.asarray(idx)

    def __iter__(self):
        n = self.batch["X"].shape[0]
        for s in range(0, n, self.bs):
            b = {k: v[s:s + self.bs] for k, v in self.batch.items() if k in ("X", "D", "qc", "library_size", "cfX")}
            b["idx"] = torch.arange(s, min(n, s + self.bs))
            yield b


def test_iwelbo_matches_reference():
    case = gio.load_predictor("tiny", torch.float32, DEV)
    pred = _predictor(case)
    iw = case["iwelbo"]
    P, bs = int(iw["n_particles"]), int(iw["batch"])
    noise = {k[len("noise/"):]: v for k, v in iw.items() if k.startswith("noise/")}
    df = pred.compute_predictive_iwelbo(_Loader(case["batch"], bs), n_particles=P, noise=noise)
    ref = iw["values"].cpu().numpy()
    assert list(df.index) == list(iw["ids"].cpu().numpy())
    np.testing.assert_allclose(df["IWELBO"].to_numpy(), ref, rtol=1e-4, atol=1e-3)


def test_iwelbo_live_variant_equals_both_variants_and_falls_back():
    """With qc in {0,1} the predictor scores one decoder variant per cell; it must equal the
    reference's two-variant form, and fractional qc must fall back to that form by itself."""
    case = gio.load_predictor("tiny", torch.float32, DEV)
    pred = _predictor(case)
    iw = case["iwelbo"]
    P, bs = int(iw["n_particles"]), int(iw["batch"])
    noise = {k[len("noise/"):]: v for k, v in iw.items() if k.startswith("noise/")}
    assert pred.live_variant_only
    live = pred.compute_predictive_iwelbo(_Loader(case["batch"], bs), n_particles=P, noise=noise)["IWELBO"].to_numpy()
    pred.live_variant_only = False
    both = pred.compute_predictive_iwelbo(_Loader(case["batch"], bs), n_particles=P, noise=noise)["IWELBO"].to_numpy()
    np.testing.assert_allclose(live, both, rtol=1e-6, atol=1e-5)
    frac = dict(case["batch"])
    frac["qc"] = frac["qc"].clone()
    frac["qc"][0, 0] = 0.25
    both_frac = pred.compute_predictive_iwelbo(_Loader(frac, bs), n_particles=P, noise=noise)["IWELBO"].to_numpy()
    pred.live_variant_only = True
    auto_frac = pred.compute_predictive_iwelbo(_Loader(frac, bs), n_particles=P, noise=noise)["IWELBO"].to_numpy()
    assert pred.live_variant_only  # restored after the fallback
    np.testing.assert_allclose(auto_frac, both_frac, rtol=1e-6, atol=1e-5)
    assert abs(auto_frac[0] - live[0]) > 0  # the fractional cell really scores differently


@pytest.mark.parametrize("method", ["mean", "perturbseq"])
def test_ate_matches_reference(method):
    case = gio.load_predictor("tiny", torch.float32, DEV)
    pred = _predictor(case)
    ate = case["ate"]
    n_particles, chunk = int(ate["n_particles"]), int(ate["chunk"])
    chunks = []
    for ci in range((n_particles + chunk - 1) // chunk):
        chunks.append({k[len(f"noise{ci}/"):]: v for k, v in ate.items() if k.startswith(f"noise{ci}/")})
    Q = case["cfg"]["n_qc"]
    res = pred.estimate_average_treatment_effects(
        dosages_alt=ate["D_alt"], dosages_control=ate["D_control"], quality=torch.ones(1, Q), method=method,
        n_particles=n_particles, condition_values=dict(library_size=1e4 * torch.ones(1)), batch_size=chunk,
        perturbation_names_alt=[f"p{i}" for i in range(ate["D_alt"].shape[0])], perturbation_name_control="ctl",
        sampler="mean", noise_chunks=chunks)
    ref = ate[method].cpu().numpy()
    assert res.X.shape == ref.shape and res.uns["control"] == "ctl" and list(res.obs.index)[0] == "p0"
    np.testing.assert_allclose(res.X, ref, rtol=1e-4, atol=1e-4 * float(np.abs(ref).max()))


def test_sample_observations_matches_reference():
    """predictor.sample_observations against the reference's own output (count sampler in mean mode, latent
    draws injected): values, row order, and the obs columns exactly as the reference writes them -- including
    its particle_idx column, which holds a dosage counter (predictor.py:251); the particle index is opt-in."""
    case = gio.load_predictor("tiny", torch.float32, DEV)
    pred = _predictor(case)
    so = case["samples"]
    P = int(so["n_particles"])
    noise = {k[len("noise/"):]: v for k, v in so.items() if k.startswith("noise/")}
    Q = case["cfg"]["n_qc"]
    names = [f"pert{i}" for i in range(so["dosages"].shape[0])]
    kw = dict(dosages=so["dosages"], perturbation_names=names, n_particles=P,
              condition_values=dict(library_size=1e4 * torch.ones(1)), qc=torch.ones(1, Q), sampler="mean", noise=noise)
    good, bad = pred.sample_observations(**kw)
    for got, ref in ((good, so["good"]), (bad, so["bad"])):
        ref = ref.cpu().numpy()
        np.testing.assert_allclose(got.X, ref, rtol=1e-4, atol=1e-4 * float(np.abs(ref).max()))
        assert np.array_equal(got.obs["perturbation_idx"].to_numpy(), so["perturbation_idx"].cpu().numpy())
        assert np.array_equal(got.obs["particle_idx"].to_numpy(), so["particle_idx"].cpu().numpy())
        assert list(got.obs["perturbation_name"][:P + 1]) == ["pert0"] * P + ["pert1"]
    assert not np.allclose(good.X, bad.X)
    fixed, _ = pred.sample_observations(particle_idx="particle", **kw)
    assert list(fixed.obs["particle_idx"][:2 * P]) == list(range(P)) * 2


def test_iwelbo_falls_back_to_dense_products_when_a_cell_has_more_dosages_than_planned():
    case = gio.load_predictor("tiny", torch.float32, DEV)
    pred = _predictor(case)
    iw = case["iwelbo"]
    P, bs = int(iw["n_particles"]), int(iw["batch"])
    noise = {k[len("noise/"):]: v for k, v in iw.items() if k.startswith("noise/")}
    assert int((case["batch"]["D"] != 0).sum(1).max()) >= 2
    pred.guide.max_dosages_per_cell = pred.model.max_dosages_per_cell = 1  # two-guide cells overflow the plan
    df = pred.compute_predictive_iwelbo(_Loader(case["batch"], bs), n_particles=P, noise=noise)
    assert not pred.guide.dense_dosage_matmul  # restored after the dense re-run
    np.testing.assert_allclose(df["IWELBO"].to_numpy(), iw["values"].cpu().numpy(), rtol=1e-4, atol=1e-3)


def _sampler_inputs(P, C, G, seed=0):
    g = torch.Generator().manual_seed(seed)
    logits = (2.0 * torch.randn(P * C, G, generator=g)).to(DEV)
    # same logits for every particle of a condition so that particles are i.i.d. draws
    logits = logits[:C].repeat(P, 1).contiguous()
    log_c = (0.8 * torch.randn(G, generator=g)).to(DEV)
    lib = torch.tensor([200.0 * G / 16], device=DEV)
    return logits, log_c, lib


@pytest.mark.parametrize("logc_shift", [0.0, -3.0])
def test_sampler_moments_and_pmf(logc_shift):
    """logc_shift = -3: concentrations around 0.05, so that most genes of a row take the Gamma -> Poisson composition
    (rate < 1/4) -- more than half of them, which is where the kernel draws them in one pass instead of re-queueing
    the small Poisson rates (sampler.hip: 2 * n_queued > G)."""
    from scipy.stats import nbinom

    from cradle_vae_amd import ops

    P, C, G = 8192, 3, 64
    logits, log_c, lib = _sampler_inputs(P, C, G)
    log_c = log_c + logc_shift
    consts = ops.nb_gene_consts(log_c)
    sums, x = ops.nb_sample_accumulate(logits, lib, consts, P, C, seed=1234, return_samples=True)
    x = x.reshape(P, C, G).double().cpu()
    assert float(x.min()) >= 0 and torch.equal(x, x.round())
    c = log_c.double().exp().cpu()
    mu = lib.double().cpu() * torch.softmax(logits[:C].double().cpu(), -1)
    rate = torch.clamp(c / (mu + 1e-4), min=1e-6)
    mean, var = c / rate, c / rate**2 * (1 + rate)
    se = (var / P).sqrt()
    z = (x.mean(0) - mean) / se
    assert float(z.abs().max()) < 5.0, float(z.abs().max())  # C*G = 192 z-scores
    # variance: relative error ~ sqrt((kurtosis-1)/P); loose bound
    rel = (x.var(0) - var).abs() / var
    assert float(rel.median()) < 0.08 and float(rel.max()) < 0.6, (float(rel.median()), float(rel.max()))
    # the sums output is the sum of the returned draws over particles (exact: integers in fp32)
    assert torch.allclose(sums.double().cpu(), x.sum(0), rtol=1e-6)
    # pmf of a few (condition, gene) cells against scipy's negative binomial (chi-square)
    # cells chosen by rate so that both exact samplers are checked on either side of their switch
    # (NegBinomial inversion for rate >= 1/4, Gamma -> Poisson below): extremes and the two nearest 1/4
    flat = rate.flatten()
    above, below = torch.nonzero(flat >= 0.25).flatten(), torch.nonzero(flat < 0.25).flatten()
    if logc_shift == 0.0:
        assert len(above) > 10 and len(below) > 10
    else:
        assert float((rate < 0.25).float().mean(dim=1).min()) > 0.5  # every row: the one-pass form
    picks = []
    for grp in (above, below):
        if len(grp) >= 3:
            picks += [grp[flat[grp].argmax()], grp[flat[grp].argmin()], grp[len(grp) // 2]]
    assert len(picks) >= 3
    for (t, g) in [(int(i) // G, int(i) % G) for i in picks]:
        xs = x[:, t, g].numpy().astype(np.int64)
        kmax = int(np.quantile(xs, 0.995)) + 1
        obs = np.bincount(np.minimum(xs, kmax), minlength=kmax + 1).astype(np.float64)
        dist = nbinom(n=float(c[g]), p=float(rate[t, g] / (1 + rate[t, g])))
        exp = dist.pmf(np.arange(kmax + 1))
        exp[kmax] = 1 - exp[:kmax].sum()
        exp *= P
        keep = exp > 5
        chi2 = float((((obs - exp) ** 2) / np.maximum(exp, 1e-9))[keep].sum())
        dof = int(keep.sum())
        assert chi2 < dof + 6 * np.sqrt(2 * dof) + 10, (t, g, chi2, dof)


def test_sampler_large_rates_and_determinism():
    from cradle_vae_amd import ops

    P, C, G = 2048, 2, 16
    logits, log_c, _ = _sampler_inputs(P, C, G, seed=3)
    lib = torch.tensor([5.0e4, 20.0], device=DEV)  # condition 0: means in the hundreds/thousands (PTRS branch)
    log_c = log_c + 2.0
    consts = ops.nb_gene_consts(log_c)
    _, x1 = ops.nb_sample_accumulate(logits, lib, consts, P, C, seed=7, return_samples=True)
    _, x2 = ops.nb_sample_accumulate(logits, lib, consts, P, C, seed=7, return_samples=True)
    _, x3 = ops.nb_sample_accumulate(logits, lib, consts, P, C, seed=8, return_samples=True)
    assert torch.equal(x1, x2) and not torch.equal(x1, x3)
    x = x1.reshape(P, C, G).double().cpu()
    c = log_c.double().exp().cpu()
    mu = lib.double().cpu().reshape(C, 1) * torch.softmax(logits[:C].double().cpu(), -1)
    rate = torch.clamp(c / (mu + 1e-4), min=1e-6)
    mean, var = c / rate, c / rate**2 * (1 + rate)
    z = (x.mean(0) - mean) / (var / P).sqrt()
    assert float(z.abs().max()) < 5.0, float(z.abs().max())
    rel = (x.var(0) - var).abs() / var
    assert float(rel.median()) < 0.1, float(rel.median())


def test_sampler_high_concentration_near_poisson_genes():
    """Concentrations of 12-150 with means of 20-500 (trained near-Poisson genes) on the rate >= 1/4 side:
    pmf(0) = (rate/(1+rate))^c is far below fp32 resolution there, so a search upwards from 0 must not be
    cut off before the mode (it returned 0 for every draw once).  Moments and pmf against scipy."""
    from scipy.stats import nbinom

    from cradle_vae_amd import ops

    P, C, G = 8192, 2, 64
    g = torch.Generator().manual_seed(21)
    log_c = 2.5 + 2.5 * torch.rand(G, generator=g)
    c = log_c.double().exp()
    # target means in [20, 500], capped at 3.9 c so that rate = c / mean stays >= 1/4 (second condition: half of that)
    mean_t = torch.minimum(20.0 + 480.0 * torch.rand(C, G, generator=g).double(), 3.9 * c)
    mean_t[1] = torch.clamp(0.5 * mean_t[1], min=15.0)
    lib = mean_t.sum(-1)
    logits = mean_t.log().float().repeat(P, 1).contiguous().to(DEV)  # softmax * lib = the target means
    consts = ops.nb_gene_consts(log_c.to(DEV))
    _, x = ops.nb_sample_accumulate(logits, lib.float().to(DEV), consts, P, C, seed=4321, return_samples=True)
    x = x.reshape(P, C, G).double().cpu()
    c32 = log_c.exp().double()
    mu = lib.float().double().reshape(C, 1) * torch.softmax(logits[:C].double().cpu(), -1)
    rate = torch.clamp(c32 / (mu + 1e-4), min=1e-6)
    assert float(rate.min()) >= 0.25
    mean, var = c32 / rate, c32 / rate**2 * (1 + rate)
    assert float(x.mean(0).min()) > 10.0  # the failure mode: all-zero draws
    z = (x.mean(0) - mean) / (var / P).sqrt()
    assert float(z.abs().max()) < 5.0, float(z.abs().max())
    rel = (x.var(0) - var).abs() / var
    assert float(rel.median()) < 0.05 and float(rel.max()) < 0.2, (float(rel.median()), float(rel.max()))
    for (t, gi) in [(0, 0), (0, int(c32.argmax())), (1, int(c32.argmin())), (1, int(mean[1].argmax()))]:
        xs = x[:, t, gi].numpy().astype(np.int64)
        lo, hi = int(np.quantile(xs, 0.005)), int(np.quantile(xs, 0.995)) + 1
        obs = np.bincount(np.clip(xs, lo, hi) - lo, minlength=hi - lo + 1).astype(np.float64)
        dist = nbinom(n=float(c32[gi]), p=float(rate[t, gi] / (1 + rate[t, gi])))
        exp = dist.pmf(np.arange(lo, hi + 1))
        exp[0] = dist.cdf(lo)
        exp[-1] = 1 - dist.cdf(hi - 1)
        exp *= P
        keep = exp > 5
        chi2 = float((((obs - exp) ** 2) / np.maximum(exp, 1e-9))[keep].sum())
        dof = int(keep.sum())
        assert chi2 < dof + 6 * np.sqrt(2 * dof) + 10, (t, gi, chi2, dof)


@pytest.mark.parametrize("G", [256, 301])  # 301: rows run as 304 genes, the tail masked (no draw, no sum)
def test_sampler_perturbseq_accumulation_matches_torch(G):
    from cradle_vae_amd import ops

    P, C = 64, 5
    logits, log_c, lib = _sampler_inputs(P, C, G, seed=5)
    consts = ops.nb_gene_consts(log_c)
    sums = torch.full((C, G), 0.5, device=DEV)  # accumulates into existing content
    _, x = ops.nb_sample_accumulate(logits, lib, consts, P, C, sums=sums, method="perturbseq", seed=99, return_samples=True)
    x = x.reshape(P, C, G).double()
    ref = 0.5 + torch.log2(1e4 * x / x.sum(-1, keepdim=True) + 1).sum(0)
    assert torch.allclose(sums.double(), ref, rtol=2e-5, atol=1e-4)
    # sampler="mean": deterministic, equals c/rate
    s2, xm = ops.nb_sample_accumulate(logits, lib, consts, P, C, sampler="mean", return_samples=True)
    c = log_c.exp()
    mu = lib * torch.softmax(logits, -1)
    assert torch.allclose(xm, c / torch.clamp(c / (mu + 1e-4), min=1e-6), rtol=2e-5, atol=1e-6)
    assert torch.allclose(s2, xm.reshape(P, C, G).sum(0), rtol=1e-5)


def test_sample_observations_and_sampled_ate_statistics():
    """The sampled ATE converges to the mean-sampler ATE (method 'mean' is linear in the draws)."""
    case = gio.load_predictor("tiny", torch.float32, DEV)
    pred = _predictor(case)
    ate = case["ate"]
    Q = case["cfg"]["n_qc"]
    kw = dict(dosages_alt=ate["D_alt"], dosages_control=ate["D_control"], quality=torch.ones(1, Q), method="mean",
              n_particles=4096, condition_values=dict(library_size=1e4 * torch.ones(1)), batch_size=1024, seed=11)
    a = pred.estimate_average_treatment_effects(sampler="nb", **kw).X
    b = pred.estimate_average_treatment_effects(sampler="mean", **kw).X  # same latent draws (same seed)
    a2 = pred.estimate_average_treatment_effects(sampler="nb", **kw).X
    assert np.array_equal(a, a2)  # deterministic given the seed
    scale = np.abs(b).max()
    assert np.abs(a - b).max() < 0.25 * scale + 5.0, (np.abs(a - b).max(), scale)
    # per-entry s.e. of the difference is ~7 here (means ~300, NB variance ~1e5, 4096 particles; the
    # control draw is shared by the rows), so the mean over the 128 entries has s.e. ~1
    assert abs((a - b).mean()) < 4.5, float((a - b).mean())
    good, bad = pred.sample_observations(
        dosages=torch.cat([ate["D_control"], ate["D_alt"]]), perturbation_names=None, n_particles=3,
        condition_values=dict(library_size=1e4 * torch.ones(1)), qc=torch.ones(1, Q))
    n = 1 + ate["D_alt"].shape[0]
    assert good.X.shape == (3 * n, case["cfg"]["n_phenos"]) == bad.X.shape
    assert (good.X >= 0).all() and np.array_equal(good.X, np.round(good.X))
    assert list(good.obs["perturbation_idx"][:4]) == [0, 0, 0, 1]
    assert list(good.obs["particle_idx"][:4]) == [0, 1, 2, 3]  # the reference's column (predictor.py:251)


def test_sampler_draws_do_not_depend_on_the_particle_split():
    """Philox counters are (gene, row, draw): the same seed gives the same draws whether one workgroup
    walks all particles of a condition (no workspace) or they are split over many (workspace)."""
    from cradle_vae_amd import hip, ops

    P, C, G = 96, 5, 128
    logits, log_c, lib = _sampler_inputs(P, C, G, seed=11)
    consts = ops.nb_gene_consts(log_c)
    lib_ = hip.load()
    outs = []
    for use_ws in (False, True):
        sums = torch.zeros((C, G), device=DEV)
        samples = torch.empty((P * C, G), device=DEV)
        need = lib_.cradle_nb_sample_workspace_floats(P, C, G)
        assert need > 0
        ws = torch.empty(need, device=DEV) if use_ws else None
        hip.check(lib_.cradle_nb_sample_accumulate_ws(
            hip.ptr(logits), logits.stride(0), hip.ptr(lib), lib.numel(), hip.ptr(consts), hip.ptr(sums), hip.ptr(samples),
            P, C, G, 1, 0, 77, 0, hip.ptr(ws), need if use_ws else 0, hip.stream()), "nb_sample_accumulate_ws")
        torch.cuda.synchronize()
        outs.append((sums.clone(), samples.clone()))
    assert torch.equal(outs[0][1], outs[1][1])
    assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-4)


def _ate_setup(T, G, L, H, n_alt, seed):
    """Random-weight model/guide of the given size and a set of one- and two-guide perturbations."""
    from cradle_vae_amd import synthetic as syn
    from oracle import cradle_oracle as co

    g = torch.Generator().manual_seed(seed)
    cfg = dict(n_latent=L, n_treatments=T, n_phenos=G, n_qc=1, decoder_n_layers=1, decoder_n_hidden=H,
               basal_encoder_n_layers=1, basal_encoder_n_hidden=H, embedding_encoder_n_layers=4, embedding_encoder_n_hidden=H,
               mask_prior_prob=0.01, embedding_prior_scale=1.0, beta=0.5, alpha=None, n_particles=1)
    params = co.init_params(**{k: cfg[k] for k in cfg if k not in ("beta", "alpha", "n_particles")}, seed=seed,
                            log_x_mean=torch.zeros(G), log_x_std=torch.ones(G))
    if T > 4096:
        params.pop("guide.treatment_one_hot")  # eye(T): 400 MB at 10 k treatments, not a buffer of the module there
    D_alt = torch.zeros(n_alt, T)
    a, b = torch.randint(0, T, (n_alt,), generator=g), torch.randint(0, T, (n_alt,), generator=g)
    D_alt[torch.arange(n_alt), a] = 1.0
    two = torch.rand(n_alt, generator=g) < 0.4
    D_alt[torch.arange(n_alt)[two], b[two]] = 1.0
    D_ctl = torch.zeros(1, T)
    D_ctl[0, 0] = 1.0
    return cfg, params, D_ctl, D_alt


@pytest.mark.parametrize("method,G", [("mean", 64), ("perturbseq", 64), ("perturbseq", 67)])
def test_ate_gather_and_slab_path_equals_dense_oracle(method, G, monkeypatch):
    """The module path without the O(T^2) one-hot and without dense matmul(D, .): embedding encoder through the
    column lookup, evaluated in slabs of treatments, offsets through the dosage plan -- against the oracle, which
    does what the reference does (eye(T) input, dense products), on the same injected draws."""
    from cradle_vae_amd import guide as guide_mod
    from cradle_vae_amd.predictor import cradle_vae_Predictor
    from oracle import cradle_oracle as co

    T, L, H, n_alt, c, n_particles = 64, 16, 32, 20, 16, 48
    cfg, params, D_ctl, D_alt = _ate_setup(T, G, L, H, n_alt, seed=5)
    g = torch.Generator().manual_seed(77)
    chunks = []
    for _ in range(n_particles // c):
        chunks.append(dict(u_mask=torch.rand(c, T, L, generator=g), eps_E=torch.randn(c, T, L, generator=g),
                           eps_qc_E=torch.randn(c, 1, L, generator=g), z_control=torch.randn(c, 1, L, generator=g),
                           z_alt=torch.randn(n_alt, c, 1, L, generator=g)))
    lib = 1e4 * torch.ones(1)
    # oracle: dense, as the reference (fp64)
    p64 = {k: (v.double() if v.is_floating_point() else v) for k, v in params.items()}
    Xc, Xa = [], []
    for nz in chunks:
        n64 = {k: v.double() for k, v in nz.items()}
        _, s = co.guide_forward(p64, n64, c)
        s = dict(s)
        s["z_basal"] = n64["z_control"]
        Xc.append(co.nb_mean_good(p64, D_ctl.double(), s, lib.double())[:, 0])
        alt = []
        for i in range(n_alt):
            s["z_basal"] = n64["z_alt"][i]
            alt.append(co.nb_mean_good(p64, D_alt[i:i + 1].double(), s, lib.double())[:, 0])
        Xa.append(torch.stack(alt))
    ref = co.ate_from_draws(torch.cat(Xc), torch.cat(Xa, dim=1), method).numpy()

    case = dict(cfg=cfg, params={k: v.to(DEV) for k, v in params.items()})
    lm = build_loss_module(case)
    pred = cradle_vae_Predictor(lm.model, lm.guide)
    kw = dict(dosages_alt=D_alt, dosages_control=D_ctl, quality=torch.ones(1, 1), method=method, n_particles=n_particles,
              condition_values=dict(library_size=lib), batch_size=c, sampler="mean", noise_chunks=chunks)
    whole = pred.estimate_average_treatment_effects(**kw).X
    monkeypatch.setattr(guide_mod, "MAX_PLATED_ROWS", c * 10)  # 16 particles x 64 treatments -> slabs of 10 treatments
    slabbed = pred.estimate_average_treatment_effects(**kw).X
    scale = float(np.abs(ref).max())
    np.testing.assert_allclose(whole, ref, rtol=1e-4, atol=1e-4 * scale)
    np.testing.assert_allclose(slabbed, ref, rtol=1e-4, atol=1e-4 * scale)
    np.testing.assert_allclose(slabbed, whole, rtol=1e-6, atol=1e-6 * scale)  # slabs change no row's arithmetic


def test_ate_gene_order_of_the_sampler_changes_no_effect(monkeypatch):
    """ate.py hands the decoder's output layer to the sampler with its genes ordered by expected count (G > 256) and puts
    the sums back in the data's order: with the deterministic "sampler" the ordered call equals the plain one to fp32
    rounding, both methods; with counts drawn the two differ only by their draws (same distribution: their difference is
    within the Monte-Carlo error of either), and the ordered call is reproducible and independent of the chunking of the
    particles over ranks (the order is a function of the parameters and the seed alone)."""
    from cradle_vae_amd.predictor import cradle_vae_Predictor

    T, G, L, H, n_alt, c = 24, 301, 16, 32, 12, 64
    cfg, params, D_ctl, D_alt = _ate_setup(T, G, L, H, n_alt, seed=3)
    case = dict(cfg=cfg, params={k: v.to(DEV) for k, v in params.items()})
    lm = build_loss_module(case)
    pred = cradle_vae_Predictor(lm.model, lm.guide)
    lib = 2e3 * torch.ones(1)
    for method in ("mean", "perturbseq"):
        kw = dict(dosages_alt=D_alt, dosages_control=D_ctl, quality=torch.ones(1, 1), method=method, n_particles=1024,
                  condition_values=dict(library_size=lib), batch_size=c, seed=4)
        monkeypatch.setenv("CRADLE_ATE_SORT_GENES", "0")
        plain = pred.estimate_average_treatment_effects(sampler="mean", **kw).X
        drawn_plain = pred.estimate_average_treatment_effects(sampler="nb", **kw).X
        monkeypatch.setenv("CRADLE_ATE_SORT_GENES", "2")
        ordered = pred.estimate_average_treatment_effects(sampler="mean", **kw).X
        monkeypatch.setenv("CRADLE_ATE_SORT_GENES", "1")  # the default: counts drawn, G > 256 -> ordered
        drawn = pred.estimate_average_treatment_effects(sampler="nb", **kw).X
        drawn_again = pred.estimate_average_treatment_effects(sampler="nb", **kw).X
        scale = float(np.abs(plain).max())
        # (an effect is a difference of two means of order 10: 1e-5 of the effects' scale is fp32 rounding of those means)
        np.testing.assert_allclose(ordered, plain, rtol=0, atol=1e-5 * scale)
        assert np.array_equal(drawn, drawn_again) and not np.array_equal(drawn, drawn_plain)
        # both sampled estimates scatter around the mean-sampler's value alike (1024 particles)
        e1, e0 = drawn - plain, drawn_plain - plain
        if method == "mean":  # (perturbseq: the transform of a draw is not the transform of its mean)
            assert abs(e1.mean()) < 4 * e0.std() / np.sqrt(e0.size) + 1e-3 * scale, (e1.mean(), e0.std())
        assert 0.7 < e1.std() / e0.std() < 1.4, (e1.std(), e0.std())


def test_ate_draws_beside_the_next_decoder_pass_change_no_number(monkeypatch):
    """ate.py launches a slab's draws on a side stream beside the next slab's decoder pass.  Launches on one stream keep
    their order and every (condition, gene) sum is added to by that stream alone: the overlapped call returns the bits
    of the one-stream call (CRADLE_ATE_OVERLAP=0), several slabs per chunk and several chunks, both samplers."""
    import cradle_vae_amd.ate as ate_mod
    from cradle_vae_amd.predictor import cradle_vae_Predictor

    T, G, L, H, n_alt, c = 24, 301, 16, 32, 12, 64
    cfg, params, D_ctl, D_alt = _ate_setup(T, G, L, H, n_alt, seed=5)
    case = dict(cfg=cfg, params={k: v.to(DEV) for k, v in params.items()})
    lm = build_loss_module(case)
    pred = cradle_vae_Predictor(lm.model, lm.guide)
    monkeypatch.setattr(ate_mod, "MAX_ROWS_PER_LAUNCH", 4 * c)  # 13 conditions in slabs of 4: four sampler launches a chunk
    for sampler in ("nb", "mean"):
        kw = dict(dosages_alt=D_alt, dosages_control=D_ctl, quality=torch.ones(1, 1), method="perturbseq", n_particles=5 * c,
                  condition_values=dict(library_size=2e3 * torch.ones(1)), batch_size=c, seed=9, sampler=sampler)
        monkeypatch.setenv("CRADLE_ATE_OVERLAP", "0")
        one_stream = pred.estimate_average_treatment_effects(**kw).X
        monkeypatch.setenv("CRADLE_ATE_OVERLAP", "1")
        for _ in range(3):
            assert np.array_equal(pred.estimate_average_treatment_effects(**kw).X, one_stream)


def test_ate_with_ten_thousand_treatments_fits_in_two_gigabytes():
    """BASELINE config 5's predictor shape: T = 10 000 treatments, G = 8000 genes, 128-particle chunks.  As the
    reference writes it this needs a (128 * 10 000) x 10 100 one-hot encoder input (52 GB) and a dense
    (n, 10 000) x (128, 10 000, L) product per call; here the peak stays under 2 GB beyond the parameters."""
    from cradle_vae_amd.predictor import cradle_vae_Predictor

    T, G, L, H, n_alt, c = 10000, 8000, 100, 400, 96, 128
    cfg, params, D_ctl, D_alt = _ate_setup(T, G, L, H, n_alt, seed=9)
    case = dict(cfg=cfg, params={k: v.to(DEV) for k, v in params.items()})
    lm = build_loss_module(case)
    del case, params
    pred = cradle_vae_Predictor(lm.model, lm.guide)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    res = pred.estimate_average_treatment_effects(
        dosages_alt=D_alt, dosages_control=D_ctl, quality=torch.ones(1, 1), method="perturbseq", n_particles=c,
        condition_values=dict(library_size=1e4 * torch.ones(1)), batch_size=c, seed=3)
    torch.cuda.synchronize()
    extra = torch.cuda.max_memory_allocated() - base
    assert res.X.shape == (n_alt, G) and np.isfinite(res.X).all() and np.abs(res.X).max() > 0
    assert extra < 2 * 1024**3, f"{extra / 1024**3:.2f} GiB beyond the module"


def test_ate_in_the_split_bf16_gemm_mode_stays_within_tolerance():
    """`with ops.gemm_precision("bf16x3")`: the predictor's large GEMMs (here the decoder's output layer, 2688 x 5000 x
    400 = 10.8 GFLOP per chunk) run on the split-bf16 kernels; the effects move by ~1e-5 of their scale (bar: 1e-4,
    north_star's ELBO tolerance), and the mode really is taken (the results are not bit-identical)."""
    from cradle_vae_amd import ops
    from cradle_vae_amd.predictor import cradle_vae_Predictor

    T, G, L, H, n_alt, c = 32, 5000, 16, 400, 20, 128
    cfg, params, D_ctl, D_alt = _ate_setup(T, G, L, H, n_alt, seed=11)
    lm = build_loss_module(dict(cfg=cfg, params={k: v.to(DEV) for k, v in params.items()}))
    pred = cradle_vae_Predictor(lm.model, lm.guide)
    kw = dict(dosages_alt=D_alt, dosages_control=D_ctl, quality=torch.ones(1, 1), method="perturbseq", n_particles=c,
              condition_values=dict(library_size=1e4 * torch.ones(1)), batch_size=c, sampler="mean", seed=3)
    exact = pred.estimate_average_treatment_effects(**kw).X
    with ops.gemm_precision("bf16x3"):
        split = pred.estimate_average_treatment_effects(**kw).X
    again = pred.estimate_average_treatment_effects(**kw).X
    assert np.array_equal(exact, again)  # the switch is scoped
    scale = float(np.abs(exact).max())
    err = float(np.abs(split - exact).max())
    assert 0 < err <= 1e-4 * scale, (err, scale)
