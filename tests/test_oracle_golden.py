"""CPU: the oracle restatement reproduces every golden fixture the reference produced."""
import math

import pytest
import torch

from oracle import cradle_oracle as co
from tests import golden_io as gio


def _run(case, dtype):
    b, cfg = case["batch"], case["cfg"]
    return co.elbo_loss_and_grads(
        case["params"], X=b["X"], D=b["D"], qc=b["qc"], cfX=b["cfX"], library_size=b["library_size"],
        D_obs_counts=b["D_obs_counts"], qc_obs_counts=b["qc_obs_counts"], noise=case["noise"],
        cf_noise=case["cf_noise"], n_particles=cfg["n_particles"], beta=cfg["beta"], alpha=cfg["alpha"],
    )


@pytest.mark.parametrize("name", gio.elbo_cases())
@pytest.mark.parametrize("tag,dtype,rtol", [("f64", torch.float64, 1e-11), ("f32", torch.float32, 2e-5)])
def test_elbo_matches_reference(name, tag, dtype, rtol):
    case = gio.load_elbo(name, dtype)
    loss, metrics, _, grads = _run(case, dtype)
    ref = case[tag]
    assert abs(float(loss) - ref["loss"]) <= rtol * max(1.0, abs(ref["loss"]))
    for k, v in ref["metrics"].items():
        assert abs(float(metrics[k]) - v) <= rtol * max(1.0, abs(v)), k
    for k, g in ref["grads"].items():
        scale = max(float(g.abs().max()), 1e-6)
        assert float((grads[k].to(g.dtype) - g).abs().max()) <= 50 * rtol * scale, k


def test_gamma_poisson_closed_form():
    """pyro's GammaPoisson is not installed: pin the restated log_prob to scipy's nbinom."""
    from scipy.stats import nbinom

    g = torch.Generator().manual_seed(0)
    c = torch.exp(torch.randn(64, generator=g, dtype=torch.float64))
    rate = torch.exp(2 * torch.randn(64, generator=g, dtype=torch.float64))
    x = torch.poisson(5 * torch.rand(64, generator=g, dtype=torch.float64))
    lp = co.gamma_poisson_log_prob(x, c, rate)
    ref = nbinom(n=c.numpy(), p=(rate / (1 + rate)).numpy()).logpmf(x.numpy())
    assert float((lp - torch.from_numpy(ref)).abs().max()) < 1e-10


def test_predictor_fixture():
    case = gio.load_predictor("tiny")
    p, b, cfg = case["params"], case["batch"], case["cfg"]
    # ---- IWELBO
    iw = case["iwelbo"]
    P = int(iw["n_particles"])
    _, plated = co.guide_forward(p, {k[len("noise/"):]: v for k, v in iw.items() if k.startswith("noise/")}, P)
    plated = {k: plated[k] for k in ("mask", "E", "qc_E")}
    vals = []
    bs = int(iw["batch"])
    for s in range(0, cfg["n_cells"], bs):
        sl = slice(s, s + bs)
        vals.append(co.predictive_iwelbo(
            p, X=b["X"][sl], D=b["D"][sl], qc=b["qc"][sl], library_size=b["library_size"][sl], plated=plated,
            eps_z=iw["noise/eps_z"][:, sl], n_particles=P))
    vals = torch.cat(vals)
    assert torch.allclose(vals, iw["values"].to(vals.dtype), rtol=2e-5, atol=1e-4)
    # ---- ATE (count sampler = its mean in the fixture)
    ate = case["ate"]
    n_particles, chunk = int(ate["n_particles"]), int(ate["chunk"])
    lib = 1e4 * torch.ones((1,))
    Xc, Xa = [], []
    for ci, i in enumerate(range(0, n_particles, chunk)):
        nz = {k[len(f"noise{ci}/"):]: v for k, v in ate.items() if k.startswith(f"noise{ci}/")}
        c = nz["u_mask"].shape[0]
        _, pl = co.guide_forward(p, nz, c)
        s = dict(pl)
        s["z_basal"] = nz["z_control"]
        Xc.append(co.nb_mean_good(p, ate["D_control"], s, lib)[:, 0])
        per_t = []
        for t in range(ate["D_alt"].shape[0]):
            s["z_basal"] = nz["z_alt"][t]
            per_t.append(co.nb_mean_good(p, ate["D_alt"][t:t + 1], s, lib)[:, 0])
        Xa.append(torch.stack(per_t))
    Xc, Xa = torch.cat(Xc), torch.cat(Xa, dim=1)
    for method in ("mean", "perturbseq"):
        got = co.ate_from_draws(Xc, Xa, method)
        ref = ate[method]
        assert torch.allclose(got, ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max())), method


@pytest.mark.parametrize("name", gio.elbo_cases())
@pytest.mark.parametrize("tag,dtype,rtol", [("f64", torch.float64, 1e-12), ("f32", torch.float32, 2e-6)])
def test_guide_samples_match_reference(name, tag, dtype, rtol):
    """The straight-through mask is a hard 0/1 tensor: bit-exact against the reference's own draw;
    E, qc_E, z_basal are reparameterised Normal samples (floating point)."""
    case = gio.load_elbo(name, dtype)
    b, cfg = case["batch"], case["cfg"]
    _, s = co.guide_forward(case["params"], case["noise"], cfg["n_particles"], X=b["X"], D=b["D"], qc=b["qc"])
    ref = case[tag]["samples"]
    assert set(ref) == {"mask", "E", "qc_E", "z_basal"}
    assert torch.equal(s["mask"].detach().to(ref["mask"].dtype), ref["mask"])
    for k in ("E", "qc_E", "z_basal"):
        r = ref[k]
        err = float((s[k].detach().to(r.dtype) - r).abs().max()) / max(float(r.abs().max()), 1e-6)
        assert err <= 50 * rtol, (k, err)


def test_sampled_ate_follows_the_references_draws():
    """The "as written" ATE port (what bench.py times as the CPU baseline of the predictor) against the reference's own
    estimate_model_average_treatment_effect run WITH its count draws under torch.manual_seed (fixture:
    oracle/make_ate_sampled_golden.py): same generator, same consumption order -> the same effects, draw for draw.
    (CPU generator streams are a property of the torch build; the fixture was written with the build in this image.)"""
    import os

    import numpy as np

    z = np.load(os.path.join(gio.GOLDEN, "ate_sampled_tiny.npz"), allow_pickle=False)
    params = {k[len("params/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("params/")}
    T, Q = params["model.p_E_loc"].shape[0], params["model.p_qc_E_loc"].shape[0]
    for method in ("mean", "perturbseq"):
        for seed in (0, 5):
            ref = torch.from_numpy(z[f"ate/{method}/seed{seed}"])
            got, calls = co.model_average_treatment_effect(
                params, dosages_alt=torch.from_numpy(z["D_alt"]), dosages_control=torch.from_numpy(z["D_control"]),
                quality=torch.ones(1, Q), library_size=1e4 * torch.ones(1), n_particles=int(z["n_particles"]),
                method=method, batch_size=int(z["chunk"]), seed=seed)
            assert calls == 3 * (1 + ref.shape[0])  # 20 particles in chunks of 8: 3 chunks x (control + 4 alternatives)
            assert float((got - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())), (method, seed)
