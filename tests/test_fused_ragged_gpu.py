"""Fused HIP ELBO step against the CPU oracle (fp64) on ragged shapes the golden fixtures do not
hold: odd latent / hidden widths (the non-vectorised GEMM loaders, partial tiles everywhere), a
single cell, all-control batches, treatments without any cell, the compact treatment axis with
unused slots, three encoder layers.  Same bar as the golden tests: loss and every loss term within
1e-4 relative, every gradient within 2e-3 of its largest entry."""
import dataclasses

import pytest
import torch

from tests.test_fused_step_gpu import run_fused
from tests.test_modular_gpu import DEV, GRAD_L2_RTOL, build_loss_module

pytestmark = pytest.mark.gpu

# name: (N, P, G, T, L, Hd, He, Hm, layers(dec, enc, emb), kmax, tweak)
CASES = {
    "odd_everything": (37, 2, 36, 11, 7, 13, 10, 9, (1, 1, 2), 8, None),
    "single_cell": (1, 3, 8, 5, 3, 6, 6, 6, (1, 1, 1), 8, None),
    "all_control": (9, 2, 12, 4, 5, 8, 8, 8, (2, 1, 2), 8, "control"),
    "compact_sparse": (5, 2, 16, 40, 3, 8, 8, 8, (1, 2, 3), 2, None),
    "compact_all_control": (6, 2, 16, 30, 4, 8, 8, 8, (1, 1, 2), 2, "control"),
    "no_counterfactuals_but_one": (12, 2, 20, 6, 6, 8, 12, 8, (1, 1, 2), 8, "one_cf"),
}


def make_case(name):
    from cradle_vae_amd import synthetic as syn
    from oracle import cradle_oracle as co

    N, P, G, T, L, Hd, He, Hm, (ld, le, lm_), kmax, tweak = CASES[name]
    shape = dataclasses.replace(
        syn.TINY, n_cells=N, n_particles=P, n_phenos=G, n_treatments=T, n_latent=L, decoder_n_hidden=Hd,
        basal_encoder_n_hidden=He, embedding_encoder_n_hidden=Hm)
    b = syn.make_batch(shape, seed=11)
    if tweak == "control":
        b["D"].zero_()
    if tweak == "one_cf":
        b["cfX"][1:] = 0.0
        b["cfX"][0] = b["X"][0] + 1.0
    cfg = dict(
        n_latent=L, n_treatments=T, n_phenos=G, n_qc=1, decoder_n_layers=ld, decoder_n_hidden=Hd,
        basal_encoder_n_layers=le, basal_encoder_n_hidden=He, embedding_encoder_n_layers=lm_,
        embedding_encoder_n_hidden=Hm, mask_prior_prob=shape.mask_prior_prob, embedding_prior_scale=1.0,
        beta=0.7, alpha=0.6 if name in ("single_cell", "all_control", "compact_sparse") else None, n_particles=P)
    # (alpha=None is the reference's qc-fraction weighting, loss_modules.py:137-139, which zeroes the
    # reconstruction term of a batch without a QC-failed cell: those cases get a fixed alpha)
    params = co.init_params(**{k: cfg[k] for k in cfg if k not in ("beta", "alpha", "n_particles")},
                            log_x_mean=b["log_x_mean"], log_x_std=b["log_x_std"], seed=4)
    noise, cf_noise = syn.make_noise(shape, 1), syn.make_noise(shape, 2)
    return cfg, params, b, noise, cf_noise, kmax


@pytest.mark.parametrize("name", list(CASES))
def test_fused_step_matches_oracle_on_ragged_shapes(name):
    from oracle import cradle_oracle as co

    cfg, params, b, noise, cf_noise, kmax = make_case(name)
    f64 = lambda d: {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()}
    ref_loss, ref_metrics, _, ref_grads = co.elbo_loss_and_grads(
        f64(params), X=b["X"].double(), D=b["D"].double(), qc=b["qc"].double(), cfX=b["cfX"].double(),
        library_size=b["library_size"].double(), D_obs_counts=b["D_obs_counts"].double(),
        qc_obs_counts=b["qc_obs_counts"].double(), noise=f64(noise), cf_noise=f64(cf_noise),
        n_particles=cfg["n_particles"], beta=cfg["beta"], alpha=cfg["alpha"])
    assert torch.isfinite(ref_loss)

    dev = lambda d: {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in d.items()}
    case = dict(cfg=cfg, params=dev(params), batch=dev(b), noise=dev(noise), cf_noise=dev(cf_noise))
    lm = build_loss_module(case)
    lm.max_dosages_per_cell = kmax
    loss, metrics, grads = run_fused(lm, case)
    d = lm._fused.desc
    assert (d.N * d.kmax < d.T) == name.startswith("compact")

    rtol = 1e-4
    assert abs(float(loss.detach()) - float(ref_loss)) <= rtol * max(1.0, abs(float(ref_loss))), (float(loss), float(ref_loss))
    for k, v in ref_metrics.items():
        assert abs(float(metrics[k]) - float(v)) <= rtol * max(1.0, abs(float(v))), (k, float(metrics[k]), float(v))
    for k, g in ref_grads.items():
        got = grads[k].detach().cpu().double()
        err = float((got - g).abs().max()) / max(float(g.abs().max()), 1e-8)
        assert err <= 2e-3, (k, err)
        l2 = float((got - g).norm()) / max(float(g.norm()), 1e-12)
        assert l2 <= GRAD_L2_RTOL, (k, l2)
