"""GPU parity of the fused HIP ELBO step (cradle_elbo_step) against the reference's golden
fixtures (fp64 run of the reference): loss, every loss term, every parameter gradient."""
import pytest
import torch

from tests import golden_io as gio
from tests.test_modular_gpu import DEV, build_loss_module, check_against_golden

pytestmark = pytest.mark.gpu


def run_fused(lm, case):
    b, cfg = case["batch"], case["cfg"]
    lm.step = "fused"
    for p in lm.parameters():
        p.grad = None
    loss, metrics = lm.loss(
        X=b["X"], D=b["D"], D_obs_counts=b["D_obs_counts"], qc=b["qc"], qc_obs_counts=b["qc_obs_counts"],
        cfX=b["cfX"], condition_values={"library_size": b["library_size"]}, n_particles=cfg["n_particles"],
        noise=case["noise"], cf_noise=case["cf_noise"])
    loss.backward()
    lm._fused.check_flags()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in lm.named_parameters()}
    return loss, metrics, grads


@pytest.mark.parametrize("name", [n for n in gio.elbo_cases() if n != "tiny_q2"] + ["tiny_q2"])
def test_fused_step_matches_reference(name):
    case = gio.load_elbo(name, torch.float32, DEV)
    lm = build_loss_module(case)
    loss, metrics, grads = run_fused(lm, case)
    check_against_golden(case, loss, metrics, grads)


@pytest.mark.parametrize("kmax", [8, 2])
def test_fused_equals_modular_at_replogle_like_shape(kmax):
    """T > N, L = 100, two-layer-free encoder: sizes the golden fixtures do not reach.  kmax=2 (the
    data set's true bound) makes N*kmax < T, i.e. the step runs q(M), q(E|M) on the compact axis of
    treatments present in the batch; kmax=8 keeps the full axis."""
    import dataclasses

    from cradle_vae_amd import synthetic as syn
    from oracle import cradle_oracle as co

    shape = dataclasses.replace(syn.REPLOGLE, n_cells=96, n_phenos=1200, n_treatments=700, n_particles=3)
    b = syn.make_batch(shape, seed=5, device=DEV)
    cfg = dict(
        n_latent=shape.n_latent, n_treatments=shape.n_treatments, n_phenos=shape.n_phenos, n_qc=1,
        decoder_n_layers=1, decoder_n_hidden=shape.decoder_n_hidden, basal_encoder_n_layers=1,
        basal_encoder_n_hidden=shape.basal_encoder_n_hidden, embedding_encoder_n_layers=4,
        embedding_encoder_n_hidden=shape.embedding_encoder_n_hidden, mask_prior_prob=shape.mask_prior_prob,
        embedding_prior_scale=1.0, beta=shape.beta, alpha=None, n_particles=shape.n_particles)
    params = co.init_params(**{k: cfg[k] for k in cfg if k not in ("beta", "alpha", "n_particles")},
                            log_x_mean=b["log_x_mean"].cpu(), log_x_std=b["log_x_std"].cpu(), seed=9)
    case = dict(cfg=cfg, params={k: v.to(DEV) for k, v in params.items()}, batch=b,
                noise=syn.make_noise(shape, 1, DEV), cf_noise=syn.make_noise(shape, 2, DEV))
    lm = build_loss_module(case)
    lm.max_dosages_per_cell = kmax
    loss_f, met_f, g_f = run_fused(lm, case)
    assert lm._fused.desc.kmax == kmax and (lm._fused.desc.N * kmax < lm._fused.desc.T) == (kmax == 2)
    g_f = {k: v.clone() for k, v in g_f.items()}
    lm.step = "modular"
    for p in lm.parameters():
        p.grad = None
    loss_m, met_m = lm.loss(
        X=b["X"], D=b["D"], D_obs_counts=b["D_obs_counts"], qc=b["qc"], qc_obs_counts=b["qc_obs_counts"],
        cfX=b["cfX"], condition_values={"library_size": b["library_size"]}, n_particles=cfg["n_particles"],
        noise=case["noise"], cf_noise=case["cf_noise"])
    loss_m.backward()
    assert abs(float(loss_f) - float(loss_m)) <= 2e-5 * abs(float(loss_m)), (float(loss_f), float(loss_m))
    for k in met_m:
        assert abs(float(met_f[k]) - float(met_m[k])) <= 2e-5 * max(1.0, abs(float(met_m[k]))), k
    for k, p in lm.named_parameters():
        ref = p.grad
        err = float((g_f[k] - ref).abs().max()) / max(float(ref.abs().max()), 1e-8)
        assert err < 2e-3, (k, err)


def test_compact_axis_overflow_is_reported():
    """More non-zero dosages in a cell than the declared bound: the step must say so, not mis-score."""
    import dataclasses

    from cradle_vae_amd import hip
    from cradle_vae_amd import synthetic as syn
    from oracle import cradle_oracle as co

    shape = dataclasses.replace(syn.TINY, n_cells=4, n_treatments=12)
    b = syn.make_batch(shape, seed=3, device=DEV)
    b["D"][0, :3] = 1.0  # three dosages in one cell
    cfg = dict(
        n_latent=shape.n_latent, n_treatments=shape.n_treatments, n_phenos=shape.n_phenos, n_qc=1,
        decoder_n_layers=1, decoder_n_hidden=8, basal_encoder_n_layers=1, basal_encoder_n_hidden=8,
        embedding_encoder_n_layers=2, embedding_encoder_n_hidden=8, mask_prior_prob=shape.mask_prior_prob,
        embedding_prior_scale=1.0, beta=shape.beta, alpha=None, n_particles=shape.n_particles)
    params = co.init_params(**{k: cfg[k] for k in cfg if k not in ("beta", "alpha", "n_particles")},
                            log_x_mean=b["log_x_mean"].cpu(), log_x_std=b["log_x_std"].cpu(), seed=1)
    case = dict(cfg=cfg, params={k: v.to(DEV) for k, v in params.items()}, batch=b,
                noise=syn.make_noise(shape, 1, DEV), cf_noise=syn.make_noise(shape, 2, DEV))
    lm = build_loss_module(case)
    lm.max_dosages_per_cell = 2
    with pytest.raises(hip.CradleHipError, match="kmax"):
        run_fused(lm, case)


def test_forward_only_matches_training_forward():
    case = gio.load_elbo("mid", torch.float32, DEV)
    lm = build_loss_module(case)
    loss, metrics, _ = run_fused(lm, case)
    b, cfg = case["batch"], case["cfg"]
    with torch.no_grad():
        loss2, metrics2 = lm.loss(
            X=b["X"], D=b["D"], D_obs_counts=b["D_obs_counts"], qc=b["qc"], qc_obs_counts=b["qc_obs_counts"],
            cfX=b["cfX"], condition_values={"library_size": b["library_size"]}, n_particles=cfg["n_particles"],
            noise=case["noise"], cf_noise=case["cf_noise"])
    assert abs(float(loss) - float(loss2)) <= 1e-6 * abs(float(loss))


def _step_outputs(fs, case, **kw):
    b, cfg = case["batch"], case["cfg"]
    out = fs.run(X=b["X"], D=b["D"], D_obs_counts=b["D_obs_counts"], qc=b["qc"], qc_obs_counts=b["qc_obs_counts"],
                 cfX=b["cfX"], library_size=b["library_size"], n_particles=cfg["n_particles"], noise=case["noise"],
                 cf_noise=case["cf_noise"], **kw)
    torch.cuda.synchronize()
    return out.clone(), fs.arena.grad.clone()


def test_graph_replay_equals_eager_launch():
    """cradle_elbo_step_graphed (whole-step graph, and the two-graph form with K1 timed eagerly)
    against cradle_elbo_step; replays, a second batch at new addresses, input staging."""
    from cradle_vae_amd import hip

    case = gio.load_elbo("mid", torch.float32, DEV)
    lm = build_loss_module(case)
    fs = lm._fused_step()
    fs.use_graph = False
    out_e, g_e = _step_outputs(fs, case)
    fs.use_graph = True
    fs._recaptures = 0
    for _ in range(3):  # capture, then replays
        out_g, g_g = _step_outputs(fs, case)
        assert torch.allclose(out_g, out_e, rtol=1e-6, atol=1e-6)
        assert float((g_g - g_e).abs().max()) <= 1e-6 * float(g_e.abs().max())
    assert fs._recaptures == 1
    fs.k1_events = []
    out_s, g_s = _step_outputs(fs, case)
    assert torch.allclose(out_s, out_e, rtol=1e-6, atol=1e-6) and torch.allclose(g_s, g_g, rtol=1e-5, atol=1e-7)
    assert len(fs.k1_events) == 1 and 0.0 < fs.k1_events[0].elapsed_ms() < 50.0
    fs.k1_events = None
    # forward-only graph
    out_f, _ = _step_outputs(fs, case, compute_grads=False)
    assert torch.allclose(out_f, out_e, rtol=1e-6, atol=1e-6)
    # another batch (new addresses, new values) and back: both graphs stay cached
    case2 = gio.load_elbo("mid", torch.float32, DEV)
    case2["batch"] = dict(case2["batch"])
    case2["batch"]["X"] = torch.roll(case2["batch"]["X"], 1, dims=0).contiguous()
    fs.use_graph = False
    out2_e, g2_e = _step_outputs(fs, case2)
    fs.use_graph = True
    n0 = fs._recaptures
    out2_g, g2_g = _step_outputs(fs, case2)
    assert torch.allclose(out2_g, out2_e, rtol=1e-6, atol=1e-6) and fs._recaptures == n0 + 1
    assert not torch.allclose(out2_g, out_e, rtol=1e-4)
    out_b, _ = _step_outputs(fs, case)
    assert torch.allclose(out_b, out_e, rtol=1e-6, atol=1e-6) and fs._recaptures == n0 + 1
    # staged inputs: addresses never reach the graph, so no further capture after the staging one
    fs.stage_inputs = True
    out_st, g_st = _step_outputs(fs, case2)
    n1 = fs._recaptures
    out_st1, _ = _step_outputs(fs, case)
    assert fs._recaptures == n1
    assert torch.allclose(out_st, out2_e, rtol=1e-6, atol=1e-6) and torch.allclose(out_st1, out_e, rtol=1e-6, atol=1e-6)
    assert float((g_st - g2_e).abs().max()) <= 1e-6 * float(g2_e.abs().max())
    assert isinstance(hip.load().cradle_last_error(), bytes)


@pytest.mark.parametrize("name", gio.elbo_cases())
def test_fused_step_draws_match_reference(name):
    """The hot path's own intermediates against the reference guide's draws (same injected noise):
    the straight-through mask bit-exact, E / qc_E / z_basal within fp32 GEMM tolerance."""
    case = gio.load_elbo(name, torch.float32, DEV)
    lm = build_loss_module(case)
    run_fused(lm, case)
    fs, cfg = lm._fused, case["cfg"]
    P, N, T, L, Q = cfg["n_particles"], case["batch"]["X"].shape[0], cfg["n_treatments"], cfg["n_latent"], cfg["n_qc"]
    ref = case["f64"]["samples"]
    mask = fs.intermediate("mask").view(2 * P, T, L)[:P].cpu().double()  # rows P.. are the counterfactual pass
    assert torch.equal(mask, ref["mask"].double())
    got = dict(E=fs.intermediate("E").view(2 * P, T, L)[:P], qc_E=fs.intermediate("qc_E").view(2 * P, Q, L)[:P],
               z_basal=fs.intermediate("z_offset").view(P, N, 2 * L)[:, :, :L])
    for k, v in got.items():
        r = ref[k].double()
        err = float((v.cpu().double() - r).abs().max()) / max(float(r.abs().max()), 1e-6)
        assert err <= 2e-5, (k, err)
