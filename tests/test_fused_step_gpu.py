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


def _replogle_like_case(n_phenos=1200):
    import dataclasses

    from cradle_vae_amd import synthetic as syn
    from oracle import cradle_oracle as co

    shape = dataclasses.replace(syn.REPLOGLE, n_cells=96, n_phenos=n_phenos, n_treatments=700, n_particles=3)
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
    return case, cfg, b


@pytest.mark.parametrize("kmax,n_phenos", [(8, 1200), (2, 1200), (2, 1201)])
def test_fused_equals_modular_at_replogle_like_shape(kmax, n_phenos):
    """T > N, L = 100, two-layer-free encoder: sizes the golden fixtures do not reach.  kmax=2 (the
    data set's true bound) makes N*kmax < T, i.e. the step runs q(M), q(E|M) on the compact axis of
    treatments present in the batch; kmax=8 keeps the full axis.  n_phenos=1201: a gene count that is not a
    multiple of 4 through both paths (the modular one hands K1 row-padded views of the logits and counts)."""
    case, cfg, b = _replogle_like_case(n_phenos)
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
        l2 = float((g_f[k] - ref).norm()) / max(float(ref.norm()), 1e-12)
        assert l2 < 1e-4, (k, l2)


def test_treatment_axis_sized_by_the_treatments_present():
    """A loader that has counted the distinct treatments of a batch tells the step (cradle_step_desc.max_present): the
    treatment axis of q(M), q(E|M) and the embedding encoder shrinks from N*kmax slots to that count (rounded up to 32)
    and nothing else changes; announcing fewer than there are is reported, not mis-scored."""
    from cradle_vae_amd import hip

    case, cfg, b = _replogle_like_case()
    lm = build_loss_module(case)
    lm.max_dosages_per_cell = 2
    fs = lm._fused_step()
    out0, g0 = _step_outputs(fs, case)
    assert fs.desc.max_present == 0
    n = int((b["D"] != 0).any(dim=0).sum())
    assert 64 < n < 160  # 96 cells x <= 2 dosages: far fewer distinct treatments than the 192 slots of the bound
    out1, g1 = _step_outputs(fs, case, treatments_present=n)
    assert fs.desc.max_present == (n + 31) // 32 * 32 < 192
    assert torch.allclose(out1[:7], out0[:7], rtol=2e-6, atol=1e-6)
    assert float((g1 - g0).abs().max()) <= 1e-5 * float(g0.abs().max())
    fs.check_flags()
    _step_outputs(fs, case, treatments_present=33)  # 64 slots for more than 64 treatments
    with pytest.raises(hip.CradleHipError, match="treatments_present"):
        fs.check_flags()


def test_compact_axis_draws_only_the_noise_it_reads():
    """The step's own draws on the compact axis: u_mask / eps_E are drawn for the treatments in the batch only, once the
    slot map exists -- the same values as the full draw (keyed by treatment, not by slot): same key, same loss and
    gradients whether the axis is compact (kmax = 2: 192 slots of 700), counted (treatments_present) or full."""
    case, cfg, b = _replogle_like_case()
    outs = []
    for kmax, count in ((8, None), (2, None), (2, int((b["D"] != 0).any(dim=0).sum()))):
        lm = build_loss_module(case)
        lm.max_dosages_per_cell = kmax
        fs = lm._fused_step()
        out = fs.run(X=b["X"], D=b["D"], D_obs_counts=b["D_obs_counts"], qc=b["qc"], qc_obs_counts=b["qc_obs_counts"],
                     cfX=b["cfX"], library_size=b["library_size"], n_particles=cfg["n_particles"], rng_key=(11, 12, 5),
                     treatments_present=count)
        torch.cuda.synchronize()
        assert (fs.desc.N * fs.desc.kmax < fs.desc.T) == (kmax == 2) and fs.desc.u_mask is None
        outs.append((out.clone(), fs.arena.grad.clone()))
    for out, g in outs[1:]:
        assert torch.isfinite(out[:7]).all()
        assert torch.allclose(out[:7], outs[0][0][:7], rtol=2e-6, atol=1e-6)
        assert float((g - outs[0][1]).abs().max()) <= 1e-5 * float(outs[0][1].abs().max())


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


def test_rejected_input_never_reaches_the_parameters():
    """Fractional qc in the middle of training (the fused step scores one decoder variant per cell and cannot
    weigh two): the step's loss is NaN, cradle_adam_step -- handed the step's flags word -- leaves parameters and
    moments untouched, and the host check names the reason.  The following valid step trains on."""
    from cradle_vae_amd import hip
    from cradle_vae_amd.trainer import TrainStep

    case = gio.load_elbo("mid", torch.float32, DEV)
    lm = build_loss_module(case)
    st = TrainStep(lm, lr=1e-3, gradient_clip_norm=100.0)
    assert st.fused is not None
    batch = dict(case["batch"])
    P = case["cfg"]["n_particles"]
    for _ in range(3):
        loss = st(batch, n_particles=P)
    assert torch.isfinite(loss)
    torch.cuda.synchronize()
    before = st.fused.arena.flat.clone(), st.exp_avg.clone(), st.exp_avg_sq.clone()
    bad = dict(batch)
    bad["qc"] = batch["qc"].clone()
    bad["qc"][0, 0] = 0.25
    loss_bad = st(bad, n_particles=P)  # no host check on this call (state already seen, not the 256th)
    torch.cuda.synchronize()
    assert torch.isnan(loss_bad)
    for a, b in zip(before, (st.fused.arena.flat, st.exp_avg, st.exp_avg_sq)):
        assert torch.equal(a, b)
    loss_ok = st(batch, n_particles=P)  # a valid step again: flags clear, parameters move
    torch.cuda.synchronize()
    assert torch.isfinite(loss_ok) and not torch.equal(before[0], st.fused.arena.flat)
    with pytest.raises(hip.CradleHipError, match="rejected"):  # the sticky count still tells the host
        st.fused.check_flags()
    st.fused.check_flags()  # reported once
    with pytest.raises(hip.CradleHipError, match="qc"):  # forward-only calls are checked at once
        with torch.no_grad():
            lm.loss(X=bad["X"], D=bad["D"], D_obs_counts=bad["D_obs_counts"], qc=bad["qc"],
                    qc_obs_counts=bad["qc_obs_counts"], cfX=bad["cfX"],
                    condition_values={"library_size": bad["library_size"]}, n_particles=P)


def test_step_draws_its_own_noise_when_given_none():
    """No noise tensors, no torch generators: the step draws uniforms and standard normals itself (Philox keyed by
    seed_plated / seed_local / step).  Same key -> same draws and the same loss; the plated draws depend on
    seed_plated only (identical across data-parallel ranks), z_basal's on seed_local; moments as they should be."""
    from cradle_vae_amd import synthetic as syn

    case = gio.load_elbo("mid", torch.float32, DEV)
    lm = build_loss_module(case)
    fs = lm._fused_step()
    b, P = case["batch"], case["cfg"]["n_particles"]
    kw = dict(X=b["X"], D=b["D"], D_obs_counts=b["D_obs_counts"], qc=b["qc"], qc_obs_counts=b["qc_obs_counts"],
              cfX=b["cfX"], library_size=b["library_size"], n_particles=P)

    def draw(seed_plated, seed_local, step):
        fs.seed_plated, fs.seed_local, fs.rng_step = seed_plated, seed_local, step
        out = fs.run(**kw).clone()
        return out, {k: fs.intermediate(k) for k in ("u_mask", "eps_E", "eps_qc_E", "eps_z")}

    o1, n1 = draw(5, 6, 0)
    o2, n2 = draw(5, 6, 0)
    assert torch.equal(o1, o2) and all(torch.equal(n1[k], n2[k]) for k in n1) and torch.isfinite(o1[0])
    assert fs.rng_step == 1  # advanced by the call
    o3, n3 = draw(5, 6, 1)
    assert not torch.equal(n1["eps_z"], n3["eps_z"]) and not torch.equal(n1["u_mask"], n3["u_mask"]) and float(o3[0]) != float(o1[0])
    _, n4 = draw(5, 77, 0)  # another rank: same plated draws, its own z_basal noise
    assert all(torch.equal(n1[k], n4[k]) for k in ("u_mask", "eps_E", "eps_qc_E")) and not torch.equal(n1["eps_z"], n4["eps_z"])
    _, n5 = draw(9, 6, 0)
    assert not torch.equal(n1["eps_E"], n5["eps_E"]) and torch.equal(n1["eps_z"], n5["eps_z"])
    # moments at a size where they mean something
    shape = syn.NORMAN
    fs2 = None
    import bench
    lm2, batch = bench.build(shape, DEV, seed=3)
    fs2 = lm2._fused_step()
    fs2.seed_plated, fs2.seed_local = 11, 12
    fs2.run(X=batch["X"], D=batch["D"], D_obs_counts=batch["D_obs_counts"], qc=batch["qc"],
            qc_obs_counts=batch["qc_obs_counts"], cfX=batch["cfX"], library_size=batch["library_size"],
            n_particles=shape.n_particles, compute_grads=False)
    u = fs2.intermediate("u_mask").double()
    assert 0.0 < float(u.min()) and float(u.max()) < 1.0
    n = u.numel()
    assert abs(float(u.mean()) - 0.5) < 5 * (1 / 12 / n) ** 0.5 and abs(float(u.var()) - 1 / 12) < 5e-3
    for k in ("eps_E", "eps_z"):
        e = fs2.intermediate(k).double()
        m = e.numel()
        assert abs(float(e.mean())) < 5 / m ** 0.5, (k, float(e.mean()))
        assert abs(float(e.var()) - 1.0) < 5 * (2 / m) ** 0.5, (k, float(e.var()))
        assert abs(float((e ** 4).mean()) - 3.0) < 0.1 and float(e.abs().max()) < 7.0  # kurtosis, tails
        lag = float((e[:-1] * e[1:]).mean())
        assert abs(lag) < 5 / m ** 0.5, (k, lag)  # neighbours (the two halves of a Box-Muller pair) uncorrelated


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
