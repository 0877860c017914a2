"""Generates tests/golden/*.npz by executing the reference (build container only).

    python -m oracle.make_golden

TEST INFRASTRUCTURE ONLY.  Each fixture holds plain arrays: the inputs (fp32), the
injected noise, the full state dict, and the reference's outputs from an fp32 run and
from an fp64 run of the same fp32-representable inputs.  Nothing of the reference's
source travels; see oracle/ref_harness.py for how it is imported and which two pyro
classes are restated (parity unpinned for those).
"""
from __future__ import annotations

import dataclasses
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cradle_vae_amd import synthetic as syn  # noqa: E402
from oracle import cradle_oracle as co  # noqa: E402
from oracle import ref_harness as rh  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def cfg_of(shape: syn.Shape, **over):
    cfg = dict(
        n_latent=shape.n_latent,
        n_treatments=shape.n_treatments,
        n_phenos=shape.n_phenos,
        n_qc=shape.n_qc,
        decoder_n_layers=shape.decoder_n_layers,
        decoder_n_hidden=shape.decoder_n_hidden,
        basal_encoder_n_layers=shape.basal_encoder_n_layers,
        basal_encoder_n_hidden=shape.basal_encoder_n_hidden,
        embedding_encoder_n_layers=shape.embedding_encoder_n_layers,
        embedding_encoder_n_hidden=shape.embedding_encoder_n_hidden,
        mask_prior_prob=shape.mask_prior_prob,
        embedding_prior_scale=1.0,
        beta=shape.beta,
        alpha=None,
        n_particles=shape.n_particles,
        n_cells=shape.n_cells,
    )
    cfg.update(over)
    return cfg


def init_kwargs(cfg):
    keys = (
        "n_latent n_treatments n_phenos n_qc decoder_n_layers decoder_n_hidden basal_encoder_n_layers "
        "basal_encoder_n_hidden embedding_encoder_n_layers embedding_encoder_n_hidden mask_prior_prob "
        "embedding_prior_scale"
    ).split()
    return {k: cfg[k] for k in keys}


def elbo_case(name, shape, seed, mutate=None, **cfg_over):
    ref = rh.load_reference()
    cfg = cfg_of(shape, **cfg_over)
    batch = syn.make_batch(shape, seed=seed, dtype=torch.float32)
    if mutate is not None:
        mutate(batch)
        batch["library_size"] = batch["X"].sum(1)
    params = co.init_params(
        **init_kwargs(cfg), log_x_mean=batch["log_x_mean"], log_x_std=batch["log_x_std"], seed=seed + 100,
        dtype=torch.float32,
    )
    noise = syn.make_noise(shape, seed + 200, dtype=torch.float32)
    cf_noise = syn.make_noise(shape, seed + 300, dtype=torch.float32)
    out = {"cfg": np.array(json.dumps(cfg))}
    for k, v in batch.items():
        out[f"batch/{k}"] = v.numpy()
    for k, v in params.items():
        out[f"params/{k}"] = v.numpy()
    for k, v in noise.items():
        out[f"noise/{k}"] = v.numpy()
    for k, v in cf_noise.items():
        out[f"cf_noise/{k}"] = v.numpy()
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        cast = lambda d: {k: (v if v.dtype == torch.int64 else v.to(dt)) for k, v in d.items()}  # noqa: E731
        lm = rh.build_reference_loss_module(ref, cast(params), cfg, dt)
        loss, metrics, grads = rh.reference_loss_and_grads(lm, cast(batch), cast(noise), cast(cf_noise), cfg["n_particles"])
        out[f"{tag}/loss"] = loss.numpy()
        for k, v in metrics.items():
            out[f"{tag}/metrics/{k}"] = v.numpy()
        for k, v in grads.items():
            out[f"{tag}/grads/{k}"] = v.numpy()
        ref_samples = rh.reference_guide_samples(lm, cast(batch), cast(noise), cfg["n_particles"])
        for k, v in ref_samples.items():
            out[f"{tag}/samples/{k}"] = v.numpy()
        _, o_samples = co.guide_forward(cast(params), cast(noise), cfg["n_particles"], X=cast(batch)["X"],
                                        D=cast(batch)["D"], qc=cast(batch)["qc"])
        assert torch.equal(o_samples["mask"].detach(), ref_samples["mask"]), (name, tag, "mask")
        # the oracle must reproduce the reference before a fixture is written
        ol, om, _, og = co.elbo_loss_and_grads(
            cast(params), X=cast(batch)["X"], D=cast(batch)["D"], qc=cast(batch)["qc"], cfX=cast(batch)["cfX"],
            library_size=cast(batch)["library_size"], D_obs_counts=cast(batch)["D_obs_counts"],
            qc_obs_counts=cast(batch)["qc_obs_counts"], noise=cast(noise), cf_noise=cast(cf_noise),
            n_particles=cfg["n_particles"], beta=cfg["beta"], alpha=cfg["alpha"],
        )
        tol = 1e-12 if dt == torch.float64 else 2e-5
        assert abs(float(ol) - float(loss)) <= tol * abs(float(loss)), (name, tag, float(ol), float(loss))
    path = os.path.join(GOLDEN, f"elbo_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: loss f32 {float(out['f32/loss']):.6f} f64 {float(out['f64/loss']):.6f} ({os.path.getsize(path)} B)")


def ate_iwelbo_case(name, shape, seed, n_alt=4, ate_particles=12, ate_chunk=8, iw_particles=7, iw_batch=5):
    """Reference ATE (count sampler replaced by its mean: the sampler itself is pyro's) and
    predictive IWELBO on injected noise."""
    import types

    from torch.utils.data import DataLoader, Dataset

    ref = rh.load_reference()
    cfg = cfg_of(shape)
    batch = syn.make_batch(shape, seed=seed, dtype=torch.float32)
    params = co.init_params(
        **init_kwargs(cfg), log_x_mean=batch["log_x_mean"], log_x_std=batch["log_x_std"], seed=seed + 100,
        dtype=torch.float32,
    )
    lm = rh.build_reference_loss_module(ref, params, cfg, torch.float32)
    predictor = ref.models.cradle_vae_Predictor(lm.model, lm.guide)
    T, L, Q, G = cfg["n_treatments"], cfg["n_latent"], cfg["n_qc"], cfg["n_phenos"]
    g = torch.Generator().manual_seed(seed + 400)
    out = {"cfg": np.array(json.dumps(cfg))}
    for k, v in batch.items():
        out[f"batch/{k}"] = v.numpy()
    for k, v in params.items():
        out[f"params/{k}"] = v.numpy()

    # ---------------- ATE (analysis/average_treatment_effects.py:78-203) ----------------
    D_alt = torch.zeros(n_alt, T)
    for i in range(n_alt):
        D_alt[i, i % T] = 1.0
    D_alt[n_alt - 1, (n_alt) % T] = 1.0  # last alternative is a 2-guide combination
    D_ctl = torch.zeros(1, T)
    quality = torch.ones(1, Q)
    lib = 1e4 * torch.ones((1,))
    queue = [("normal", None), ("normal", None), ("normal", None)]  # the n_phenos probe call (:117-119)
    chunks = []
    for i in range(0, ate_particles, ate_chunk):
        c = min(ate_chunk, ate_particles - i)
        nz = dict(
            u_mask=torch.rand(c, T, L, generator=g),
            eps_E=torch.randn(c, T, L, generator=g),
            eps_qc_E=torch.randn(c, Q, L, generator=g),
            z_control=torch.randn(c, 1, L, generator=g),
            z_alt=torch.randn(n_alt, c, 1, L, generator=g),
        )
        chunks.append(nz)
        queue += rh.guide_noise_queue(nz, with_z=False) + [("normal", nz["z_control"])]
        queue += [("normal", nz["z_alt"][t]) for t in range(n_alt)]
    for method in ("mean", "perturbseq"):
        with rh.injected_noise(queue) as rest:
            ate = ref.estimate_model_average_treatment_effect(
                model=lm.model, guide=lm.guide, dosages_alt=D_alt, dosages_control=D_ctl, quality=quality,
                n_particles=ate_particles, method=method, condition_values=dict(library_size=lib.clone()),
                batch_size=ate_chunk,
            )
            assert not rest
        out[f"ate/{method}"] = ate
    out["ate/D_alt"], out["ate/D_control"] = D_alt.numpy(), D_ctl.numpy()
    out["ate/n_particles"], out["ate/chunk"] = np.array(ate_particles), np.array(ate_chunk)
    for ci, nz in enumerate(chunks):
        for k, v in nz.items():
            out[f"ate/noise{ci}/{k}"] = v.numpy()

    # ---------------- predictive IWELBO (models/utils/predictor.py:77-200) ----------------
    class DS(Dataset):
        def __len__(self):
            return shape.n_cells

        def __getitem__(self, i):
            return dict(idx=i, X=batch["X"][i], D=batch["D"][i], qc=batch["qc"][i],
                        library_size=batch["library_size"][i], cfX=batch["cfX"][i])

        def convert_idx_to_ids(self, idx):
            return np.asarray(idx)

    P = iw_particles
    plated = dict(
        u_mask=torch.rand(P, T, L, generator=g), eps_E=torch.randn(P, T, L, generator=g),
        eps_qc_E=torch.randn(P, Q, L, generator=g),
    )
    queue = rh.guide_noise_queue(plated, with_z=False)
    eps_z = torch.randn(P, shape.n_cells, L, generator=g)
    for s in range(0, shape.n_cells, iw_batch):
        queue.append(("normal", eps_z[:, s : s + iw_batch].contiguous()))
    with rh.injected_noise(queue) as rest:
        df = predictor.compute_predictive_iwelbo(DataLoader(DS(), batch_size=iw_batch), n_particles=P)
        assert not rest
    out["iwelbo/values"] = df["IWELBO"].to_numpy()
    out["iwelbo/ids"] = df.index.to_numpy()
    out["iwelbo/n_particles"], out["iwelbo/batch"] = np.array(P), np.array(iw_batch)
    for k, v in plated.items():
        out[f"iwelbo/noise/{k}"] = v.numpy()
    out["iwelbo/noise/eps_z"] = eps_z.numpy()

    # ---------------- sample_observations (models/utils/predictor.py:202-273) ----------------
    # count sampler in "mean" mode (pyro's sampler is not in the reference); per-dosage prior draws of z_basal
    # injected.  (noise drawn after everything above: the earlier sections of the fixture keep their values)
    Ps = 3
    D_obs = torch.cat([D_ctl, D_alt])
    so = dict(u_mask=torch.rand(Ps, T, L, generator=g), eps_E=torch.randn(Ps, T, L, generator=g),
              eps_qc_E=torch.randn(Ps, Q, L, generator=g), z=torch.randn(D_obs.shape[0], Ps, 1, L, generator=g))
    queue = rh.guide_noise_queue(so, with_z=False) + [("normal", so["z"][i]) for i in range(D_obs.shape[0])]
    names = [f"pert{i}" for i in range(D_obs.shape[0])]
    with rh.injected_noise(queue) as rest:
        good, bad = predictor.sample_observations(
            dosages=D_obs, perturbation_names=names, n_particles=Ps,
            condition_values=dict(library_size=lib.clone()), qc=torch.ones(1, Q))
        assert not rest
    out["samples/good"], out["samples/bad"] = np.asarray(good.X), np.asarray(bad.X)
    out["samples/perturbation_idx"] = good.obs["perturbation_idx"].to_numpy()
    out["samples/particle_idx"] = good.obs["particle_idx"].to_numpy()
    out["samples/n_particles"] = np.array(Ps)
    out["samples/dosages"] = D_obs.numpy()
    for k, v in so.items():
        out[f"samples/noise/{k}"] = v.numpy()
    path = os.path.join(GOLDEN, f"predictor_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: ({os.path.getsize(path)} B)")


def metrics_case(seed=21, n_cells=240, n_genes=64, n_pert=6):
    """Reference analysis helpers on synthetic arrays: data-side ATE (both methods), the eval.py metric
    stage (get_ate_metrics / jaccard_sim), align_adatas, and the train-set log statistics."""
    import importlib.util

    import pandas as pd

    ref = rh.load_reference()
    import anndata  # the harness stub
    from cradle_vae.analysis.average_treatment_effects import estimate_data_average_treatment_effects
    from cradle_vae.data.utils.anndata import align_adatas
    from cradle_vae.data.utils.batch_statistics import batch_log_mean, batch_log_std

    cwd = os.getcwd()
    spec = importlib.util.spec_from_file_location("_ref_eval", os.path.join(rh.REFERENCE_ROOT, "eval.py"))
    ref_eval = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_eval)  # chdirs into the reference
    os.chdir(cwd)

    rng = np.random.default_rng(seed)
    lam = np.exp(rng.normal(0.5, 1.0, size=n_genes))
    labels = np.array(["ctrl"] + [f"g{i}" for i in range(n_pert)])[rng.integers(0, n_pert + 1, size=n_cells)]
    shift = np.exp(rng.normal(0, 0.4, size=(n_pert + 1, n_genes)))
    shift[0] = 1.0
    lab_idx = np.array([0 if l == "ctrl" else int(l[1:]) + 1 for l in labels])
    X = rng.poisson(lam * shift[lab_idx]).astype(np.float32)
    X[:, 0] += 1  # no empty cell
    adata = anndata.AnnData(X=X, obs=pd.DataFrame({"pert": labels}, index=[f"c{i}" for i in range(n_cells)]),
                            var=pd.DataFrame(index=[f"gene{i}" for i in range(n_genes)]))
    out = {"X": X, "labels": labels.astype("U8")}
    ates = {}
    for method in ("mean", "perturbseq"):
        res = estimate_data_average_treatment_effects(adata, "pert", "ctrl", method)
        ates[method] = res
        out[f"data_ate/{method}"] = np.asarray(res.X)
        out[f"data_ate/{method}_index"] = np.array(list(res.obs.index)).astype("U8")
    data_ate = ates["perturbseq"]
    model_X = np.asarray(data_ate.X) + rng.normal(0, 0.15, size=data_ate.X.shape)
    order = rng.permutation(model_X.shape[0])[:-1]  # the model side misses one perturbation, other order
    model_ate = anndata.AnnData(X=model_X[order].astype(np.float32),
                                obs=pd.DataFrame(index=np.array(list(data_ate.obs.index))[order]))
    a, b = align_adatas(data_ate, model_ate)
    out["model_ate/X"], out["model_ate/index"] = np.asarray(model_ate.X), np.array(list(model_ate.obs.index)).astype("U8")
    out["aligned/index"] = np.array(list(a.obs.index)).astype("U8")
    out["aligned/data_X"], out["aligned/model_X"] = np.asarray(a.X), np.asarray(b.X)
    for k, v in ref_eval.get_ate_metrics(a, b).items():
        out[f"metrics/{k}"] = np.float64(v)
    Xt = torch.from_numpy(X)
    lm = batch_log_mean(Xt, step_size=100)
    out["stats/log_mean"], out["stats/log_std"] = lm.numpy(), batch_log_std(Xt, lm, step_size=100).numpy()
    path = os.path.join(GOLDEN, "metrics_tiny.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: ({os.path.getsize(path)} B)", {k: float(v) for k, v in out.items() if k.startswith("metrics/")})


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    rh._GammaPoisson.SAMPLE_MODE = "mean"
    if len(sys.argv) > 1 and sys.argv[1] == "metrics":  # only the analysis-helper fixture
        metrics_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "predictor":  # only the predictor fixture
        ate_iwelbo_case("tiny", syn.TINY, 9)
        return
    T = syn.TINY
    elbo_case("tiny", T, 1)
    elbo_case("mid", syn.MID, 2)

    def edge(b):  # control cell, 2-guide cell, an unobserved treatment, a dead gene
        b["D"][0] = 0.0
        b["D"][1] = 0.0
        b["D"][1, 0] = 1.0
        b["D"][1, 3] = 1.0
        b["D_obs_counts"][2] = 0.0  # 1/0 -> inf -> 0 (loss_modules.py:226-227)
        b["X"][:, 5] = 0.0  # zero train std: (0-0)/0 = nan -> 1e-6 (normalization.py:44)
        b["cfX"][:, 5] = 0.0
        b["log_x_mean"][5] = 0.0
        b["log_x_std"][5] = 0.0
        b["qc"][0] = 1.0
        b["qc"][1] = 0.0

    elbo_case("tiny_edge", T, 3, mutate=edge)

    def no_cf(b):
        b["cfX"].zero_()  # idx empty -> KL = 0 (loss_modules.py:145-151)

    elbo_case("tiny_nocf", T, 4, mutate=no_cf)

    def all_pass(b):
        b["qc"].zero_()  # alpha = 0

    elbo_case("tiny_allpass", T, 5, mutate=all_pass)

    def all_fail(b):
        b["qc"].fill_(1.0)  # alpha = 1

    elbo_case("tiny_allfail", T, 6, mutate=all_fail)
    elbo_case("tiny_alpha", T, 7, alpha=0.3, beta=2.0)
    elbo_case("tiny_q2", dataclasses.replace(T, n_qc=2), 8)  # use_each_qc-style multi-column qc
    ate_iwelbo_case("tiny", T, 9)
    metrics_case()


if __name__ == "__main__":
    main()
