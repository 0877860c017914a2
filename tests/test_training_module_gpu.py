"""The config-built training module on the GPU: training_step + backward through the fused
step's autograd node, the Lightning-free fit loop over the device batch source, checkpoint
round trip."""
import dataclasses
import os

import numpy as np
import pytest
import torch

from tests import golden_io as gio
from tests.test_host_logic_cpu import _module

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_training_step_backward_matches_fused_gradients():
    case = gio.load_elbo("mid", torch.float32, DEV)
    mod = _module(gio.load_elbo("mid"))
    mod.loss_module.load_state_dict(gio.load_elbo("mid")["params"], strict=True)
    mod = mod.to(DEV)
    b = {k: case["batch"][k] for k in ("X", "D", "qc", "cfX", "library_size")}
    torch.manual_seed(0)
    loss = mod.training_step(b, 0)
    assert loss.requires_grad and torch.isfinite(loss)
    loss.backward()
    arena = mod.loss_module._fused_step().arena
    for p, g in zip(arena.params(), arena.grad_views()):
        assert p.grad is not None and torch.equal(p.grad, g)
    assert "train/loss" in mod.logged and "train/loss_term_reconstruction" in mod.logged
    # same draws -> same value in validation mode (no autograd node, no gradient kernels)
    torch.manual_seed(0)
    v = mod.validation_step(b, 0)
    assert not v.requires_grad and torch.allclose(v, loss.detach(), rtol=1e-6)


def test_fit_reduces_loss_and_checkpoint_roundtrip(tmp_path):
    from cradle_vae_amd import synthetic as syn
    from cradle_vae_amd.data import DeviceBatchLoader, SCRNASeqTensorPerturbationDataset
    from cradle_vae_amd.lightning_module import save_checkpoint
    from cradle_vae_amd.trainer import fit

    shape = dataclasses.replace(syn.MID, n_cells=512)
    data = syn.make_batch(shape, seed=2)
    ds = SCRNASeqTensorPerturbationDataset(data["X"], data["D"], data["qc"], data["cfX"])
    case = gio.load_elbo("mid")
    case["batch"] = dict(case["batch"])
    case["batch"]["D_obs_counts"] = ds.get_dosage_obs_per_dim().float()
    case["batch"]["qc_obs_counts"] = ds.get_qc_obs_per_dim().float()
    case["params"]["guide.normalization_module.log_mean"] = data["log_x_mean"]
    case["params"]["guide.normalization_module.log_scale"] = data["log_x_std"]
    torch.manual_seed(1)
    mod = _module(case).to(DEV)
    mod.lr = 3e-3
    train = DeviceBatchLoader(ds, batch_size=64, shuffle=True, drop_last=True, seed=0, device=DEV)
    val = DeviceBatchLoader(ds, batch_size=128, device=DEV)
    hist = fit(mod, train, val, max_epochs=6, gradient_clip_val=100.0, seed=0)
    assert len(hist) == 6 and hist[-1]["steps"] == 6 * 8
    assert all(torch.isfinite(torch.tensor(h["val/loss"])) for h in hist)
    assert hist[-1]["val/loss"] < hist[0]["val/loss"] and hist[-1]["train/loss"] < hist[0]["train/loss"], hist
    path = os.path.join(tmp_path, "m.ckpt")
    save_checkpoint(mod, path)
    again = type(mod).load_from_checkpoint(path, map_location=DEV)
    b = next(iter(val))
    torch.manual_seed(5)
    a = mod.test_step(b, 0)
    torch.manual_seed(5)
    c = again.test_step(b, 0)
    assert torch.equal(a, c)
    # the reloaded predictor shares the reloaded weights
    assert again.predictor.model is again.loss_module.model


def test_end_to_end_synthetic_train_checkpoint_evaluate(tmp_path):
    """train_*.py + eval.py flow on the synthetic data module: fit, checkpoint, reload, evaluate."""
    from cradle_vae_amd.data import SyntheticPerturbationDataModule
    from cradle_vae_amd.evaluate import evaluate_checkpoint
    from cradle_vae_amd.lightning_module import TrainConfigPerturbationLightningModule, save_checkpoint
    from cradle_vae_amd.trainer import fit

    G, T = 64, 6
    dm_kwargs = dict(n_cells=3000, n_genes=G, n_guides=T, batch_size=128, seed=1)
    dm = SyntheticPerturbationDataModule(device=DEV, **dm_kwargs)
    common = dict(n_latent=8, n_treatments=T, n_phenos=G, n_qc=1)
    config = dict(
        model="cradle_vae_Model", guide="cradle_vae_CorrelatedNormalGuide", loss_module="cradle_vae_ELBOLossModule",
        predictor="cradle_vae_Predictor", data_module="SyntheticPerturbationDataModule",
        data_module_kwargs=dict(dm_kwargs, **dm.get_qc_n()),
        lightning_module_kwargs=dict(lr=2e-3, n_particles=3), loss_module_kwargs=dict(beta=0.5),
        model_kwargs=dict(mask_prior_prob=0.05, embedding_prior_scale=1.0, likelihood_key="library_nb",
                          decoder_n_layers=1, decoder_n_hidden=32, sum_mode=False, **common),
        guide_kwargs=dict(basal_encoder_n_layers=1, basal_encoder_n_hidden=32,
                          basal_encoder_input_normalization="log_standardize", embedding_encoder_n_layers=2,
                          embedding_encoder_n_hidden=32, x_normalization_stats=dm.get_x_train_statistics(), **common))
    torch.manual_seed(0)
    mod = TrainConfigPerturbationLightningModule(
        config=config, D_obs_counts_train=dm.get_train_perturbation_obs_counts().float(),
        D_obs_counts_val=dm.get_val_perturbation_obs_counts().float(),
        D_obs_counts_test=dm.get_test_perturbation_obs_counts().float(),
        qc_obs_counts_train=dm.get_train_qc_obs_counts().float(), qc_obs_counts_val=dm.get_val_qc_obs_counts().float(),
        qc_obs_counts_test=dm.get_test_qc_obs_counts().float()).to(DEV)
    hist = fit(mod, dm.train_dataloader(), dm.val_dataloader(), max_epochs=12)
    assert hist[-1]["val/loss"] < hist[0]["val/loss"]
    path = os.path.join(tmp_path, "e2e.ckpt")
    save_checkpoint(mod, path)
    metrics = evaluate_checkpoint(path, "perturbseq", batch_size=128, ate_n_particles=256, data_module=dm)
    for k in ("test/IWELBO", "ATE_pearsonr-all", "ATE_r2-all", "ATE_pearsonr_top20-all", "jaccard_sim_top20-all",
              "jaccard_sim_top50-all", "ATE_pearsonr-test"):
        assert k in metrics and metrics[k] == metrics[k], k  # present and not NaN
    assert metrics["ATE_n_particles"] == 256 and -1.0 <= metrics["ATE_pearsonr-all"] <= 1.0
    # the data module rebuilt from the checkpoint's config is the same data
    dm2 = type(mod).load_from_checkpoint(path).get_data_module()
    assert torch.equal(dm2.datasets["test"].X.cpu(), dm.datasets["test"].X.cpu())


def test_train_and_eval_entry_points_from_a_yaml(tmp_path):
    """train.py --config <yaml in the reference's format> -> results dir with metrics.csv, checkpoints, summary.csv
    (train_norman.py:48-129), then eval.py on that directory (eval.py:206-243): IWELBO + ATE metrics."""
    import csv

    import yaml

    import eval as eval_entry
    import train as train_entry

    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "demo",
                                           "cradle_vae_synthetic.yaml")))
    cfg.update({"max_epochs": 3, "data_module_kwargs.n_cells": 1500, "data_module_kwargs.n_genes": 64,
                "data_module_kwargs.n_guides": 5, "data_module_kwargs.batch_size": 128, "n_latent": 8,
                "model_kwargs.decoder_n_hidden": 32, "guide_kwargs.basal_encoder_n_hidden": 32,
                "guide_kwargs.embedding_encoder_n_hidden": 32, "use_wandb": True, "devices": [7]})
    path = tmp_path / "cfg.yaml"
    path.write_text(yaml.safe_dump(cfg))
    train_entry.main(["--config", str(path), "--results-dir", str(tmp_path / "results"), "--some-sweep-arg", "1"])
    rdir = tmp_path / "results" / "cradle_vae_synthetic"
    rows = list(csv.DictReader(open(rdir / "metrics.csv")))
    assert len(rows) == 3 and float(rows[-1]["train/loss"]) < float(rows[0]["train/loss"])
    summary = next(csv.DictReader(open(rdir / "summary.csv")))
    assert os.path.exists(summary["best_checkpoint_path"]) and np.isfinite(float(summary["val/IWELBO"]))
    train_entry.main(["--config", str(path), "--results-dir", str(tmp_path / "results"), "--max_epochs", "1"])
    assert (tmp_path / "results" / "cradle_vae_synthetic-2" / "summary.csv").exists()  # repeated runs enumerate
    m = eval_entry.main(["--experiment_path", str(rdir), "--batch_size", "64", "--ate_n_particles", "64",
                         "--qc_pass", "false"])
    assert np.isfinite(m["test/IWELBO"]) and "pearsonr-all" in m or any(k.endswith("-all") for k in m)
    assert (rdir / "test_metrics.csv").exists()


def test_module_loaded_from_a_reference_layout_checkpoint_gives_the_references_loss():
    """The state dict the reference's own Lightning module accepted, loaded here, gives the ELBO and the loss terms
    the reference computed from it on the same batch and noise (fixture: oracle/make_checkpoint_golden.py)."""
    from cradle_vae_amd.models.utils.perturbation_lightning_module import TrainConfigPerturbationLightningModule
    from tests.test_host_logic_cpu import reference_checkpoint_case

    case = reference_checkpoint_case(device="cuda")
    mod = TrainConfigPerturbationLightningModule(config=case["config"])
    mod.load_state_dict(case["state_dict"], strict=True)
    mod = mod.to("cuda")
    b = case["batch"]
    for impl in ("fused", "modular"):
        mod.loss_module.step = impl
        loss, metrics = mod.loss_module.loss(
            X=b["X"], D=b["D"], D_obs_counts=mod.D_obs_counts_train, qc=b["qc"], qc_obs_counts=mod.qc_obs_counts_train,
            cfX=b["cfX"], condition_values={"library_size": b["library_size"]}, n_particles=case["n_particles"],
            noise=case["noise"], cf_noise=case["cf_noise"])
        assert abs(float(loss) - case["loss"]) <= 1e-4 * abs(case["loss"]), (impl, float(loss), case["loss"])
        for k, v in case["metrics"].items():
            assert abs(float(metrics[k]) - v) <= 1e-4 * max(abs(v), 1e-3), (impl, k, float(metrics[k]), v)
