"""The config-built training module on the GPU: training_step + backward through the fused
step's autograd node, the Lightning-free fit loop over the device batch source, checkpoint
round trip."""
import dataclasses
import os

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
