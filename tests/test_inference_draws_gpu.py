"""cradle_mask_sample / cradle_gauss_sample (the one-launch draws of guide.forward under no_grad) against the module
path's torch form, which tests/test_modular_gpu.py pins to the reference's own draws: the mask bit for bit, the Gaussian
draw to one fused multiply-add's rounding; non-finite inputs take the reference's maps (sanitize_tensor, nan_to_num)."""
import pytest
import torch

from tests import golden_io as gio
from tests.test_modular_gpu import build_loss_module

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("P,T,L", [(5, 12, 16), (3, 7, 10), (128, 105, 200)])
def test_mask_sample_equals_the_distribution_rsample(P, T, L):
    from cradle_vae_amd import ops
    from cradle_vae_amd.distributions import GumbelSoftmaxBernoulliStraightThrough

    g = torch.Generator().manual_seed(P * 1000 + T)
    logits = (3.0 * torch.randn(T, L, generator=g)).to(DEV)
    logits[0, 0], logits[1, 1], logits[2, 2] = float("nan"), float("inf"), -float("inf")
    u = torch.rand(P, T, L, generator=g).to(DEV)
    u[0, 0, 1], u[0, 0, 2] = 0.0, 1.0  # clamped to [eps, 1 - eps] like torch's clamp_probs
    for tau in (1.0, 0.5):
        got = ops.mask_sample(logits, u, tau)
        clean = torch.where(torch.isfinite(logits), logits, torch.zeros_like(logits))  # guide.sanitize_tensor
        ref = GumbelSoftmaxBernoulliStraightThrough(torch.tensor([tau], device=DEV), clean).rsample((P,), uniforms=u)
        assert got.shape == ref.shape and torch.equal(got, ref)
        assert set(got.unique().tolist()) <= {0.0, 1.0}


@pytest.mark.parametrize("rows,L", [(40, 16), (33, 10), (13440, 200)])
def test_gauss_sample_equals_the_head_s_torch_form(rows, L):
    from cradle_vae_amd import ops
    from cradle_vae_amd.mlp import nan_to_num_ref

    g = torch.Generator().manual_seed(rows + L)
    heads = (2.0 * torch.randn(rows, 2 * L + 4, generator=g)).to(DEV)  # strided views: both halves of one buffer
    mean_raw, lv_raw = heads[:, :L], heads[:, L + 4:2 * L + 4] if L % 4 == 0 else heads[:, L:2 * L]
    heads[0, 0], heads[1, 1] = float("nan"), float("inf")          # mean: nan -> 1e-6, +inf -> 1e6
    lv_raw[2, 0], lv_raw[3, 1], lv_raw[4, 2] = float("nan"), 200.0, -200.0  # var: nan -> 1e-6, exp overflow -> 1e6, eps floor
    eps = torch.randn(rows, L, generator=g).to(DEV)
    got = ops.gauss_sample(mean_raw, lv_raw, eps)
    mean = nan_to_num_ref(mean_raw)
    var = nan_to_num_ref(torch.abs(torch.exp(lv_raw) + 1e-4))
    ref = mean.double() + var.double().sqrt() * eps.double()
    assert torch.isfinite(got).all()
    assert torch.allclose(got.double(), ref, rtol=2e-6, atol=1e-6)


def test_guide_under_no_grad_draws_the_same_with_and_without_the_fused_launches(monkeypatch):
    case = gio.load_elbo("mid", torch.float32, DEV)
    lm = build_loss_module(case)
    cfg = case["cfg"]
    P = cfg["n_particles"]
    noise = case["noise"]
    outs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("CRADLE_FUSED_DRAWS", fused)
        with torch.no_grad():
            _, s = lm.guide(n_particles=P, noise=noise)
        outs.append(s)
    assert torch.equal(outs[0]["mask"], outs[1]["mask"])
    assert torch.allclose(outs[0]["E"], outs[1]["E"], rtol=1e-6, atol=1e-6)
    assert torch.equal(outs[0]["qc_E"], outs[1]["qc_E"])
    # the reference guide's own draw on the same noise (the fixture holds it)
    assert torch.equal(outs[0]["mask"].cpu().double(), case["f64"]["samples"]["mask"].double())
