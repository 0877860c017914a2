"""GPU parity of the module-level API (guide / model / loss_module, modular path) against the
golden fixtures the reference produced (fp64 run) -- loss, every metric, every gradient."""
import pytest
import torch

from tests import golden_io as gio

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build_loss_module(case, device=DEV):
    from cradle_vae_amd.guide import ObservationNormalizationStatistics, cradle_vae_CorrelatedNormalGuide
    from cradle_vae_amd.loss_module import cradle_vae_ELBOLossModule
    from cradle_vae_amd.model import cradle_vae_Model

    cfg, p = case["cfg"], case["params"]
    model = cradle_vae_Model(
        n_latent=cfg["n_latent"], n_treatments=cfg["n_treatments"], n_phenos=cfg["n_phenos"], n_qc=cfg["n_qc"],
        mask_prior_prob=cfg["mask_prior_prob"], embedding_prior_scale=cfg["embedding_prior_scale"],
        likelihood_key="library_nb", decoder_n_layers=cfg["decoder_n_layers"],
        decoder_n_hidden=cfg["decoder_n_hidden"], sum_mode=False)
    stats = ObservationNormalizationStatistics(
        log_x_mean=p["guide.normalization_module.log_mean"].clone(),
        log_x_std=p["guide.normalization_module.log_scale"].clone())
    guide = cradle_vae_CorrelatedNormalGuide(
        n_latent=cfg["n_latent"], n_treatments=cfg["n_treatments"], n_phenos=cfg["n_phenos"], n_qc=cfg["n_qc"],
        basal_encoder_n_layers=cfg["basal_encoder_n_layers"], basal_encoder_n_hidden=cfg["basal_encoder_n_hidden"],
        basal_encoder_input_normalization="log_standardize",
        embedding_encoder_n_layers=cfg["embedding_encoder_n_layers"],
        embedding_encoder_n_hidden=cfg["embedding_encoder_n_hidden"], x_normalization_stats=stats)
    lm = cradle_vae_ELBOLossModule(model=model, guide=guide, beta=cfg["beta"], alpha=cfg["alpha"])
    missing, unexpected = lm.load_state_dict(p, strict=True)  # same keys as the reference's state dict
    assert not missing and not unexpected
    return lm.to(device)


GRAD_L2_RTOL = 1e-4  # ||g_hip - g_ref||_2 <= 1e-4 ||g_ref||_2 per parameter tensor (fp32 kernels vs the fp64 reference)


def check_against_golden(case, loss, metrics, grads, rtol=1e-4):
    ref = case["f64"]
    assert abs(float(loss) - ref["loss"]) <= rtol * max(1.0, abs(ref["loss"])), (float(loss), ref["loss"])
    for k, v in ref["metrics"].items():
        assert abs(float(metrics[k]) - v) <= rtol * max(1.0, abs(v)), (k, float(metrics[k]), v)
    for k, g in ref["grads"].items():
        got = grads[k].detach().cpu().double()
        scale = max(float(g.abs().max()), 1e-8)
        err = float((got - g).abs().max()) / scale
        assert err <= 2e-3, (k, err)
        # and in the L2 sense, which small entries of a tensor cannot hide behind its largest one
        l2 = float((got - g).norm()) / max(float(g.norm()), 1e-12)
        assert l2 <= GRAD_L2_RTOL, (k, l2)


@pytest.mark.parametrize("name", gio.elbo_cases())
def test_modular_loss_matches_reference(name):
    case = gio.load_elbo(name, torch.float32, DEV)
    lm = build_loss_module(case)
    lm.step = "modular"
    b, cfg = case["batch"], case["cfg"]
    loss, metrics = lm.loss(
        X=b["X"], D=b["D"], D_obs_counts=b["D_obs_counts"], qc=b["qc"], qc_obs_counts=b["qc_obs_counts"],
        cfX=b["cfX"], condition_values={"library_size": b["library_size"]}, n_particles=cfg["n_particles"],
        noise=case["noise"], cf_noise=case["cf_noise"])
    loss.backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in lm.named_parameters()}
    check_against_golden(case, loss, metrics, grads)


@pytest.mark.parametrize("name", gio.elbo_cases())
def test_guide_samples_match_reference(name):
    """guide.forward on the GPU against the reference guide's own draws (same injected noise): the
    mask bit-exact, the reparameterised Normal samples within fp32 GEMM tolerance."""
    case = gio.load_elbo(name, torch.float32, DEV)
    lm = build_loss_module(case)
    b, cfg = case["batch"], case["cfg"]
    with torch.no_grad():
        _, s = lm.guide(X=b["X"], D=b["D"], qc=b["qc"], n_particles=cfg["n_particles"], noise=case["noise"])
    ref = case["f64"]["samples"]
    assert torch.equal(s["mask"].cpu().double(), ref["mask"].double())
    for k in ("E", "qc_E", "z_basal"):
        r = ref[k].double()
        err = float((s[k].cpu().double() - r).abs().max()) / max(float(r.abs().max()), 1e-6)
        assert err <= 2e-5, (k, err)


def test_dense_dosage_product_runs_on_the_hip_gemm():
    """matmul(D, table) for a dosage matrix denser than the gather plans for (the fallback of guide / model / loss module /
    ATE when a cell carries more than max_dosages_per_cell dosages): one HIP GEMM per particle forward, D^T dout backward
    -- against torch in fp64, real-valued dosages, a width that is not a multiple of 4, shared and per-particle tables."""
    from cradle_vae_amd import functional as F_hip

    g = torch.Generator().manual_seed(4)
    N, T, L, P = 37, 53, 18, 3
    D = (torch.rand(N, T, generator=g) < 0.4).float() * torch.rand(N, T, generator=g)
    for shape in ((P, T, L), (1, T, L), (T, L)):
        table = torch.randn(*shape, generator=g)
        t_dev = table.to(DEV).requires_grad_(True)
        out = F_hip.dense_dosage_matmul(D.to(DEV), t_dev, P)
        w = torch.randn(P, N, L, generator=g)
        (out * w.to(DEV)).sum().backward()
        t64 = table.double().requires_grad_(True)
        ref = torch.matmul(D.double(), t64 if t64.dim() == 3 else t64.unsqueeze(0)).expand(P, N, L)
        (ref * w.double()).sum().backward()
        assert out.shape == (P, N, L)
        assert float((out.detach().cpu().double() - ref.detach()).abs().max()) <= 1e-5 * float(ref.abs().max())
        assert t_dev.grad.shape == table.shape
        assert float((t_dev.grad.cpu().double() - t64.grad).abs().max()) <= 1e-5 * float(t64.grad.abs().max())
