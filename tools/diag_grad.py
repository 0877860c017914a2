"""Where a gradient tensor of the fused step differs from the fp64 oracle: python tools/diag_grad.py <shape> <kmax> <seed> <param key substring>
Prints the error by row and by column, the largest entries, and the mask bits that differ from the oracle's."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cradle_vae_amd import synthetic as syn  # noqa: E402
from oracle import cradle_oracle as co  # noqa: E402
from tests.test_fused_step_gpu import run_fused  # noqa: E402
from tests.test_modular_gpu import DEV, build_loss_module  # noqa: E402

shape, kmax, seed, key = syn.SHAPES[sys.argv[1]], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
b = syn.make_batch(shape, seed=seed)
cfg = dict(
    n_latent=shape.n_latent, n_treatments=shape.n_treatments, n_phenos=shape.n_phenos, n_qc=shape.n_qc,
    decoder_n_layers=shape.decoder_n_layers, decoder_n_hidden=shape.decoder_n_hidden,
    basal_encoder_n_layers=shape.basal_encoder_n_layers, basal_encoder_n_hidden=shape.basal_encoder_n_hidden,
    embedding_encoder_n_layers=shape.embedding_encoder_n_layers,
    embedding_encoder_n_hidden=shape.embedding_encoder_n_hidden, mask_prior_prob=shape.mask_prior_prob,
    embedding_prior_scale=1.0, beta=shape.beta, alpha=None, n_particles=shape.n_particles)
params = co.init_params(**{k: cfg[k] for k in cfg if k not in ("beta", "alpha", "n_particles")},
                        log_x_mean=b["log_x_mean"], log_x_std=b["log_x_std"], seed=23)
noise, cf_noise = syn.make_noise(shape, 31), syn.make_noise(shape, 32)
f64 = lambda d: {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()}
torch.set_num_threads(16)
ref_loss, ref_metrics, inter, ref_grads = co.elbo_loss_and_grads(
    f64(params), X=b["X"].double(), D=b["D"].double(), qc=b["qc"].double(), cfX=b["cfX"].double(),
    library_size=b["library_size"].double(), D_obs_counts=b["D_obs_counts"].double(),
    qc_obs_counts=b["qc_obs_counts"].double(), noise=f64(noise), cf_noise=f64(cf_noise),
    n_particles=cfg["n_particles"], beta=cfg["beta"], alpha=cfg["alpha"])
dev = lambda d: {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in d.items()}
case = dict(cfg=cfg, params=dev(params), batch=dev(b), noise=dev(noise), cf_noise=dev(cf_noise))
lm = build_loss_module(case)
lm.max_dosages_per_cell = kmax
loss, metrics, grads = run_fused(lm, case)
fs = lm._fused
P, T, L = shape.n_particles, shape.n_treatments, shape.n_latent
if fs.desc.N * fs.desc.kmax >= T:
    mask = fs.intermediate("mask").view(2 * P, T, L).cpu().double()
    d_main, d_cf = (mask[:P] != inter["mask"]), (mask[P:] != inter["cf_mask"])
    print("mask bits differing from the oracle: main", int(d_main.sum()), "cf", int(d_cf.sum()))
    for name, dd in (("main", d_main), ("cf", d_cf)):
        for idx in dd.nonzero()[:10].tolist():
            print("  ", name, "flip at (p, t, l) =", idx)
for k, g in ref_grads.items():
    got = grads[k].detach().cpu().double()
    l2 = float((got - g).norm() / max(float(g.norm()), 1e-30))
    if key in k or l2 > 2e-5:
        print(f"{k}: rel L2 {l2:.3e}  |g| {float(g.norm()):.3e}")
    if key in k and g.dim() == 2:
        e = got - g
        rows = e.norm(dim=1)
        top = torch.topk(rows, min(5, rows.numel()))
        print("  error norm by row (top 5):", [(int(i), f"{float(v):.2e}", f"row |g| {float(g[i].norm()):.2e}") for v, i in zip(top.values, top.indices)])
        cols = e.norm(dim=0)
        topc = torch.topk(cols, min(5, cols.numel()))
        print("  error norm by column (top 5):", [(int(i), f"{float(v):.2e}") for v, i in zip(topc.values, topc.indices)])
        flat = e.abs().flatten()
        tv = torch.topk(flat, 8)
        for v, i in zip(tv.values, tv.indices):
            r, c = divmod(int(i), g.shape[1])
            print(f"  entry ({r}, {c}): got {float(got[r, c]):+.6e} ref {float(g[r, c]):+.6e}")
        print("  share of the squared error in the 8 largest entries:", float((tv.values ** 2).sum() / (flat ** 2).sum()))
