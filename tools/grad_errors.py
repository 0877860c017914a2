"""Per-tensor gradient errors of the fused step against the fp64 oracle at a full-size shape (diagnostic).
python tools/grad_errors.py [norman|replogle] [batch seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cradle_vae_amd import synthetic as syn
from oracle import cradle_oracle as co
from tests.test_fused_step_gpu import run_fused
from tests.test_modular_gpu import build_loss_module

name = sys.argv[1] if len(sys.argv) > 1 else "norman"
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 17
DEV = "cuda:0"
shape = syn.SHAPES[name]
b = syn.make_batch(shape, seed=SEED)
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
kw = lambda dt: dict(X=b["X"].to(dt), D=b["D"].to(dt), qc=b["qc"].to(dt), cfX=b["cfX"].to(dt),
                     library_size=b["library_size"].to(dt), D_obs_counts=b["D_obs_counts"].to(dt),
                     qc_obs_counts=b["qc_obs_counts"].to(dt), noise={k: v.to(dt) for k, v in noise.items()},
                     cf_noise={k: v.to(dt) for k, v in cf_noise.items()}, n_particles=cfg["n_particles"],
                     beta=cfg["beta"], alpha=cfg["alpha"])
torch.set_num_threads(16)
conv = lambda dt: {k: (v.to(dt) if v.is_floating_point() else v) for k, v in params.items()}
l64, m64, i64, g64 = co.elbo_loss_and_grads(conv(torch.float64), **kw(torch.float64))
l32, m32, i32_, g32 = co.elbo_loss_and_grads(conv(torch.float32), **kw(torch.float32))
dev = lambda d: {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in d.items()}
case = dict(cfg=cfg, params=dev(params), batch=dev(b), noise=dev(noise), cf_noise=dev(cf_noise))
lm = build_loss_module(case)
lm.max_dosages_per_cell = 2
loss, metrics, grads = run_fused(lm, case)
print(f"loss hip {float(loss):.6f} torch32 {float(l32):.6f} f64 {float(l64):.6f}")
print(f"{'tensor':55s} {'hip L2':>10s} {'hip max':>10s} {'torch-fp32 L2':>14s} {'torch-fp32 max':>14s}")
for k, g in g64.items():
    a = grads[k].detach().cpu().double()
    t = g32[k].double()
    n, mx = max(float(g.norm()), 1e-300), max(float(g.abs().max()), 1e-300)
    print(f"{k:55s} {float((a-g).norm())/n:10.2e} {float((a-g).abs().max())/mx:10.2e} {float((t-g).norm())/n:14.2e} {float((t-g).abs().max())/mx:14.2e}")

fs = lm._fused
P, N, L, Hd = cfg["n_particles"], shape.n_cells, shape.n_latent, shape.decoder_n_hidden
g = g64["model.decoder.mlp.layers.0.0.weight"]
a = grads["model.decoder.mlp.layers.0.0.weight"].detach().cpu().double()
for nm, sl in (("z", slice(0, L)), ("off", slice(L, 2 * L)), ("qc", slice(2 * L, 3 * L))):
    print(f"dec W0 grad block {nm}: L2 {float((a[:, sl]-g[:, sl]).norm())/float(g[:, sl].norm()):.2e}  norm {float(g[:, sl].norm()):.3e}")
da0 = fs.intermediate("dec_da0").view(P * N, Hd).double()
zoff = fs.intermediate("z_offset").view(P * N, 2 * L).double()
chk = (da0.t() @ zoff).cpu()
print("HIP dW0[:, :2L] vs fp64 product of HIP's own da0, zoff:", float((a[:, :2 * L] - chk).norm()) / float(chk.norm()))
print("fp64 product of HIP's da0, zoff vs oracle:", float((g[:, :2 * L] - chk).norm()) / float(g[:, :2*L].norm()))
gb = g64["model.decoder.mlp.layers.0.0.bias"]
ab = grads["model.decoder.mlp.layers.0.0.bias"].detach().cpu().double()
cb = da0.sum(0).cpu()
print("HIP db0 vs colsum of HIP da0:", float((ab - cb).norm()) / float(cb.norm()), " colsum(HIP da0) vs oracle:", float((gb - cb).norm()) / float(gb.norm()))
print("sum|da0| / |sum da0| (cancellation):", float(da0.abs().sum(0).norm()) / float(da0.sum(0).norm()))

# forward: z, offsets and the decoder's first layer against the oracle (fp64) and torch-fp32
def pre0(inter, p, dt):
    z = inter["z_basal"].to(dt)
    off = torch.matmul(b["D"].to(dt), (inter["E"] * inter["mask"]).to(dt))
    qs = inter["qc_E"].to(dt).sum(1, keepdim=True) * b["qc"].to(dt).max(1)[0].reshape(1, -1, 1)
    zin = torch.cat([z, off, qs.expand(-1, -1, L) if qs.shape[-1] != L else qs], -1)
    W, bb = p["model.decoder.mlp.layers.0.0.weight"].to(dt), p["model.decoder.mlp.layers.0.0.bias"].to(dt)
    return z, off, zin @ W.t() + bb
z64, off64, p64 = pre0(i64, params, torch.float64)
z32, off32, p32 = pre0(i32_, params, torch.float32)
zh, offh = zoff.view(P, N, 2 * L)[..., :L].cpu(), zoff.view(P, N, 2 * L)[..., L:].cpu()
print("z   : hip max abs err %.3e (rel L2 %.2e)   torch32 %.3e (%.2e)" % (float((zh - z64).abs().max()), float((zh - z64).norm() / z64.norm()), float((z32.double() - z64).abs().max()), float((z32.double() - z64).norm() / z64.norm())))
print("off : hip max abs err %.3e (rel L2 %.2e)   torch32 %.3e (%.2e)" % (float((offh - off64).abs().max()), float((offh - off64).norm() / off64.norm()), float((off32.double() - off64).abs().max()), float((off32.double() - off64).norm() / off64.norm())))
y0h = fs.intermediate("dec_y0").view(P, N, Hd).cpu().double()
print("decoder layer-0 sign flips: hip %d, torch32 %d of %d; min |pre0| %.2e" % (int(((y0h > 0) != (p64 > 0)).sum()), int(((p32 > 0) != (p64 > 0)).sum()), p64.numel(), float(p64.abs().min())))
act64 = torch.where(p64 > 0, p64, 0.01 * p64)
print("y0 : hip max abs err %.3e   torch32 %.3e" % (float((y0h - act64).abs().max()), float((torch.where(p32 > 0, p32, 0.01 * p32).double() - act64).abs().max())))
