"""End-to-end training-loop throughput (trainer.fit over a device-resident data set) at a bench shape,
to compare with bench.py's step rate.  python tools/bench_fit.py [shape] [n_cells] [epochs]"""
import dataclasses, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cradle_vae_amd import synthetic as syn
from cradle_vae_amd.data import DeviceBatchLoader, SCRNASeqTensorPerturbationDataset
from cradle_vae_amd.guide import ObservationNormalizationStatistics
from cradle_vae_amd.models.utils.perturbation_lightning_module import TrainConfigPerturbationLightningModule
from cradle_vae_amd.trainer import fit

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
shape = syn.SHAPES[argv[0] if len(argv) > 0 else "norman"]
n_cells = int(argv[1]) if len(argv) > 1 else 32768
epochs = int(argv[2]) if len(argv) > 2 else 2
dev = "cuda:0"
data = syn.make_batch(dataclasses.replace(shape, n_cells=n_cells), seed=2)
ds = SCRNASeqTensorPerturbationDataset(data["X"], data["D"], data["qc"], data["cfX"])
s = shape
config = dict(
    model="cradle_vae_Model", guide="cradle_vae_CorrelatedNormalGuide", loss_module="cradle_vae_ELBOLossModule",
    predictor="cradle_vae_Predictor", data_module="NormanOODCombinationDataModule", data_module_kwargs=dict(n_qc_pass=80, n_qc_fail=20),
    lightning_module_kwargs=dict(lr=3e-4, n_particles=s.n_particles), loss_module_kwargs=dict(beta=s.beta),
    model_kwargs=dict(n_latent=s.n_latent, n_treatments=s.n_treatments, n_phenos=s.n_phenos, n_qc=s.n_qc,
                      mask_prior_prob=s.mask_prior_prob, embedding_prior_scale=1.0, likelihood_key="library_nb",
                      decoder_n_layers=s.decoder_n_layers, decoder_n_hidden=s.decoder_n_hidden, sum_mode=False),
    guide_kwargs=dict(n_latent=s.n_latent, n_treatments=s.n_treatments, n_phenos=s.n_phenos, n_qc=s.n_qc,
                      basal_encoder_n_layers=s.basal_encoder_n_layers, basal_encoder_n_hidden=s.basal_encoder_n_hidden,
                      basal_encoder_input_normalization="log_standardize",
                      embedding_encoder_n_layers=s.embedding_encoder_n_layers,
                      embedding_encoder_n_hidden=s.embedding_encoder_n_hidden,
                      x_normalization_stats=ObservationNormalizationStatistics(log_x_mean=data["log_x_mean"], log_x_std=data["log_x_std"])))
counts = dict(D_obs_counts_train=ds.get_dosage_obs_per_dim().float(), qc_obs_counts_train=ds.get_qc_obs_per_dim().float())
counts.update({k.replace("train", sp): v for k, v in list(counts.items()) for sp in ("val", "test")})
mod = TrainConfigPerturbationLightningModule(config=config, **counts).to(dev)
train = DeviceBatchLoader(ds, batch_size=shape.n_cells, shuffle=True, drop_last=True, seed=0, device=dev)
fit(mod, train, None, max_epochs=1, max_steps=10, gradient_clip_val=100.0)  # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
hist = fit(mod, train, None, max_epochs=epochs, gradient_clip_val=100.0)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
steps = hist[-1]["steps"]
print(f"{shape.name} fit: {steps} steps of {shape.n_cells} cells in {dt*1e3:.1f} ms = {steps*shape.n_cells/dt:.0f} cells/s "
      f"({dt/steps*1e3:.3f} ms/step); train/loss {hist[0]['train/loss']:.2f} -> {hist[-1]['train/loss']:.2f}")
if "--profile" in sys.argv:  # where the host's time goes in the loop
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    fit(mod, train, None, max_epochs=1, gradient_clip_val=100.0)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(16)
