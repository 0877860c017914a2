"""Model-side ATE estimation (predictor.estimate_average_treatment_effects) at a bench shape:
python tools/bench_ate.py [shape] [n_alt] [n_particles] [chunk]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cradle_vae_amd import synthetic as syn
from cradle_vae_amd.predictor import cradle_vae_Predictor

shape = syn.SHAPES[sys.argv[1] if len(sys.argv) > 1 else "norman"]
n_alt = int(sys.argv[2]) if len(sys.argv) > 2 else 236
n_particles = int(sys.argv[3]) if len(sys.argv) > 3 else 2500
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 128
lm, batch = bench.build(shape, "cuda:0", seed=1000)
pred = cradle_vae_Predictor(lm.model, lm.guide)
T = shape.n_treatments
g = torch.Generator().manual_seed(0)
D_alt = torch.zeros(n_alt, T)
D_alt[torch.arange(n_alt), torch.randint(0, T, (n_alt,), generator=g)] = 1.0
D_alt[torch.arange(n_alt), torch.randint(0, T, (n_alt,), generator=g)] = 1.0
D_ctl = torch.zeros(1, T)
kw = dict(dosages_alt=D_alt, dosages_control=D_ctl, quality=torch.ones(1, shape.n_qc), n_particles=n_particles,
          condition_values=dict(library_size=1e4 * torch.ones(1)), batch_size=chunk)
from cradle_vae_amd import ops
for method, sampler, precision in (("mean", "nb", "f32"), ("mean", "mean", "f32"), ("perturbseq", "nb", "f32"), ("perturbseq", "mean", "f32"),
                                   ("perturbseq", "nb", "bf16x3"), ("perturbseq", "mean", "bf16x3")):
    with ops.gemm_precision(precision):
        pred.estimate_average_treatment_effects(method=method, sampler=sampler, **dict(kw, n_particles=chunk))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = pred.estimate_average_treatment_effects(method=method, sampler=sampler, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rows = n_particles * (n_alt + 1)
        if sampler == "mean":
            if precision == "f32":
                ref_mean = out.X
            else:
                print(f"  bf16x3 against f32, sampler=mean: max |diff| / max |ATE| = {abs(out.X - ref_mean).max() / abs(ref_mean).max():.2e}")
        print(f"{shape.name} ATE method={method:10s} sampler={sampler:4s} gemm={precision:6s}: {dt*1e3:8.1f} ms for {n_alt} perturbations x {n_particles} particles "
              f"(chunk {chunk}) = {rows/dt/1e6:6.2f} M decoded+sampled rows/s, {rows*shape.n_phenos/dt/1e9:6.1f} G samples/s; "
              f"finite={bool(torch.isfinite(torch.from_numpy(out.X)).all())}", flush=True)
