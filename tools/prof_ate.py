"""One model-side ATE call at the bench's size (236 perturbations x 2500 particles, chunks of 128, perturbseq, counts
drawn) for a kernel-trace profile: rocprofv3 --kernel-trace --stats -d <dir> -o a -- python3 tools/prof_ate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cradle_vae_amd import synthetic as syn
from cradle_vae_amd.predictor import cradle_vae_Predictor

shape = syn.SHAPES["norman"]
lm, batch = bench.build(shape, "cuda:0", seed=1000)
pred = cradle_vae_Predictor(lm.model, lm.guide)
D_alt, D_ctl = bench.ate_workload(shape)
kw = dict(dosages_alt=D_alt, dosages_control=D_ctl, quality=torch.ones(1, shape.n_qc), method="perturbseq",
          condition_values=dict(library_size=1e4 * torch.ones(1)), batch_size=bench.ATE_CHUNK)
pred.estimate_average_treatment_effects(n_particles=bench.ATE_CHUNK, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
pred.estimate_average_treatment_effects(n_particles=bench.ATE_PARTICLES, **kw)
torch.cuda.synchronize()
print(f"ATE call: {(time.perf_counter() - t0) * 1e3:.1f} ms (under the profiler if one is attached)")
