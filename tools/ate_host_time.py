"""Host time to enqueue one ATE call against its device time (GPU only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cradle_vae_amd import synthetic as syn
from cradle_vae_amd.predictor import cradle_vae_Predictor

dev = torch.device("cuda:0")
shape = syn.SHAPES["norman"]
lm, batch = bench.build(shape, dev, seed=1000)
pred = cradle_vae_Predictor(lm.model, lm.guide)
D_alt, D_ctl = bench.ate_workload(shape)
kw = dict(dosages_alt=D_alt, dosages_control=D_ctl, quality=torch.ones(1, shape.n_qc), method="perturbseq",
          condition_values=dict(library_size=1e4 * torch.ones(1)), batch_size=bench.ATE_CHUNK)
pred.estimate_average_treatment_effects(n_particles=bench.ATE_CHUNK, **kw)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    pred.estimate_average_treatment_effects(n_particles=bench.ATE_PARTICLES, **kw)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print(f"draws beside the next decoder pass: {sorted(ts)[1]*1e3:.1f} ms", flush=True)
from cradle_vae_amd import parallel
orig = parallel.allreduce_sum_
stamp = {}


def stamped(t, world):
    stamp["t"] = time.perf_counter()  # the loop's last launch has been enqueued
    return orig(t, world)


parallel.allreduce_sum_ = stamped
import cradle_vae_amd.ate as ate_mod
ate_mod.parallel.allreduce_sum_ = stamped
os.environ["CRADLE_ATE_OVERLAP"] = "0"
for _ in range(3):
    t0 = time.perf_counter()
    pred.estimate_average_treatment_effects(n_particles=bench.ATE_PARTICLES, **kw)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"one stream: host has enqueued everything after {(stamp['t']-t0)*1e3:.1f} ms, call done after {(t1-t0)*1e3:.1f} ms", flush=True)
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    pred.estimate_average_treatment_effects(n_particles=bench.ATE_PARTICLES, **kw)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
