"""Every GEMM of one ATE call at the bench's size, by shape: launches, total time, TFLOP/s (HIP events around each launch;
the events serialise nothing here, the call is one stream).  python tools/ate_gemm_shapes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from collections import defaultdict
from cradle_vae_amd import ops, synthetic as syn
from cradle_vae_amd.predictor import cradle_vae_Predictor

shape = syn.SHAPES["norman"]
lm, _ = bench.build(shape, "cuda:0", seed=1000)
pred = cradle_vae_Predictor(lm.model, lm.guide)
D_alt, D_ctl = bench.ate_workload(shape)
kw = dict(dosages_alt=D_alt, dosages_control=D_ctl, quality=torch.ones(1, shape.n_qc), method="perturbseq",
          condition_values=dict(library_size=1e4 * torch.ones(1)), batch_size=bench.ATE_CHUNK)
pred.estimate_average_treatment_effects(n_particles=bench.ATE_CHUNK, **kw)
torch.cuda.synchronize()
ops.KERNEL_TIMERS.clear()
ops.KERNEL_TIMING = "all"
pred.estimate_average_treatment_effects(n_particles=bench.ATE_PARTICLES, **kw)
torch.cuda.synchronize()
ops.KERNEL_TIMING = False
agg = defaultdict(lambda: [0, 0.0])
other = defaultdict(lambda: [0, 0.0])
for t in ops.KERNEL_TIMERS:
    ms = t[1].elapsed_time(t[2])
    (agg if isinstance(t[0], tuple) else other)[t[0]][0] += 1
    (agg if isinstance(t[0], tuple) else other)[t[0]][1] += ms
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    _, M, N, K = k
    print(f"gemm M={M:6d} N={N:5d} K={K:5d}: {n:4d} launches {ms:8.2f} ms  {2.0 * M * N * K * n / ms / 1e9:6.1f} TFLOP/s")
for k, (n, ms) in other.items():
    print(f"{k}: {n} launches {ms:.2f} ms")
