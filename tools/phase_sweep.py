"""Cumulative steady-state cost of the forward phases of the fused step, measured without a profiler by
truncating the step after each internal checkpoint (cradle_step_debug_stop_after).
python tools/phase_sweep.py [shape] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cradle_vae_amd import hip, synthetic as syn
from cradle_vae_amd.trainer import TrainStep

shape = syn.SHAPES[sys.argv[1] if len(sys.argv) > 1 else "norman"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
lm, batch = bench.build(shape, "cuda:0", seed=1000)
st = TrainStep(lm, lr=3e-4, weight_decay=1e-6, gradient_clip_norm=100.0)
f, lib = st.fused, hip.load()
kw = dict(X=batch["X"], D=batch["D"], D_obs_counts=batch["D_obs_counts"], qc=batch["qc"], qc_obs_counts=batch["qc_obs_counts"],
          cfX=batch["cfX"], library_size=batch["library_size"], n_particles=shape.n_particles)
f._checked_states.add((shape.n_particles, shape.n_cells, f.arena.base))


def timeit(fn, n=steps):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


names = {1: "noise + plan/side streams + q(M) sample", 2: "+ embedding encoder (5 GEMMs)", 3: "+ gather, offset GEMM, join (xW, qc ready)",
         4: "+ enc combine + encoder heads GEMM", 5: "+ zhead + decoder hidden layers", 6: "+ decoder output layer", 0: "+ K1 + finalize (whole forward)"}
f.run(compute_grads=True, **kw)
prev = 0.0
for k in (1, 2, 3, 4, 5, 6, 0):
    lib.cradle_step_debug_stop_after(k)
    f._calls = 1  # no periodic flag check (the truncated step leaves flags/out incomplete)
    t = timeit(lambda: f.run(compute_grads=True, **kw) if k else f.run(compute_grads=False, **kw))
    print(f"checkpoint {k}: {t:8.1f} us cumulative  (+{t - prev:6.1f})  {names[k]}")
    prev = t
lib.cradle_step_debug_stop_after(0)
