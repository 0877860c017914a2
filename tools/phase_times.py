"""Steady-state time of the step's phases without a profiler attached (which makes the run host-bound):
forward only, forward+backward, forward+backward+clip+Adam.  python tools/phase_times.py [shape] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cradle_vae_amd import synthetic as syn
from cradle_vae_amd.trainer import TrainStep

shape = syn.SHAPES[sys.argv[1] if len(sys.argv) > 1 else "norman"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
lm, batch = bench.build(shape, "cuda:0", seed=1000)
st = TrainStep(lm, lr=3e-4, weight_decay=1e-6, gradient_clip_norm=100.0)
f = st.fused
kw = dict(X=batch["X"], D=batch["D"], D_obs_counts=batch["D_obs_counts"], qc=batch["qc"], qc_obs_counts=batch["qc_obs_counts"],
          cfX=batch["cfX"], library_size=batch["library_size"], n_particles=shape.n_particles)


def timeit(fn, n=steps):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def host_only(fn, n=50):
    """Host enqueue time: the GPU is kept idle-free by syncing first, time only the Python/launch calls."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return t


print(f"{shape.name}: forward only          {timeit(lambda: f.run(compute_grads=False, **kw)):8.1f} us/step")
print(f"{shape.name}: forward + backward    {timeit(lambda: f.run(compute_grads=True, **kw)):8.1f} us/step")
print(f"{shape.name}: + clip + Adam         {timeit(lambda: st(batch, n_particles=shape.n_particles)):8.1f} us/step")
print(f"{shape.name}: host enqueue (full)   {host_only(lambda: st(batch, n_particles=shape.n_particles)):8.1f} us/step (first 50 calls after a sync: queue never full)")
noise = lambda: f._set_noise(None, None, st.gen_plated, st.gen_local)
print(f"{shape.name}: noise draws alone     {timeit(noise):8.1f} us/step")
