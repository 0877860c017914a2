"""How long the host takes to enqueue one training step (all of its launches, events and the optimiser) against how
long the GPU takes to run it: python tools/host_time.py [shape] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from cradle_vae_amd import synthetic as syn
from cradle_vae_amd.trainer import TrainStep

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
shape = syn.SHAPES[argv[0] if len(argv) > 0 else "norman"]
steps = int(argv[1]) if len(argv) > 1 else 8
lm, batch = bench.build(shape, "cuda:0", seed=1000)
st = TrainStep(lm, lr=3e-4, weight_decay=1e-6, gradient_clip_norm=100.0)
for _ in range(20):
    st(batch, n_particles=shape.n_particles)
torch.cuda.synchronize()
host, total = [], []
for _ in range(10):
    t0 = time.perf_counter()
    for _ in range(steps):
        st(batch, n_particles=shape.n_particles)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) / steps * 1e3)
    total.append((t2 - t0) / steps * 1e3)
host.sort(); total.sort()
print(f"{shape.name}: host enqueue {host[len(host)//2]:.3f} ms/step (min {host[0]:.3f}), enqueue+drain {total[len(total)//2]:.3f} ms/step "
      f"over bursts of {steps} steps")
if "--profile" in sys.argv:  # the Python side of a step, by function (the C library's launches are one ctypes call)
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        st(batch, n_particles=shape.n_particles)
        if _ % 6 == 5:
            torch.cuda.synchronize()  # keep the queue short: a full queue would show up as host time
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
