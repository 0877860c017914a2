"""Device time of each of the first steps after a synchronisation (events on the caller's stream after every step):
where the fixed cost of a short timed region sits.  python tools/first_steps.py [idle_ms]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from cradle_vae_amd import synthetic as syn
from cradle_vae_amd.trainer import TrainStep

idle = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
shape = syn.SHAPES["norman"]
lm, batch = bench.build(shape, "cuda:0", seed=1000)
st = TrainStep(lm, lr=3e-4, weight_decay=1e-6, gradient_clip_norm=100.0)
for _ in range(30):
    st(batch, n_particles=shape.n_particles)
for rep in range(3):
    torch.cuda.synchronize()
    time.sleep(idle * 1e-3)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(25)]
    t0 = time.perf_counter()
    ev[0].record()
    host = []
    for i in range(24):
        st(batch, n_particles=shape.n_particles)
        ev[i + 1].record()
        host.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    d = [ev[i].elapsed_time(ev[i + 1]) for i in range(24)]
    print(f"idle {idle} ms, rep {rep}: wall {wall:.2f} ms for 24 steps = {wall/24:.4f} ms/step; device ms per step: "
          + " ".join(f"{x:.3f}" for x in d))
    print("   host has enqueued step i after (ms): " + " ".join(f"{x:.2f}" for x in host[:12]))
