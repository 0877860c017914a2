"""20000 training steps at the Norman shape: sustained rate, device / host memory growth, loss.  python tools/soak.py"""
import os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from cradle_vae_amd import synthetic as syn
from cradle_vae_amd.trainer import TrainStep
shape = syn.NORMAN
lm, batch = bench.build(shape, "cuda:0", seed=1)
st = TrainStep(lm, lr=3e-4, weight_decay=1e-6, gradient_clip_norm=100.0)
def rss(): return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
for i in range(50): st(batch, n_particles=5)
torch.cuda.synchronize()
m0, r0 = torch.cuda.memory_allocated(), rss()
t0 = time.perf_counter()
for i in range(20000):
    loss = st(batch, n_particles=5)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"20000 steps: {dt:.1f} s = {20000*512/dt:.0f} cells/s, loss {float(loss):.3f}, device mem {m0/1e6:.1f} -> {torch.cuda.memory_allocated()/1e6:.1f} MB, host RSS {r0:.0f} -> {rss():.0f} MB")
