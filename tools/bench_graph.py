"""Step time at a bench shape: eager 3-stream launch vs HIP-graph replay (whole / split around K1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cradle_vae_amd import synthetic as syn
from cradle_vae_amd.trainer import TrainStep

shape = syn.SHAPES[sys.argv[1] if len(sys.argv) > 1 else "norman"]
lm, batch = bench.build(shape, "cuda:0", seed=1000)
st = TrainStep(lm, lr=3e-4)
def run(n):
    for _ in range(n):
        st(batch, n_particles=shape.n_particles)
for mode in ("eager", "graph", "graph+k1events", "eager"):
    st.fused.use_graph = mode.startswith("graph")
    st.fused.k1_events = [] if "k1" in mode else None
    run(10); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(100); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    t1 = time.perf_counter(); run(100); host = (time.perf_counter() - t1) / 100; torch.cuda.synchronize()
    print(f"{mode:16s} {dt*1e3:.3f} ms/step  {shape.n_cells/dt:,.0f} cells/s   host enqueue {host*1e3:.3f} ms/step", flush=True)
