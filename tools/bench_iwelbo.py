"""Predictive IWELBO (predictor.compute_predictive_iwelbo, predictor.py:77-200) at a bench shape:
python tools/bench_iwelbo.py [shape] [n_cells] [n_particles] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cradle_vae_amd import synthetic as syn
from cradle_vae_amd.data import DeviceBatchLoader, SCRNASeqTensorPerturbationDataset
from cradle_vae_amd.predictor import cradle_vae_Predictor
import dataclasses

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
shape = syn.SHAPES[argv[0] if len(argv) > 0 else "norman"]
n_cells = int(argv[1]) if len(argv) > 1 else 4096
P = int(argv[2]) if len(argv) > 2 else 100
bs = int(argv[3]) if len(argv) > 3 else 128
lm, _ = bench.build(shape, "cuda:0", seed=1000)
b = syn.make_batch(dataclasses.replace(shape, n_cells=n_cells), seed=3, device="cuda:0")
ds = SCRNASeqTensorPerturbationDataset(b["X"], b["D"], b["qc"], b["cfX"])
loader = DeviceBatchLoader(ds, bs, shuffle=False, device="cuda:0")
pred = cradle_vae_Predictor(lm.model, lm.guide)
pred.compute_predictive_iwelbo(loader, n_particles=P)
torch.cuda.synchronize()
t0 = time.perf_counter()
df = pred.compute_predictive_iwelbo(loader, n_particles=P)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{shape.name} IWELBO: {n_cells} cells x {P} particles, batch {bs}: {dt*1e3:.1f} ms = {n_cells/dt:.0f} cells/s "
      f"({n_cells*P/dt/1e6:.2f} M particle-rows/s); mean {float(df['IWELBO'].mean()):.3f}")
if "--profile" in sys.argv:  # where the host's time goes (a host that waits for the device shows up as a sync here)
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    pred.compute_predictive_iwelbo(loader, n_particles=P)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
