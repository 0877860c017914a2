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

shape = syn.SHAPES[sys.argv[1] if len(sys.argv) > 1 else "norman"]
n_cells = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
P = int(sys.argv[3]) if len(sys.argv) > 3 else 100
bs = int(sys.argv[4]) if len(sys.argv) > 4 else 128
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
