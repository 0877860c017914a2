"""The fused step at gene counts around the Norman and Replogle widths, one per residue mod 4 (GPU only):
python tools/bench_ragged.py [steps]"""
import dataclasses, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from cradle_vae_amd import synthetic as syn
from cradle_vae_amd.trainer import TrainStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for base in ("norman", "replogle"):
    g0 = syn.SHAPES[base].n_phenos
    for precision in ("f32", "bf16x3"):
        for G in (g0, g0 + 1, g0 + 2, g0 + 3, g0):
            shape = dataclasses.replace(syn.SHAPES[base], name=f"{base}_g{G}", n_phenos=G)
            lm, batch = bench.build(shape, "cuda:0", seed=1000)
            lm.precision = precision
            st = TrainStep(lm, lr=3e-4, weight_decay=1e-6, gradient_clip_norm=100.0, world_size=1)
            assert st.fused is not None
            for _ in range(10):
                st(batch, n_particles=shape.n_particles)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                st(batch, n_particles=shape.n_particles)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            print(json.dumps({"shape": base, "precision": precision, "n_phenos": G, "ms_per_step": round(dt * 1e3, 4),
                              "cells_per_s": round(shape.n_cells / dt)}), flush=True)
            del st, lm, batch
