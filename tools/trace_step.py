"""A few training steps enqueued behind a long blocker kernel, so that under rocprofv3 --kernel-trace the host is
far ahead of the GPU and the trace shows the GPU-side schedule of a step (not the profiler-slowed enqueue):
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python3 tools/trace_step.py [shape]
  python tools/step_timeline.py gpurun_out/trace/t_kernel_trace.csv 2 --full"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cradle_vae_amd import synthetic as syn
from cradle_vae_amd.trainer import TrainStep

shape = syn.SHAPES[sys.argv[1] if len(sys.argv) > 1 else "norman"]
lm, batch = bench.build(shape, "cuda:0", seed=1000)
st = TrainStep(lm, lr=3e-4, weight_decay=1e-6, gradient_clip_norm=100.0)
for _ in range(5):
    st(batch, n_particles=shape.n_particles)
torch.cuda.synchronize()
a = torch.randn(8192, 8192, device="cuda:0")
for _ in range(3):  # ~30 ms of GPU work: time for the host to enqueue the steps below
    a = torch.mm(a, a) * 1e-4
for _ in range(5):
    st(batch, n_particles=shape.n_particles)
torch.cuda.synchronize()
print("done")
