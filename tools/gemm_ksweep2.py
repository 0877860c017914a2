"""t(K) of the output layer's forward product (NT 2560 x 5000 x K): where the time that is not proportional to K goes.
python tools/gemm_ksweep2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cradle_vae_amd import ops, hip

dev = "cuda:0"


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


M, N = 2560, 5000
prev = None
for K in (16, 32, 64, 128, 256, 400, 800, 1600):
    A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    out = torch.empty(M, N, device=dev)
    t = timeit(lambda: ops.gemm(ops.NT, A, B, out=out, workspace=None))
    fl = 2.0 * M * N * K
    msg = f"K={K:5d}: {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF"
    if prev:
        dk, dt = K - prev[0], t - prev[1]
        msg += f"   marginal {2.0*M*N*dk/dt/1e12:6.1f} TF over the last {dk} of K"
    print(msg, flush=True)
    prev = (K, t)
