"""Fixed vs per-K cost of the small-problem GEMM kernels: time over K at fixed M, N (GPU only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cradle_vae_amd import ops, hip

dev = "cuda:0"
lib = hip.load()
REP = 20


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP * 1e3


for M, N in ((2560, 400), (1050, 400), (7680, 400), (5, 400)):
    row = []
    for K in (32, 100, 200, 400, 800, 1600):
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
        out = torch.empty(M, N, device=dev)
        ws = torch.empty(lib.cradle_gemm_workspace_floats(ops.NT, M, N, K, 0), device=dev)
        row.append(f"K={K}: {timeit(lambda: ops.gemm(ops.NT, A, B, out=out, workspace=ws, repeat=REP)):6.1f}us")
    print(f"NT {M}x{N}: " + "  ".join(row), flush=True)
