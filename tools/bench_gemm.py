"""Micro-benchmark of the fp32 MFMA GEMM at the step's shapes.  python tools/bench_gemm.py"""
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


shapes = [  # layout, M, N, K, what
    (ops.NT, 2560, 5000, 400, "decoder out fwd"),
    (ops.NN, 2560, 400, 5000, "decoder out dgrad"),
    (ops.TN, 5000, 400, 2560, "decoder out wgrad"),
    (ops.NT, 1024, 400, 5000, "xW fwd"),
    (ops.TN, 400, 5000, 1024, "xW wgrad"),
    (ops.NT, 2560, 400, 400, "hidden fwd"),
    (ops.NN, 2560, 400, 400, "hidden dgrad"),
    (ops.TN, 400, 400, 2560, "hidden wgrad"),
    (ops.NT, 7680, 400, 400, "enc heads fwd"),
    (ops.TN, 400, 400, 7680, "enc heads wgrad"),
    (ops.NT, 1050, 400, 400, "emb enc fwd"),
    (ops.NT, 5120, 400, 200, "offW fwd"),
    (ops.NT, 4096, 4096, 4096, "square 4096"),
]
for layout, M, N, K, what in shapes:
    if layout == ops.NT:
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    elif layout == ops.NN:
        A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
    else:
        A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
    out = torch.empty(M, N, device=dev)
    ws = torch.empty(hip.load().cradle_gemm_workspace_floats(layout, M, N, K, 0), device=dev)
    t = timeit(lambda: ops.gemm(layout, A, B, out=out, workspace=ws))
    if layout == ops.NT:
        t_ref = timeit(lambda: torch.matmul(A, B.t(), out=out))
    elif layout == ops.NN:
        t_ref = timeit(lambda: torch.matmul(A, B, out=out))
    else:
        t_ref = timeit(lambda: torch.matmul(A.t(), B, out=out))
    fl = 2.0 * M * N * K
    print(f"{what:20s} L{layout} {M:5d}x{N:5d}x{K:5d}  {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF | torch(hipBLASLt) {t_ref*1e6:8.1f} us {fl/t_ref/1e12:6.1f} TF")
