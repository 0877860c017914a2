"""The fp32 MFMA GEMM of this repository against torch.mm (hipBLASLt / rocBLAS fp32) at the step's
shapes: context for the kernel's numbers, not a dependency (the product path never calls torch.mm)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cradle_vae_amd import ops, hip

dev = "cuda:0"
lib = hip.load()
torch.backends.cuda.matmul.allow_tf32 = False
REP = 20


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP * 1e-3


shapes = [(2560, 5000, 400, "dec out fwd (NT)"), (2560, 400, 5000, "dec out dgrad (NN)"), (5000, 400, 2560, "dec out wgrad (TN)"),
          (1024, 400, 5000, "xW fwd (NT)"), (7680, 400, 400, "enc heads fwd (NT)"), (2560, 400, 400, "hidden fwd (NT)"),
          (1050, 400, 400, "emb fwd (NT)")]
for M, N, K, what in shapes:
    if "(NT)" in what:
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
        mine = lambda: ops.gemm(ops.NT, A, B, out=out, workspace=ws)
        vend = lambda: torch.mm(A, B.t(), out=out2)
        layout = ops.NT
    elif "(NN)" in what:
        A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
        mine = lambda: ops.gemm(ops.NN, A, B, out=out, workspace=ws)
        vend = lambda: torch.mm(A, B, out=out2)
        layout = ops.NN
    else:
        A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
        mine = lambda: ops.gemm(ops.TN, A, B, out=out, workspace=ws)
        vend = lambda: torch.mm(A.t(), B, out=out2)
        layout = ops.TN
    out, out2 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    ws = torch.empty(lib.cradle_gemm_workspace_floats(layout, M, N, K, 0), device=dev)
    tm, tv = timeit(mine), timeit(vend)
    err = float((out - out2).abs().max()) / float(out2.abs().max())
    fl = 2.0 * M * N * K
    print(f"{what:22s} {M}x{N}x{K}: this {tm*1e6:7.1f} us {fl/tm/1e12:6.1f} TF | torch.mm {tv*1e6:7.1f} us {fl/tv/1e12:6.1f} TF | rel.diff {err:.1e}", flush=True)
