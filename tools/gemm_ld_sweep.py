"""Does the leading dimension matter?  The output layer's forward product (NT 2560 x 5000 x 400) with the operands' rows
400 floats apart (1600 B: every other row starts in the middle of a 128-byte line) against padded rows; and neighbouring
K (GPU only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cradle_vae_amd import ops, hip

dev = "cuda:0"
lib = hip.load()
REP = 20


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / REP * 1e-3)
    return best


M, N = 2560, 5000
for K, ld in [(400, 400), (400, 416), (400, 432), (400, 448), (400, 480), (400, 512), (400, 528), (384, 384), (384, 400), (416, 416), (448, 448), (512, 512), (512, 528)]:
    A = torch.randn(M, ld, device=dev)[:, :K]
    B = torch.randn(N, ld, device=dev)[:, :K]
    for ldc in (5000, 5024):
        out = torch.empty(M, ldc, device=dev)[:, :N]
        t = timeit(lambda: ops.gemm(ops.NT, A, B, out=out, repeat=REP))
        print(f"NT K={K} ld={ld} ldc={ldc}: {t*1e6:7.1f} us {2.0*M*N*K/t*1e-12:6.1f} TF   {t*1e6/K:.4f} us per unit of K", flush=True)
# the data gradient (NN 2560 x 400 x 5000: B = W (5000, 400) read MN-major) and the weight gradient (TN 5000 x 400 x 2560)
for ld in (400, 416, 448, 512):
    A = torch.randn(M, 5000, device=dev)
    W = torch.randn(5000, ld, device=dev)[:, :400]
    out = torch.empty(M, ld, device=dev)[:, :400]
    ws = torch.empty(1 << 25, device=dev)
    t = timeit(lambda: ops.gemm(ops.NN, A, W, out=out, workspace=ws, repeat=REP))
    print(f"NN dgrad ld(W, out)={ld}: {t*1e6:7.1f} us {2.0*M*5000*400/t*1e-12:6.1f} TF", flush=True)
    H = torch.randn(M, ld, device=dev)[:, :400]
    out2 = torch.empty(5000, ld, device=dev)[:, :400]
    t = timeit(lambda: ops.gemm(ops.TN, A, H, out=out2, workspace=ws, repeat=REP))
    print(f"TN wgrad ld(H, out)={ld}: {t*1e6:7.1f} us {2.0*M*5000*400/t*1e-12:6.1f} TF", flush=True)
