"""Rectangular wave tiles (64x32 / 32x64 per wave: 128x64 and 64x128 workgroup tiles) against the 64x64 tile of the
large products, on the output layer's three products (GPU only)."""
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


shapes = [(ops.NT, 2560, 5000, 400, "dec out fwd"), (ops.NN, 2560, 400, 5000, "dec out dgrad"),
          (ops.TN, 5000, 400, 2560, "dec out wgrad"), (ops.NT, 2560, 5000, 1600, "fwd shape, K = 1600"),
          (ops.NT, 10240, 5000, 400, "fwd shape, 4 x rows")]
cfgs = [(1, 1, 16, 2, 1), (2, 1, 16, 2, 1), (1, 2, 16, 2, 1), (1, 1, 16, 2, 1), (2, 2, 16, 2, 1), (1, 1, 16, 2, 1)]
for layout, M, N, K, what in shapes:
    if layout == ops.NT:
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    elif layout == ops.NN:
        A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
    else:
        A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
    out = torch.empty(M, N, device=dev)
    ws = torch.empty(min(lib.cradle_gemm_workspace_floats(layout, M, N, K, 0), 1 << 26), device=dev)
    print(f"{what}: {M} x {N} x {K}", flush=True)
    for _ in range(8):  # the chip needs ~100 ms of load before its times settle (the first configuration read 15 % slow)
        ops.gemm(layout, A, B, out=out, workspace=ws, repeat=200)
    torch.cuda.synchronize()
    for tm, tn, bk, dp, wk in cfgs:
        line = f"  wave tile {32*tm}x{32*tn} d{dp}:"
        for sk in (1, 2, 4, 8):
            if sk > max(1, (K + bk - 1) // bk // 2) or (sk > 1 and M * N > 4e6):
                continue
            lib.cradle_gemm_tune(tm, tn, bk, dp, wk, sk)
            t = timeit(lambda: ops.gemm(layout, A, B, out=out, workspace=ws, repeat=REP))
            line += f"  split {sk}: {t*1e6:7.1f} us {2.0*M*N*K/t*1e-12:6.1f} TF"
        print(line, flush=True)
    lib.cradle_gemm_tune(-1, -1, -1, -1, -1, -1)
