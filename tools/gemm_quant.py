"""Tile-quantisation probe of the fp32 MFMA GEMM (64x64 tiles, 8 workgroups per CU = 2048 resident): the output layer's
product at row counts that fill whole rounds of resident workgroups and at the step's own.  python tools/gemm_quant.py"""
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


for K in (400, 1600):
    for M in (1664, 2560, 3328, 4992, 5120, 10240):
        N = 5000
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
        out = torch.empty(M, N, device=dev)
        ws = torch.empty(hip.load().cradle_gemm_workspace_floats(ops.NT, M, N, K, 0), device=dev)
        t = timeit(lambda: ops.gemm(ops.NT, A, B, out=out, workspace=ws))
        t_ref = timeit(lambda: torch.matmul(A, B.t(), out=out))
        fl = 2.0 * M * N * K
        tiles = ((M + 63) // 64) * ((N + 63) // 64)
        print(f"NT {M:5d}x{N}x{K:4d}: {tiles:5d} tiles = {tiles/2048:5.2f} rounds  {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF | "
              f"torch {t_ref*1e6:8.1f} us {fl/t_ref/1e12:6.1f} TF", flush=True)
