"""fp32 MFMA vs split-bf16 (precision="bf16x3") at the step's saturating GEMM shapes (GPU only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cradle_vae_amd import ops, hip

dev, lib, REP = "cuda:0", hip.load(), 20


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP * 1e-3


shapes = [(ops.NT, 2560, 5000, 400, "dec out fwd"), (ops.NN, 2560, 400, 5000, "dec out dgrad"), (ops.TN, 5000, 400, 2560, "dec out wgrad"),
          (ops.NT, 1024, 400, 5000, "xW fwd"), (ops.TN, 400, 5000, 1024, "xW wgrad"), (ops.NT, 7680, 400, 400, "enc heads fwd"),
          (ops.NN, 7680, 400, 400, "enc heads dgrad"), (ops.TN, 400, 400, 7680, "enc heads wgrad"),
          (ops.NT, 2560, 8000, 400, "replogle out fwd"), (ops.NN, 2560, 400, 8000, "replogle out dgrad"), (ops.TN, 8000, 400, 2560, "replogle out wgrad"),
          (ops.NT, 10240, 400, 400, "replogle emb fwd"), (ops.NT, 2560, 400, 400, "hidden fwd (0.8 GF: stays fp32)")]
tiles = [(-1, -1), (2, 2), (2, 1), (1, 2), (1, 1)] if "--tiles" in sys.argv else [(-1, -1)]
splits = [int(a) for a in sys.argv[1:] if a != "--tiles"] or [-1]
for layout, M, N, K, what in shapes:
    if layout == ops.NT: A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    elif layout == ops.NN: A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
    else: A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
    out = torch.empty(M, N, device=dev)
    ws = torch.empty(min(lib.cradle_gemm_workspace_floats(layout, M, N, K, 0), 1 << 26), device=dev)
    fl = 2.0 * M * N * K
    t32 = timeit(lambda: ops.gemm(layout, A, B, out=out, workspace=ws, repeat=REP))
    line = f"{what:32s} L{layout} {M}x{N}x{K}: f32 {t32*1e6:6.1f} us {fl/t32/1e12:5.1f} TF |"
    for tm, tn in tiles:
      for sk in splits:
        lib.cradle_gemm_tune(tm, tn, 32, 2, 1, sk)
        tb = timeit(lambda: ops.gemm(layout, A, B, out=out, workspace=ws, repeat=REP, precision="bf16x3"))
        line += f" | t{tm}{tn} s{sk} {tb*1e6:6.1f} us x{t32/tb:.2f}"
    lib.cradle_gemm_tune(-1, -1, -1, -1, -1, -1)
    print(line, flush=True)
