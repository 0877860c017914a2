"""A few launches of the fp32 MFMA GEMM at the step's shapes, for rocprofv3 --pmc passes.
python3 tools/pmc_gemm.py   (shapes: decoder output layer fwd / dgrad / wgrad, encoder trunk, hidden layer)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cradle_vae_amd import hip, ops  # noqa: E402

dev = "cuda:0"
torch.manual_seed(0)
lib = hip.load()
shapes = [(ops.NT, 2560, 5000, 400), (ops.NN, 2560, 400, 5000), (ops.TN, 5000, 400, 2560), (ops.NT, 7680, 400, 400),
          (ops.NT, 2560, 400, 400), (ops.NT, 1050, 400, 400)]
for layout, M, N, K in shapes:
    if layout == ops.NT:
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    elif layout == ops.NN:
        A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
    else:
        A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
    out = torch.empty(M, N, device=dev)
    ws = torch.empty(lib.cradle_gemm_workspace_floats(layout, M, N, K, 0), device=dev)
    for _ in range(5):
        ops.gemm(layout, A, B, out=out, workspace=ws)
    torch.cuda.synchronize()
print("done")
