"""Sweep tile / K-step / K-split of the fp32 MFMA GEMM at the ELBO step's shapes (GPU only)."""
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
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP * 1e-3


shapes = [
    (ops.NT, 2560, 5000, 400, "dec out fwd"), (ops.NN, 2560, 400, 5000, "dec out dgrad"),
    (ops.TN, 5000, 400, 2560, "dec out wgrad"), (ops.NT, 1024, 400, 5000, "xW fwd"),
    (ops.TN, 400, 5000, 1024, "xW wgrad"), (ops.NT, 2560, 400, 400, "hidden fwd"),
    (ops.NN, 2560, 400, 400, "hidden dgrad"), (ops.TN, 400, 400, 2560, "hidden wgrad"),
    (ops.NT, 7680, 400, 400, "enc heads fwd"), (ops.NN, 7680, 400, 400, "enc heads dgrad"),
    (ops.TN, 400, 400, 7680, "enc heads wgrad"), (ops.NT, 1050, 400, 400, "emb fwd"),
    (ops.NN, 1050, 400, 400, "emb dgrad"), (ops.TN, 400, 400, 1050, "emb wgrad"),
    (ops.NT, 2560, 400, 200, "offW fwd"), (ops.NN, 5120, 200, 400, "offW dgrad"), (ops.TN, 400, 200, 2560, "offW wgrad"),
    (ops.NT, 1, 400, 400, "qc enc"), (ops.NN, 5, 400, 400, "qc dgrad"), (ops.TN, 400, 400, 5, "qc wgrad"),
    (ops.NT, 1050, 400, 200, "emb l0 fwd"), (ops.NN, 1050, 200, 400, "emb dmask"), (ops.TN, 400, 200, 1050, "emb l0 wgrad"),
    (ops.NT, 5, 400, 200, "wq fwd"),
]
replogle = [  # N=512, P=5, G=8000, L=100, He=200, Hd=Hm=400, 1024 treatment slots
    (ops.NT, 2560, 8000, 400, "dec out fwd"), (ops.NN, 2560, 400, 8000, "dec out dgrad"),
    (ops.TN, 8000, 400, 2560, "dec out wgrad"), (ops.NT, 1024, 200, 8000, "xW fwd"),
    (ops.TN, 200, 8000, 1024, "xW wgrad"), (ops.NT, 7680, 200, 200, "enc heads fwd"),
    (ops.NN, 7680, 200, 200, "enc heads dgrad"), (ops.TN, 200, 200, 7680, "enc heads wgrad"),
    (ops.NT, 10240, 400, 400, "emb fwd"), (ops.NN, 10240, 400, 400, "emb dgrad"), (ops.TN, 400, 400, 10240, "emb wgrad"),
    (ops.NT, 10240, 400, 100, "emb l0 fwd"), (ops.NT, 10240, 200, 400, "emb heads fwd"),
    (ops.NN, 10240, 100, 400, "emb dmask"), (ops.TN, 400, 100, 10240, "emb l0 wgrad"),
]
huge = [  # batches of 1024 / 2048 cells at G = 8000
    (ops.NT, 5120, 8000, 400, "dec out fwd 1k"), (ops.NN, 5120, 400, 8000, "dec out dgrad 1k"),
    (ops.TN, 8000, 400, 5120, "dec out wgrad 1k"), (ops.NT, 10240, 8000, 400, "dec out fwd 2k"),
    (ops.NN, 10240, 400, 8000, "dec out dgrad 2k"), (ops.TN, 8000, 400, 10240, "dec out wgrad 2k"),
]
args = sys.argv[1:]
if args and args[0] == "--huge":
    shapes, args = huge, args[1:]
if args and args[0] == "--replogle":
    shapes, args = replogle, args[1:]
if args:
    shapes = [s for s in shapes if any(a in s[4] for a in args)]
cfgs = [(2, 2, 16, 2, 1), (1, 1, 16, 2, 1), (1, 1, 16, 4, 1), (1, 1, 32, 2, 1), (1, 1, 32, 2, 2), (1, 1, 32, 4, 2), (1, 2, 32, 2, 2), (2, 2, 32, 2, 2),
        (1, 1, 32, 2, 4), (1, 1, 32, 4, 4), (1, 1, 32, 8, 4), (1, 1, 64, 2, 4), (1, 1, 64, 4, 4), (1, 1, 32, 4, 1), (1, 2, 32, 2, 4), (2, 2, 32, 2, 4)]
splits = [1, 2, 4, 8, 16]
for layout, M, N, K, what in shapes:
    if layout == ops.NT:
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    elif layout == ops.NN:
        A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
    else:
        A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
    out = torch.empty(M, N, device=dev)
    ws = torch.empty(min(lib.cradle_gemm_workspace_floats(layout, M, N, K, 0), 1 << 26), device=dev)
    lib.cradle_gemm_tune(-1, -1, -1, -1, -1, -1)
    t_auto = timeit(lambda: ops.gemm(layout, A, B, out=out, workspace=ws, repeat=REP))
    res = []
    for tm, tn, bk, dp, wk in cfgs:
        for sk in splits:
            if sk > max(1, (K + bk - 1) // bk // 2):
                continue
            lib.cradle_gemm_tune(tm, tn, bk, dp, wk, sk)
            res.append((timeit(lambda: ops.gemm(layout, A, B, out=out, workspace=ws, repeat=REP)), tm, tn, f"{bk}d{dp}w{wk}", sk))
    lib.cradle_gemm_tune(-1, -1, -1, -1, -1, -1)
    res.sort()
    fl = 2.0 * M * N * K
    best = " | ".join(f"{t*1e6:6.1f}us t{tm}{tn} bk{bk} s{sk}" for t, tm, tn, bk, sk in res[:4])
    print(f"{what:16s} L{layout} {M}x{N}x{K}: auto {t_auto*1e6:6.1f}us {fl/t_auto/1e12:5.1f}TF || best {fl/res[0][0]/1e12:5.1f}TF: {best}", flush=True)
