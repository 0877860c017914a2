"""Which operand's leading dimension matters: A, B and C row pitches varied one by one on the step's products (GPU only)."""
import os, sys, itertools
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


def mat(rows, cols, ld):
    return torch.randn(rows, ld, device=dev)[:, :cols]


ws = torch.empty(1 << 26, device=dev)


def run(layout, M, N, K, la, lb, lc):
    if layout == ops.NT:
        A, B = mat(M, K, la), mat(N, K, lb)
    elif layout == ops.NN:
        A, B = mat(M, K, la), mat(K, N, lb)
    else:
        A, B = mat(K, M, la), mat(K, N, lb)
    out = torch.empty(M, lc, device=dev)[:, :N]
    return timeit(lambda: ops.gemm(layout, A, B, out=out, workspace=ws, repeat=REP))


def pads(n):
    return [n, (n + 31) // 32 * 32 + (32 if n % 32 == 0 else 0), (n + 31) // 32 * 32 + 48, (n + 31) // 32 * 32 + 64]


cases = [("dec out fwd", ops.NT, 2560, 5000, 400), ("dec out dgrad", ops.NN, 2560, 400, 5000), ("dec out wgrad", ops.TN, 5000, 400, 2560),
         ("hidden fwd", ops.NT, 2560, 400, 400), ("hidden dgrad", ops.NN, 2560, 400, 400), ("hidden wgrad", ops.TN, 400, 400, 2560),
         ("emb fwd", ops.NT, 1050, 400, 400), ("enc heads fwd", ops.NT, 7680, 400, 400), ("xW fwd", ops.NT, 1024, 400, 5000),
         ("xW wgrad", ops.TN, 400, 5000, 1024)]
sel = sys.argv[1:]
for what, layout, M, N, K in cases:
    if sel and not any(s in what for s in sel):
        continue
    a_cols = K if layout != ops.TN else M
    b_cols = K if layout == ops.NT else N
    base = run(layout, M, N, K, a_cols, b_cols, N)
    print(f"{what} ({'NT NN TN'.split()[[ops.NT, ops.NN, ops.TN].index(layout)]} {M} x {N} x {K}): pitches A {a_cols} B {b_cols} C {N}: {base*1e6:7.1f} us", flush=True)
    for name, idx, n in (("A", 0, a_cols), ("B", 1, b_cols), ("C", 2, N)):
        line = f"   {name}:"
        for ld in pads(n)[1:]:
            l = [a_cols, b_cols, N]
            l[idx] = ld
            line += f"  {ld}: {run(layout, M, N, K, *l)*1e6:7.1f}"
        print(line, flush=True)
    l = [pads(a_cols)[2], pads(b_cols)[2], pads(N)[2]]
    print(f"   all three {l}: {run(layout, M, N, K, *l)*1e6:7.1f} us", flush=True)
