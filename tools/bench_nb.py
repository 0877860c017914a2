"""Micro-benchmark of K1 (NB log-likelihood) at the Norman shape.  python tools/bench_nb.py [G] [N] [P]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cradle_vae_amd import ops

G = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
P = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = "cuda:0"
torch.manual_seed(0)
logits = torch.randn(P * N, G, device=dev)
X = torch.poisson(torch.exp(-1 + 1.5 * torch.randn(G, device=dev)).expand(N, G).contiguous())
lib = X.sum(1) + 1
gc = ops.nb_gene_consts(torch.zeros(G, device=dev))
grow = torch.full((P * N,), -1.0 / (P * N), device=dev)
ws = torch.empty(N * G * 8, device=dev)
dlc = torch.zeros(G, device=dev)
scr = logits.clone()


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


from cradle_vae_amd import hip
timeit(lambda: ops.nb_logprob_fwd_bwd(logits, X, lib, gc, grow, P, workspace=ws), iters=3000)  # ~100 ms: a chip that idled reads slow
for nt in (256, 128, 192, 256):
    for ps in (1,):
        hip.load().cradle_nb_tune(nt, ps)
        try:
            tf = timeit(lambda: ops.nb_logprob_fwd(logits, X, lib, gc, P))
            tb = timeit(lambda: ops.nb_logprob_fwd_bwd(logits, X, lib, gc, grow, P, workspace=ws))
        except hip.CradleHipError:  # no instantiation for this (threads, genes per thread)
            continue
        print(f"threads {nt:4d} split {ps}: fwd {tf*1e6:7.1f} us  fwd+bwd(+colsum) {tb*1e6:7.1f} us")
hip.load().cradle_nb_tune(-1, -1)
t_f = timeit(lambda: ops.nb_logprob_fwd(logits, X, lib, gc, P))
t_b = timeit(lambda: ops.nb_logprob_fwd_bwd(logits, X, lib, gc, grow, P, workspace=ws))
fwd_bytes = 4 * G * (P * N + N)
fb_alg = 4 * G * (3 * P * N + 2 * N)  # SURVEY 8(d) fwd+bwd, V=1
fb_moved = 4 * G * (2 * P * N + N)  # what the fused kernel actually moves
print(f"G={G} N={N} P={P}")
print(f"fwd     {t_f*1e6:8.1f} us  {fwd_bytes/t_f/1e9:8.1f} GB/s (algorithmic fwd bytes {fwd_bytes/1e6:.1f} MB)")
print(f"fwd+bwd {t_b*1e6:8.1f} us  {fb_alg/t_b/1e9:8.1f} GB/s algorithmic ({fb_alg/1e6:.1f} MB), {fb_moved/t_b/1e9:8.1f} GB/s moved ({fb_moved/1e6:.1f} MB)")
c = torch.empty_like(logits)
t_c = timeit(lambda: c.copy_(logits))
print(f"copy    {t_c*1e6:8.1f} us  {2*logits.numel()*4/t_c/1e9:8.1f} GB/s (torch copy of the logits, read+write)")
