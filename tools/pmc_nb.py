"""A few launches of K1 (NB log-likelihood, fused fwd+bwd) at the Norman shape, for the
rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in separate runs).  python3 tools/pmc_nb.py [G N P]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cradle_vae_amd import ops  # noqa: E402

G, N, P = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (5000, 512, 5)
dev = "cuda:0"
torch.manual_seed(0)
logits = torch.randn(P * N, G, device=dev)
X = torch.poisson(torch.exp(-1 + 1.5 * torch.randn(G, device=dev)).expand(N, G).contiguous())
lib = X.sum(1) + 1
gc = ops.nb_gene_consts(torch.zeros(G, device=dev))
grow = torch.full((P * N,), -1.0 / (P * N), device=dev)
ws = torch.empty(N * G * 8, device=dev)
dl = torch.empty_like(logits)
for _ in range(10):
    ops.nb_logprob_fwd_bwd(logits, X, lib, gc, grow, P, workspace=ws)
    ops.nb_logprob_fwd(logits, X, lib, gc, P)
torch.cuda.synchronize()
print("done")
