"""A few launches of K9 (Gamma-Poisson sampler + ATE accumulation) at the ATE bench's launch shape (G = 5000, one slab of
119 conditions x a chunk of 128 particles = 15 232 rows), for the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in
separate runs).  python3 tools/pmc_sampler.py [G C P]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cradle_vae_amd import ops  # noqa: E402

G, C, P = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (5000, 119, 128)
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
logits = 0.5 * torch.randn(P * C, G, device=dev, generator=g)
consts = ops.nb_gene_consts(0.5 * torch.randn(G, device=dev, generator=g))
lib = 1e4 * torch.ones(1, device=dev)
sums = torch.zeros(C, (G + 3) & ~3, device=dev)[:, :G]
for i in range(10):
    ops.nb_sample_accumulate(logits, lib, consts, P, C, sums=sums, method="perturbseq", sampler="nb", seed=i)
torch.cuda.synchronize()
print("done")
