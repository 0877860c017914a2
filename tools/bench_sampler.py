"""K9 (Gamma-Poisson sampler + ATE accumulation) alone, per thread count: python tools/bench_sampler.py [G] [C] [P]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cradle_vae_amd import hip, ops

G = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 237
P = int(sys.argv[3]) if len(sys.argv) > 3 else 128
LOGC = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0  # mean log-concentration: high = every gene by inversion
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
logits = 0.5 * torch.randn(P * C, G, device=dev, generator=g)
log_c = LOGC + 0.5 * torch.randn(G, device=dev, generator=g)
lib = 1e4 * torch.ones(1, device=dev)
consts = ops.nb_gene_consts(log_c)
lib_ = hip.load()
mu = lib * torch.softmax(logits[:64], -1)
rate = (log_c.exp() / (mu + 1e-4)).clamp(min=1e-6)
print(f"mean log c {LOGC}: {float((rate < 0.25).float().mean())*100:.1f} % of the genes below rate 1/4 (Gamma-Poisson composition), "
      f"mean count {float(mu.mean()):.2f}")
for sampler in ("nb", "mean"):
    for method in ("mean",):
        for nt in (256, 1024):
            if nt > 0 and (G + 3) // 4 > nt * (8 if nt == 256 else 4):
                continue
            lib_.cradle_nb_sample_tune(nt)
            sums = torch.zeros(C, (G + 3) & ~3, device=dev)[:, :G]
            for _ in range(2):
                ops.nb_sample_accumulate(logits, lib, consts, P, C, sums=sums, method=method, sampler=sampler, seed=1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 5
            for i in range(n):
                ops.nb_sample_accumulate(logits, lib, consts, P, C, sums=sums, method=method, sampler=sampler, seed=2 + i)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            print(f"G={G} C={C} P={P} sampler={sampler:4s} method={method:10s} threads={nt:5d}: {dt*1e3:7.3f} ms  "
                  f"{P*C*G/dt/1e9:7.1f} G samples/s  {P*C*G*4/dt/1e9:7.0f} GB/s of logits", flush=True)
lib_.cradle_nb_sample_tune(-1)
