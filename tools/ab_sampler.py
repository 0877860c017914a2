"""K9 under the library named by CRADLE_HIP_LIB (default: the tree's): a checksum of the raw draws and of the sums for
fixed seeds, and the time per launch -- run once per build and compare the lines (bit-equal draws = equal checksums).
python tools/ab_sampler.py [G] [C] [P] [mean log c]"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cradle_vae_amd import hip, ops

G, C, P = (int(a) for a in (sys.argv[1:4] + ["5000", "119", "128"][len(sys.argv) - 1:]))
LOGC = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
logits = 0.5 * torch.randn(P * C, G, device=dev, generator=g)
log_c = LOGC + 0.5 * torch.randn(G, device=dev, generator=g)
lib = 1e4 * torch.ones(1, device=dev)
consts = ops.nb_gene_consts(log_c)
digest = hashlib.sha256()
for method in ("mean", "perturbseq"):
    for sampler in ("nb", "mean"):
        sums = torch.zeros(C, (G + 3) & ~3, device=dev)[:, :G]
        _, samples = ops.nb_sample_accumulate(logits[:8 * C], lib, consts, 8, C, sums=sums, method=method, sampler=sampler,
                                              seed=11, return_samples=True)
        ops.nb_sample_accumulate(logits, lib, consts, P, C, sums=sums, method=method, sampler=sampler, seed=12)
        torch.cuda.synchronize()
        digest.update(sums.cpu().numpy().tobytes())
        digest.update(samples.cpu().numpy().tobytes())
if os.environ.get("NT"):
    hip.load().cradle_nb_sample_tune(int(os.environ["NT"]))
for sampler in ("nb", "mean"):
    sums = torch.zeros(C, (G + 3) & ~3, device=dev)[:, :G]
    for _ in range(3):
        ops.nb_sample_accumulate(logits, lib, consts, P, C, sums=sums, method="perturbseq", sampler=sampler, seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for i in range(n):
        ops.nb_sample_accumulate(logits, lib, consts, P, C, sums=sums, method="perturbseq", sampler=sampler, seed=2 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{os.environ.get('CRADLE_HIP_LIB', 'tree')} NT={os.environ.get('NT', 'auto')}: G={G} C={C} P={P} log c {LOGC} sampler={sampler}: {dt*1e3:.3f} ms/launch "
          f"{P*C*G/dt/1e9:.1f} G draws/s")
print("checksum", digest.hexdigest()[:16])
